"""GPU tests of the Deflate path: every stream must inflate back to the input through the CPU
oracle of the reference's Inflate (and zlib), framing must match the reference's encoders, and the
compressed size must stay within the stated tolerance of the reference's Deflate (the oracle)."""
import random
import zlib

import pytest

pytestmark = pytest.mark.gpu

from tests import streams  # noqa: E402

# Stated size tolerance, raw DEFLATE size against the reference's (oracle) at the same level:
#   benchmark corpora (config-3 log text, config-2 wiki-like text): <= +5 % at levels 1, 6 and 9 (measured r03:
#   +3.1 / +3.4 / +4.0 % on the log text, -2.1 / +2.6 / +2.9 % on the wiki text);
#   degenerate 12-word-vocabulary text: <= +10 % at level 6, <= +17 % at level 9 (measured +7.1 / +14.2 %; round 2's
#   4-way bucket alone was at +38 / +48 %);
#   never more than stored size + 5 bytes per 32 KiB chunk.
SIZE_TOLERANCE = {1: {"log": 1.05, "wiki": 1.05}, 6: {"log": 1.05, "wiki": 1.05, "text": 1.10}, 9: {"log": 1.05, "wiki": 1.05, "text": 1.17}}


@pytest.fixture(scope="module")
def amd(native_built):
    import archive_amd
    from archive_amd import _native as N
    assert N.lib().ahip_init(0) == 0, N.last_error()
    return archive_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def _corpora():
    from tools import corpus
    rnd = random.Random(5)
    return {
        "empty": b"", "one": b"a", "three": b"abc", "text": streams.text(200000, 2),
        "log": bytes(corpus.text(corpus.LOG, 1234, 0, 1 << 20)),
        "wiki": bytes(corpus.text(corpus.WIKI, 8, 0, 1 << 19)),
        "random": bytes(rnd.getrandbits(8) for _ in range(100000)),
        "zeros": bytes(300000), "period3": b"abc" * 50000,
        "ramp": bytes(i % 256 for i in range(0xfffff)),
        "chunk_exact": streams.text(32768 * 3, 7), "chunk_plus1": streams.text(32768 * 2 + 1, 8),
        "chunk_minus1": streams.text(32768 - 1, 9),
    }


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_round_trip_through_reference_inflate(amd, orc, level):
    for name, d in _corpora().items():
        z = amd.Deflate(d, level=level)
        c = z.get_bytes()
        assert z.crc32 == zlib.crc32(d)
        assert zlib.decompress(c, -15) == d, (name, level)
        st, out, pos = orc.inflate_raw(c + bytes(4))
        assert (st, out, pos) == (0, d, len(c)), (name, level)
        assert amd.Inflate(c + bytes(4)).get_bytes() == d  # and through the HIP inflate


@pytest.mark.parametrize("level", [1, 6, 9])
def test_size_within_tolerance_of_reference(amd, orc, level):
    for name in ("text", "log", "wiki", "zeros", "period3", "ramp", "random"):
        d = _corpora()[name]
        ours = len(amd.Deflate(d, level=level).get_bytes())
        ref = len(orc.deflate_raw(d, level)[0])
        chunks = (len(d) + 32767) // 32768
        assert ours <= len(d) + 5 * chunks + 16, name
        if name in SIZE_TOLERANCE[level]:
            assert ours <= ref * SIZE_TOLERANCE[level][name], (name, level, ours, ref, ours / ref)


@pytest.mark.parametrize("wbits", [9, 11, 14])
def test_window_bits(amd, orc, wbits):
    """Deflate(bytes, windowBits: 9..15) (deflate.dart:105-124): no match reaches further back than 2^W - 262, so the
    stream inflates with a window of that size (zlib checks it); size within the same tolerance of the reference's at
    the same window; the zlib header carries CINFO = W - 8 (_zlib_encoder_web.dart:42-63)."""
    for name in ("log", "wiki", "text", "random", "period3", "chunk_plus1"):
        d = _corpora()[name]
        c = amd.Deflate(d, level=6, window_bits=wbits).get_bytes()
        assert zlib.decompress(c, -wbits) == d, (name, wbits)          # zlib refuses distances beyond its window
        assert orc.inflate_raw(c + bytes(4))[:2] == (0, d)
        if name in ("log", "wiki"):
            ref = len(orc.deflate_raw(d, 6, window_bits=wbits)[0])
            assert len(c) <= ref * 1.08, (name, wbits, len(c), ref)
    d = _corpora()["log"][:100000]
    z = amd.ZLibEncoder().encode_bytes(d, window_bits=wbits)
    cmf = ((wbits - 8) << 4) | 8
    assert z[0] == cmf and (z[0] * 256 + z[1]) % 31 == 0 and z[1] < 31 and zlib.decompress(z) == d
    g = amd.GZipEncoder().encode_bytes(d, window_bits=wbits, mtime=0)
    assert orc.gzip_decode(g) == (0, d)


def test_invalid_parameters_are_silent(amd):
    assert amd.Deflate(b"hello", level=12).get_bytes() == b""
    assert amd.Deflate(b"hello", level=6, window_bits=20).get_bytes() == b""
    assert amd.Deflate(b"hello", level=6, window_bits=8).get_bytes() == b""


def test_encoder_framing(amd, orc):
    d = streams.text(70000, 11)
    g = amd.GZipEncoder().encode_bytes(d, mtime=0x01020304)
    assert g[:10] == bytes([0x1f, 0x8b, 8, 0, 4, 3, 2, 1, 0, 0xff])
    assert int.from_bytes(g[-8:-4], "little") == zlib.crc32(d) and int.from_bytes(g[-4:], "little") == len(d)
    assert orc.gzip_decode(g) == (0, d) and amd.GZipDecoder().decode_bytes(g) == d
    z = amd.ZLibEncoder().encode_bytes(d)
    assert z[:2] == b"\x78\x01" and int.from_bytes(z[-4:], "big") == zlib.adler32(d)
    assert orc.zlib_decode(z, verify=True) == (0, d) and amd.ZLibDecoder().decode_bytes(z, verify=True) == d
    assert zlib.decompress(amd.ZLibEncoder().encode_bytes(d, raw=True), -15) == d
    # cross test of test/gzip_test.dart:16-42: encoder -> decoder both ways
    assert amd.GZipDecoder().decode_bytes(orc.gzip_encode(d, 6)) == d


def test_two_encodes_are_byte_identical(amd):
    """The reference is one sequential pass (deflate.dart:997-1118): the same input gives the same bytes.  Here positions
    are inserted into the match tables a whole step at a time by many waves; which of several same-slot writers stays is
    DEFINED (the lowest position of the step, an LDS atomicMax on a step-reversed key), not left to the order the hardware
    serves the waves in -- so archives are reproducible and callers may compare compressed bytes."""
    from tools import corpus
    data = bytes(corpus.text(corpus.LOG, 1234, 0, 3 << 20)) + streams.text(400000, 5) + bytes(200000) + bytes(corpus.text(corpus.WIKI, 8, 0, 1 << 20))
    for level in (1, 4, 6, 9):
        for wb in (15, 11):
            first = amd.Deflate(data, level=level, window_bits=wb).get_bytes()
            assert zlib.decompress(first, -15) == data
            for _ in range(3):
                assert amd.Deflate(data, level=level, window_bits=wb).get_bytes() == first, (level, wb)
    g1 = amd.GZipEncoder().encode_bytes(data, mtime=1)
    assert all(amd.GZipEncoder().encode_bytes(data, mtime=1) == g1 for _ in range(2))


def test_carried_match_tables_change_nothing(amd, monkeypatch):
    """A workgroup of the match kernel takes a run of consecutive chunks and, from its second chunk on, keeps the hash
    tables the chunk before left (moved on by 32 KiB, the fifteen positions that could not be hashed yet added) instead of
    inserting the 32 KiB of history again.  That is an optimisation only: with one chunk per workgroup (AHIP_DF_RUNS=0,
    every chunk builds its tables from scratch) the stream must be the same byte for byte -- at every table shape (levels,
    small windows), for inputs that end inside a chunk, with runs of two chunks and runs of thirty."""
    from tools import corpus
    big = bytes(corpus.text(corpus.LOG, 77, 0, 40 << 20))          # 1 280 chunks: runs of five on 256 CUs by default
    mixed = (streams.text(300000, 3) + bytes(100000) + bytes(corpus.text(corpus.WIKI, 9, 0, 2 << 20)) + random.Random(4).randbytes(70000) +
             b"tail" * 999)                                        # 89 chunks, the last one short
    for data, cases, grids in ((big, ((6, 15),), (None,)), (mixed, ((1, 15), (4, 15), (6, 15), (9, 15), (6, 12), (6, 9), (3, 10)), ("45", "3"))):
        for level, wb in cases:
            monkeypatch.setenv("AHIP_DF_RUNS", "0")
            single = amd.Deflate(data, level=level, window_bits=wb).get_bytes()
            assert zlib.decompress(single, -15) == data
            for g in grids:
                if g is None:
                    monkeypatch.delenv("AHIP_DF_RUNS", raising=False)
                else:
                    monkeypatch.setenv("AHIP_DF_RUNS", g)
                assert amd.Deflate(data, level=level, window_bits=wb).get_bytes() == single, (len(data), level, wb, g)
    monkeypatch.delenv("AHIP_DF_RUNS", raising=False)


def test_sizes_around_every_edge(amd, orc, monkeypatch):
    """Inputs of 0 .. 70 bytes and of every length within three bytes of a chunk boundary (one, two and three chunks; the
    match kernel hashes 4 / 8 / 16 bytes ahead and a carried table gets its last fifteen positions late), with two
    workgroups taking runs of chunks and with one chunk per workgroup: the same bytes either way, and they inflate to the
    input through zlib and through the oracle's restatement of the reference's Inflate."""
    rnd = random.Random(23)
    base = streams.text(3 * 32768 + 40, 13)
    rep = (b"abcdefghijklmnop" * 5000)[: 3 * 32768 + 40]          # matches that run across every boundary
    sizes = list(range(0, 71)) + [k * 32768 + d for k in (1, 2, 3) for d in range(-17, 18)]
    for src in (base, rep):
        for n in sizes:
            d = src[:n]
            for level in (1, 6, 9):
                monkeypatch.setenv("AHIP_DF_RUNS", "2")
                a = amd.Deflate(d, level=level).get_bytes()
                monkeypatch.setenv("AHIP_DF_RUNS", "0")
                b = amd.Deflate(d, level=level).get_bytes()
                assert a == b, (n, level)
                assert zlib.decompress(a, -15) == d, (n, level)
            if n % 5 == 0:
                assert orc.inflate_raw(a + bytes(4))[:2] == (0, d), n
    monkeypatch.delenv("AHIP_DF_RUNS", raising=False)


def test_output_buffer_too_small_is_reported_and_respected(amd):
    """The chunk offsets are scanned on the device (deflate_offsets_kernel) and the gather runs before the host knows the
    total: a buffer that is too small must come back as AHIP_E_CAP with the size that IS needed, and no byte behind the
    capacity the caller gave may have been touched."""
    import ctypes
    import torch
    from archive_amd import _native as N
    L = N.lib()
    d = streams.text(32768 * 9 + 77, 21)
    d_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    full = torch.zeros(len(d) + 4096, dtype=torch.uint8, device="cuda")
    olen = ctypes.c_size_t(0)
    assert L.ahip_deflate_raw_device(d_in.data_ptr(), len(d), 6, 15, full.data_ptr(), full.numel(), ctypes.byref(olen), None) == 0
    need = olen.value
    ref = bytes(full[:need].cpu().numpy())
    assert zlib.decompress(ref, -15) == d
    for cap in (0, 1, need // 3, need - 1):
        out = torch.full((need + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        got = ctypes.c_size_t(0)
        rc = L.ahip_deflate_raw_device(d_in.data_ptr(), len(d), 6, 15, out.data_ptr(), cap, ctypes.byref(got), None)
        assert rc == N.AHIP_E_CAP and got.value == need, (cap, rc, got.value)
        assert bool((out[cap:] == 0xA5).all()), cap
    out = torch.full((need + 256,), 0xA5, dtype=torch.uint8, device="cuda")
    assert L.ahip_deflate_raw_device(d_in.data_ptr(), len(d), 6, 15, out.data_ptr(), need, ctypes.byref(olen), None) == 0
    assert bytes(out[:need].cpu().numpy()) == ref and bool((out[need:] == 0xA5).all())
