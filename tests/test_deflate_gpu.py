"""GPU tests of the Deflate path: every stream must inflate back to the input through the CPU
oracle of the reference's Inflate (and zlib), framing must match the reference's encoders, and the
compressed size must stay within the stated tolerance of the reference's Deflate (the oracle)."""
import random
import zlib

import pytest

pytestmark = pytest.mark.gpu

from tests import streams  # noqa: E402

# Stated size tolerance: at level 6 the HIP encoder's raw DEFLATE size may
# exceed the reference's (oracle, same level) by at most 10 % on the benchmark corpora (config-3 log
# text: measured +7.7 %, config-2 wiki-like text: +6.0 %), by at most 45 % on the degenerate
# 12-word-vocabulary text (measured +38 %: zlib's 128-deep hash chains find much longer matches than
# a 4-way bucket), and never exceeds stored size + 5 bytes per 32 KiB chunk.
SIZE_TOLERANCE_L6 = {"log": 1.10, "wiki": 1.10, "text": 1.45}


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


def test_size_within_tolerance_of_reference_level6(amd, orc):
    for name in ("text", "log", "wiki", "zeros", "period3", "ramp", "random"):
        d = _corpora()[name]
        ours = len(amd.Deflate(d, level=6).get_bytes())
        ref = len(orc.deflate_raw(d, 6)[0])
        chunks = (len(d) + 32767) // 32768
        assert ours <= len(d) + 5 * chunks + 16, name
        if name in SIZE_TOLERANCE_L6:
            assert ours <= ref * SIZE_TOLERANCE_L6[name], (name, ours, ref)


def test_invalid_parameters_are_silent(amd):
    assert amd.Deflate(b"hello", level=12).get_bytes() == b""
    assert amd.Deflate(b"hello", level=6, window_bits=20).get_bytes() == b""


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
