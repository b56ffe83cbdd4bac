"""bzip2 decoder: CPU tests pin the oracle (oracle/bzip2_oracle.c) against CPython's libbzip2 and the
reference's fixtures; GPU tests check the HIP block-parallel decoder against the oracle."""
import bz2
import random

import pytest

from tests import streams


def _corpora():
    from tools import corpus
    rnd = random.Random(1)
    return {
        "wiki": bytes(corpus.text(corpus.WIKI, 8, 0, 1500000)),
        "log": bytes(corpus.text(corpus.LOG, 1234, 0, 700000)),
        "zeros": bytes(1200000),
        "random": bytes(rnd.getrandbits(8) for _ in range(250000)),
        "runs": b"".join(bytes([rnd.randrange(4)]) * rnd.randrange(1, 600) for _ in range(3000)),
        "empty": b"", "one": b"x", "text": streams.text(99999, 4),
        # run-length escapes of every shape, also with the count byte equal to the run byte (4 + 97 x 'a')
        "escapes": b"".join(bytes([c]) * n + b"." for c in (0, 4, 97, 255) for n in (1, 2, 3, 4, 5, 7, 8, 9, 4 + c, 258, 259, 260, 263, 1000) for _ in range(3)),
        "fours": (b"aaaa" + b"bbbb" + b"aaaab" + b"\x04\x04\x04\x04" + b"\x00\x00\x00\x00\x00") * 2000,
    }


def _malformed():
    c = bz2.compress(b"hello world" * 1000)
    bad_block = bytearray(bz2.compress(streams.text(50000, 1), 1))
    bad_block[len(bad_block) // 2] ^= 0x10
    flips = {}
    small = bz2.compress(streams.text(20000, 2) + bytes(300) + b"abcd" * 50, 9)
    for k in range(24):  # single-bit damage all over one block: whatever the oracle makes of it, so must the GPU
        b = bytearray(small)
        pos = 40 + (k * 7919) % (len(small) - 60)
        b[pos] ^= 1 << (k % 8)
        flips["flip%d" % k] = bytes(b)
    # a block magic (byte-aligned) dropped into the middle of a block's payload, and one behind the end of the stream
    two = bytearray(bz2.compress(streams.text(150000, 3), 1))
    two[len(two) // 3:len(two) // 3 + 6] = bytes.fromhex("314159265359")
    return {**flips, "magic_inside_block": bytes(two), "magic_after_end": c + bytes.fromhex("314159265359") + bytes(40),
            "truncated_tail": c[:-3], "truncated_mid": c[:50], "no_magic": b"BZh9" + bytes(20), "not_bz": b"hello",
            "short": b"BZ", "header_only": b"BZh9", "bad_level": b"BZhx" + c[4:], "two_streams": c + c,
            "corrupt_block": bytes(bad_block)}


def test_oracle_matches_libbzip2():
    from oracle import pyoracle as orc
    for name, d in _corpora().items():
        for level in (1, 9):
            c = bz2.compress(d, level)
            assert orc.bzip2_decode(c, verify=True, cap=len(d) + 64) == (0, d), (name, level)


def test_oracle_malformed():
    from oracle import pyoracle as orc
    m = _malformed()
    assert orc.bzip2_decode(m["truncated_tail"])[0] == 2
    assert orc.bzip2_decode(m["truncated_mid"])[0] == 2
    assert orc.bzip2_decode(m["no_magic"]) == (1, b"")
    assert orc.bzip2_decode(m["not_bz"]) == (1, b"")
    assert orc.bzip2_decode(m["header_only"]) == (0, b"")
    assert orc.bzip2_decode(m["short"])[0] == 2  # the third readByte() throws before the compare
    assert orc.bzip2_decode(m["bad_level"]) == (1, b"")
    assert orc.bzip2_decode(m["two_streams"]) == (0, b"hello world" * 1000)  # ONE stream only (bzip2_decoder.dart:70-85)


@pytest.mark.gpu
@pytest.mark.parametrize("huffman_pass", ["position_parallel", "serial_wave"])
def test_gpu_matches_oracle(native_built, monkeypatch, huffman_pass):
    """Both forms of the Huffman pass: bz_jump_tiles / bz_group_starts / bz_decode_groups (the default), and the serial
    wave bz_decode_block that irregular blocks fall back to (AHIP_BZ_SERIAL_HUFFMAN=1 sends every block there)."""
    if huffman_pass == "serial_wave":
        monkeypatch.setenv("AHIP_BZ_SERIAL_HUFFMAN", "1")
    import archive_amd
    from archive_amd import _native as N
    from archive_amd import errors
    from oracle import pyoracle as orc
    assert N.lib().ahip_init(0) == 0

    def run(data, verify):
        d = archive_amd.BZip2Decoder()
        try:
            out = d.decode_bytes(data, verify=verify)
            return d.last_status, out
        except errors.RangeError:
            return 2, None
    for name, d in _corpora().items():
        for level in (1, 9):
            c = bz2.compress(d, level)
            assert run(c, True) == (0, d), (name, level)
    for name, c in _malformed().items():
        for verify in (False, True):
            st, out = orc.bzip2_decode(c, verify=verify)
            got = run(c, verify)
            assert got == ((2, None) if st == 2 else (st, out)), (name, verify, got[0], st)


@pytest.mark.gpu
def test_gpu_device_resident_api(native_built):
    import ctypes

    import torch
    from archive_amd import _native as N
    from tools import corpus
    assert N.lib().ahip_init(0) == 0
    data = bytes(corpus.text(corpus.WIKI, 8, 0, 2500000))
    comp = bz2.compress(data, 9)  # three 900k blocks
    d_in = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
    olen = ctypes.c_size_t()
    small = torch.empty(100, dtype=torch.uint8, device="cuda")
    assert N.lib().ahip_bzip2_decode_device(d_in.data_ptr(), d_in.numel(), 1, small.data_ptr(), 100, ctypes.byref(olen), None) == -1
    assert olen.value == len(data)
    d_out = torch.empty(len(data), dtype=torch.uint8, device="cuda")
    assert N.lib().ahip_bzip2_decode_device(d_in.data_ptr(), d_in.numel(), 1, d_out.data_ptr(), d_out.numel(),
                                            ctypes.byref(olen), None) == 0
    assert olen.value == len(data) and bytes(d_out.cpu().numpy()) == data
    # ... on the CALLER'S stream (round 6: the argument used to be ignored): the input arrives by an asynchronous copy on a
    # non-blocking stream of the caller's, with a long kernel queued in front of it -- the decode must wait for both there,
    # and what the caller queues behind the call on that stream must see the output
    side = torch.cuda.Stream()
    pinned = torch.frombuffer(bytearray(comp), dtype=torch.uint8).pin_memory()
    d_in2 = torch.empty(len(comp), dtype=torch.uint8, device="cuda")
    d_out2 = torch.zeros(len(data), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        busy = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        for _ in range(20):
            busy.normal_()                      # ~ tens of milliseconds of work in front of the copy
        d_in2.copy_(pinned, non_blocking=True)
        assert N.lib().ahip_bzip2_decode_device(d_in2.data_ptr(), d_in2.numel(), 1, d_out2.data_ptr(), d_out2.numel(),
                                                ctypes.byref(olen), ctypes.c_void_p(side.cuda_stream)) == 0
        total = d_out2.to(torch.int64).sum()    # queued behind the call on the same stream
    side.synchronize()
    assert olen.value == len(data) and int(total.item()) == sum(data) and bytes(d_out2.cpu().numpy()) == data


@pytest.mark.gpu
def test_gpu_many_blocks_in_small_batches(native_built, monkeypatch):
    """64 blocks of 900 k (block boundaries fall on arbitrary bit offsets), work memory limited so that the candidates
    are taken in several batches along the chain; size query with a too-small buffer reports the whole size."""
    import ctypes
    import zlib

    import torch
    from archive_amd import _native as N
    from tools import corpus
    assert N.lib().ahip_init(0) == 0
    monkeypatch.setenv("AHIP_BZ_BATCH_BYTES", str(64 << 20))  # about 11 blocks per batch
    data = bytes(corpus.text(corpus.WIKI, 8, 0, 64 * 900000 - 12345))
    comp = bz2.compress(data, 9)
    d_in = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
    olen = ctypes.c_size_t()
    small = torch.empty(5000000, dtype=torch.uint8, device="cuda")
    assert N.lib().ahip_bzip2_decode_device(d_in.data_ptr(), d_in.numel(), 1, small.data_ptr(), small.numel(), ctypes.byref(olen), None) == -1
    assert olen.value == len(data)
    d_out = torch.empty(len(data), dtype=torch.uint8, device="cuda")
    assert N.lib().ahip_bzip2_decode_device(d_in.data_ptr(), d_in.numel(), 1, d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None) == 0
    assert olen.value == len(data)
    crc = ctypes.c_uint32()
    assert N.lib().ahip_crc32_device(d_out.data_ptr(), len(data), 0, ctypes.byref(crc), None) == 0
    assert crc.value == zlib.crc32(data)
    assert bytes(d_out[:100000].cpu().numpy()) == data[:100000] and bytes(d_out[-100000:].cpu().numpy()) == data[-100000:]


@pytest.mark.gpu
def test_gpu_damage_in_header_and_selectors(native_built):
    """Round 4 moved three serial steps onto the whole wave / the lanes: the selectors (zeros delimit the unary numbers; the
    serial loop only takes over when something is out of the ordinary), the move-to-front list (a part per lane) and the
    ranking of the inverse transform's sublists (two levels).  Damage aimed at what they read -- every bit of the header and
    the first selectors, then bits spread over the selector and code-length area and the block's payload, truncations inside
    the selectors -- must leave the verdict and the bytes the oracle's, with and without CRC verification."""
    import archive_amd
    from archive_amd import _native as N
    from archive_amd import errors
    from oracle import pyoracle as orc
    assert N.lib().ahip_init(0) == 0
    rnd = random.Random(11)
    # several tables, ~ 800 groups of 50 symbols (a dozen selectors per lane)
    data = streams.text(24000, 4) + bytes(rnd.getrandbits(8) for _ in range(6000)) + bytes(500) + streams.text(9000, 6)
    c = bz2.compress(data, 9)

    pos = [None]

    def run(buf, verify):
        d = archive_amd.BZip2Decoder()
        try:
            out = d.decode_bytes(buf, verify=verify)
            pos[0] = d.input_position
            return d.last_status, out
        except errors.RangeError:
            return 2, None
    cases = []
    for bit in range(10 * 8, 10 * 8 + 19 * 8 + 3 + 15 + 64):        # block CRC, origPtr, the in-use maps' head ... first selectors
        if bit == 14 * 8:  # the `randomised` flag: the obsolete mode is AHIP_E_UNSUPPORTED here, not a verdict
            continue
        b = bytearray(c); b[bit >> 3] ^= 0x80 >> (bit & 7); cases.append(bytes(b))
    for k in range(160):                                              # selectors, code lengths, payload
        bit = 45 * 8 + (k * 104729) % ((len(c) - 60) * 8)
        b = bytearray(c); b[bit >> 3] ^= 0x80 >> (bit & 7); cases.append(bytes(b))
    for cut in list(range(4, 130)) + list(range(130, len(c) - 80, 7)) + list(range(len(c) - 80, len(c))):  # every end inside the header (52 and 53: inside the
        cases.append(c[:cut])                                                            # selector count -- RangeError, not `false`), every 7th behind it, every one in the end-of-stream marker and CRCs
    seen = set()
    for i, buf in enumerate(cases):
        for verify in (False, True):
            st, out = orc.bzip2_decode(buf, verify=verify)
            opos = orc.bzip2_last_position()
            got = run(buf, verify)
            assert got == ((2, None) if st == 2 else (st, out)), (i, verify, got[0], st)
            if st in (0, 1):   # where the reader stood -- behind the stream, or at the check that failed (ahip_last_consumed)
                assert pos[0] == opos, (i, verify, st, pos[0], opos)
            seen.add(st)
    assert {0, 1, 2} <= seen, seen  # decoded all the same (the damage hit nothing that is checked), `false`, RangeError


def test_bench_stream_of_repeated_blocks():
    """bench.py measures a long bzip2 stream it builds from a short one (bz2_repeat: the block section spliced bit by bit,
    the stream CRC folded from the block CRCs): libbzip2 and the oracle must read it as the data repeated."""
    import bench
    from oracle import pyoracle as orc
    data = streams.text(250000, 8) + bytes(5000)
    c = bz2.compress(data, 1)                      # three blocks, none of them byte aligned but the first
    big, n = bench.bz2_repeat(c, 3)
    assert n == 9 and bz2.decompress(big) == data * 3
    assert orc.bzip2_decode(big, verify=True) == (0, data * 3)


@pytest.mark.gpu
def test_gpu_small_stream_flipped_everywhere(native_built):
    """A two-block stream of 14 KB with single bits flipped in turn -- magics, CRCs, origPtr, maps, selectors, code
    lengths, payload, end-of-stream marker: verdict and bytes are the oracle's, with and without CRC verification.
    Every 37th bit of what is structure (both blocks' headers and tables, the trailer), every 370th of the payloads (a
    payload flip usually leaves a block whose pointer cycle is shorter than the block: the one-lane inverse transform,
    0.1 s per 100 k block).  The dense payload sweep runs on the CPU against the same host chain
    (tests/test_bzip2_chain.py::test_two_block_stream_flipped_everywhere); round 4 found there -- on the GPU, then --
    that a block failing BEHIND bytes it had written lost them (bzip2_decoder.dart:612-631)."""
    import archive_amd
    from archive_amd import _native as N
    from archive_amd import errors
    from oracle import pyoracle as orc
    assert N.lib().ahip_init(0) == 0
    data = streams.text(110000, 9) + bytes(3000) + streams.text(2000, 10)
    c = bz2.compress(data, 1)      # two blocks
    starts = orc.bzip2_block_bits(c)    # [first block, second block, end-of-stream marker]
    assert len(starts) == 3

    pos = [None]

    def run(buf, verify):
        d = archive_amd.BZip2Decoder()
        try:
            out = d.decode_bytes(buf, verify=verify)
            pos[0] = d.input_position
            return d.last_status, out
        except errors.RangeError:
            return 2, None
        except errors.ArchiveHipError as e:   # the obsolete randomised mode: not a verdict
            assert "randomised" in str(e)
            return None
    n = 0
    nbits = len(c) * 8
    for bit in range(0, nbits, 37):
        structural = bit < 1500 or starts[1] - 200 < bit < starts[1] + 1500 or bit > starts[2] - 200
        if not structural and (bit // 37) % 10:
            continue
        buf = bytearray(c); buf[bit >> 3] ^= 0x80 >> (bit & 7); buf = bytes(buf)
        for verify in (False, True):
            got = run(buf, verify)
            if got is None:
                continue
            st, out = orc.bzip2_decode(buf, verify=verify)
            assert got == ((2, None) if st == 2 else (st, out)), (bit, verify, got[0], st)
            if st in (0, 1):   # where decodeStream leaves its InputStream, `true` or `false` (ahip_last_consumed)
                assert pos[0] == orc.bzip2_last_position(), (bit, verify, st, pos[0], orc.bzip2_last_position())
            n += 1
    assert n > 600


@pytest.mark.gpu
def test_gpu_block_that_fails_behind_its_bytes(native_built):
    """The reference writes a block's bytes as it walks the inverse transform and can fail AFTER writing: a run of four
    equal bytes whose count byte lies beyond the block's data (`cNBlockUsed > sSaveNBlockPP`, bzip2_decoder.dart:612-631).
    decodeBytes keeps what was written.  Streams of that kind, found on the CPU (the oracle's block function), through the
    GPU: one block, and the second of two (the host chain places the partial bytes behind the first block's)."""
    import archive_amd
    from archive_amd import _native as N
    from oracle import pyoracle as orc
    assert N.lib().ahip_init(0) == 0
    data = streams.text(110000, 9) + bytes(3000) + streams.text(2000, 10)
    c = bz2.compress(data, 1)
    found = 0
    for bit in range(14 * 8 * 8, len(c) * 8 - 100, 37):
        buf = bytearray(c); buf[bit >> 3] ^= 0x80 >> (bit & 7); buf = bytes(buf)
        st, out = orc.bzip2_decode(buf, verify=False)
        if not (st == 1 and out and len(out) not in (99981, len(data))):
            continue
        found += 1
        for verify in (False, True):
            st, out = orc.bzip2_decode(buf, verify=verify)
            d = archive_amd.BZip2Decoder()
            got = d.decode_bytes(buf, verify=verify)
            assert (d.last_status, got) == (st, out), (bit, verify, d.last_status, st, len(got), len(out))
        if found >= 12:
            break
    assert found >= 3


@pytest.mark.gpu
def test_gpu_failing_getmtfval_is_not_the_end_of_the_block(native_built):
    """_getMtfVal returns -1 (selectors used up, a code longer than 20 bits, an index outside the alphabet) and only its
    FIRST result is checked (bzip2_decoder.dart:271; :304 and :385 are not): the reference goes on with -1 as a symbol --
    reads _mtfa[_mtfbase[0] - 2], stores that byte, decodes on from wherever the bit reader stands, until an end-of-block
    symbol, nblockMAX, the end of the input or index -1.  Damage in the code lengths makes such codes; bz_block_exact runs
    the reference's loop for those blocks.  Every 3rd bit of the header / selector / code-length area, every 29th behind."""
    import archive_amd
    from archive_amd import _native as N
    from archive_amd import errors
    from oracle import pyoracle as orc
    assert N.lib().ahip_init(0) == 0
    data = streams.text(9000, 3) + bytes(700) + b"abcd" * 40
    c = bz2.compress(data, 1)
    seen = set()
    for bit in list(range(14 * 8 + 1, 1400, 3)) + list(range(1400, len(c) * 8 - 80, 29)):
        buf = bytearray(c); buf[bit >> 3] ^= 0x80 >> (bit & 7); buf = bytes(buf)
        for verify in (False, True):
            st, out = orc.bzip2_decode(buf, verify=verify)
            d = archive_amd.BZip2Decoder()
            try:
                got = (0, d.decode_bytes(buf, verify=verify))
                got = (d.last_status, got[1])
            except errors.RangeError:
                got = (2, None)
            assert got == ((2, None) if st == 2 else (st, out)), (bit, verify, got[0], st)
            seen.add(st)
    assert seen == {0, 1, 2}
