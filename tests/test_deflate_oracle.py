"""CPU-only: pins the Deflate oracle (oracle/deflate_oracle.c).  The reference pins no compressed
bytes, so the anchor is C zlib 1.2.11: with the reference's block-truncation heuristic switched off
the restatement must be byte-identical to zlib at levels 1..9; with it on (the reference's
behaviour) every stream must still round-trip through the Inflate oracle and zlib."""
import random
import zlib

import pytest

from oracle import pyoracle as orc
from tests import streams


def _corpora():
    from tools import corpus
    rnd = random.Random(5)
    return {
        "text": streams.text(120000, 2),
        "log": bytes(corpus.text(corpus.LOG, 1234, 0, 1 << 18)),
        "wiki": bytes(corpus.text(corpus.WIKI, 8, 0, 1 << 17)),
        "random": bytes(rnd.getrandbits(8) for _ in range(70000)),
        "zeros": bytes(200000),
        "ramp": bytes(i % 256 for i in range(0xfffff)),  # test/deflate_test.dart:12-44
        "empty": b"", "one": b"a", "period3": b"abc" * 30000,
    }


@pytest.mark.parametrize("level", range(1, 10))
def test_byte_identical_to_zlib_without_the_truncation_heuristic(level):
    for name, d in _corpora().items():
        z = zlib.compressobj(level, zlib.DEFLATED, -15, 8)
        ref = z.compress(d) + z.flush()
        got, crc = orc.deflate_raw(d, level, truncate_heuristic=False)
        assert got == ref, (name, level)
        assert crc == zlib.crc32(d)


@pytest.mark.parametrize("wbits", [9, 10, 12, 14])
def test_window_bits_byte_identical_to_zlib(wbits):
    """Deflate(bytes, windowBits: 9..14) (deflate.dart:105-124): the window shrinks, nothing else changes."""
    for name, d in _corpora().items():
        for level in (1, 6, 9):
            z = zlib.compressobj(level, zlib.DEFLATED, -wbits, 8)
            ref = z.compress(d) + z.flush()
            got, _ = orc.deflate_raw(d, level, truncate_heuristic=False, window_bits=wbits)
            assert got == ref, (name, level, wbits)
            c, _ = orc.deflate_raw(d, level, window_bits=wbits)
            assert zlib.decompress(c, -15) == d
    assert orc.deflate_raw(b"hello", 6, window_bits=8)[0] == b"" and orc.deflate_raw(b"hello", 6, window_bits=16)[0] == b""


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_reference_behaviour_round_trips(level):
    for name, d in _corpora().items():
        c, crc = orc.deflate_raw(d, level)
        assert zlib.decompress(c, -15) == d, (name, level)
        st, out, pos = orc.inflate_raw(c + bytes(4))
        assert (st, out, pos) == (0, d, len(c)), (name, level)
        assert crc == zlib.crc32(d)
        if level == 0 and len(d) > 70000:
            # old deflate_stored: first block holds 65 531 bytes (deflate.dart:694-698)
            assert c[0] == 0 and int.from_bytes(c[1:3], "little") == 65531


def test_invalid_level_writes_nothing():
    assert orc.deflate_raw(b"hello", 12)[0] == b""


def test_gzip_and_zlib_framing():
    d = streams.text(50000, 9)
    g = orc.gzip_encode(d, 6, mtime=0x01020304)
    assert g[:10] == bytes([0x1f, 0x8b, 8, 0, 4, 3, 2, 1, 0, 0xff])
    assert int.from_bytes(g[-8:-4], "little") == zlib.crc32(d) and int.from_bytes(g[-4:], "little") == len(d)
    assert orc.gzip_decode(g) == (0, d)
    z = orc.zlib_encode(d, 6)
    assert z[:2] == b"\x78\x01" and int.from_bytes(z[-4:], "big") == zlib.adler32(d)
    assert zlib.decompress(z) == d and orc.zlib_decode(z, verify=True) == (0, d)
