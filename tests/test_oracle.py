"""CPU-only: pins the oracle (oracle/inflate_oracle.c) against the reference's golden vectors
and against C zlib, so that it can be trusted as the checker of the HIP path."""
import gzip
import zlib

import pytest

from oracle import pyoracle as orc
from tests import streams


def test_golden_vectors(golden):
    for v in golden["vectors"]:
        kind, data, exp = v["kind"], v["input"], v["expected"]
        if kind == "raw":
            st, out, pos = orc.inflate_raw(data)
        elif kind == "gzip":
            st, out = orc.gzip_decode(data)
        elif kind == "zlib_first":
            st, out = orc.zlib_decode(data, verify=True)
        elif kind == "zlib_verify":
            st, out = orc.zlib_decode(data, verify=True)
        elif kind == "bzip2":
            st, out = orc.bzip2_decode(data, verify=True)
        else:
            raise AssertionError(kind)
        assert st == orc.ORC_OK, v["name"]
        assert out == exp, v["name"]


def test_checksum_kats(golden):
    def expand(s):
        if "*" in s:
            h, n = s.split("*")
            return bytes.fromhex(h) * int(n)
        return bytes.fromhex(s)
    for h, want in golden["checksum_kat"]["crc32"]:
        assert orc.crc32(expand(h)) == int(want, 16)
    for h, want in golden["checksum_kat"]["adler32"]:
        assert orc.adler32(expand(h)) == int(want, 16)
    d = streams.text(100000, 9)
    assert orc.crc32(d) == zlib.crc32(d)
    assert orc.adler32(d) == zlib.adler32(d)
    assert orc.crc32(d[5000:], orc.crc32(d[:5000])) == zlib.crc32(d)
    assert orc.adler32(d[5000:], orc.adler32(d[:5000])) == zlib.adler32(d)


@pytest.mark.parametrize("name,raw", streams.valid_raw_streams())
def test_valid_streams_match_zlib(name, raw):
    st, out, pos = orc.inflate_raw(raw)
    if name == "stored_zero_len_bad_nlen":  # zlib rejects it; the reference accepts (quirk q4)
        assert (st, out, pos) == (orc.ORC_OK, b"", 5)
        return
    d = zlib.decompressobj(-15)
    exp = d.decompress(raw)
    assert out == exp
    if st == orc.ORC_FALSE:
        # quirk q2: _readCodeByTable wants maxCodeLength bits even for a shorter code, so a raw
        # stream whose end-of-block code sits in the last bits "fails" after producing everything
        assert pos == len(raw)
    else:
        assert st == orc.ORC_OK
        assert pos == len(raw) - len(d.unused_data)
    # with trailer-like bytes behind it (as inside gzip/zlib) the same stream is clean
    st2, out2, pos2 = orc.inflate_raw(raw + b"\x00\x00\x00\x00")
    assert (st2, out2, pos2) == (orc.ORC_OK, exp, len(raw))


def test_deflate_roundtrip_ramp():
    # test/deflate_test.dart:12-44 shape: 0xfffff-byte i%256 ramp, levels 0/1/9
    ramp = bytes(i % 256 for i in range(0xfffff))
    for level in (0, 1, 9):
        st, out, _ = orc.inflate_raw(streams.raw_deflate(ramp, level))
        assert st == 0 and out == ramp


def test_malformed_streams_are_silent():
    for name, raw in streams.malformed_raw_streams():
        st, out, pos = orc.inflate_raw(raw)
        if name == "far_distance_first_member":
            assert st == orc.ORC_RANGE
            continue
        if name in ("no_final_block", "empty_input", "trailing_bytes_after_final"):
            assert st == orc.ORC_OK, name
        else:
            assert st == orc.ORC_FALSE, name
        if name.startswith("truncated"):
            full = zlib.decompressobj(-15).decompress(raw)  # zlib's partial output is a prefix oracle
            assert out[:len(full)] == full[:len(out)]
        assert pos <= len(raw)


def test_gzip_framing():
    a, b = streams.text(1000, 1), streams.text(70000, 2)
    g = streams.gz_member(a, name=b"a.txt", comment=b"hi", hcrc=True) + streams.gz_member(b, extra=b"XY\x02\x00zz") \
        + streams.bgzf_member(a)
    assert orc.gzip_decode(g) == (orc.ORC_OK, a + b + a)
    # trailing zero padding: falls into the zlib branch, returns false, keeps output (quirk q9)
    assert orc.gzip_decode(g + bytes(5)) == (orc.ORC_FALSE, a + b + a)
    # truncated trailer: RangeError
    assert orc.gzip_decode(g[:-3])[0] == orc.ORC_RANGE
    # not gzip at all -> zlib decoder on the same (little-endian) stream
    z = zlib.compress(a)
    assert orc.gzip_decode(z) == (orc.ORC_OK, a)
    assert orc.gzip_decode(z, verify=True) == (orc.ORC_FALSE, b"")  # adler read little-endian
    assert orc.gzip_decode(b"") == (orc.ORC_OK, b"")


def test_zlib_framing_quirks():
    a, b = streams.text(3000, 3), streams.text(4000, 4)
    za, zb = zlib.compress(a), zlib.compress(b)
    assert orc.zlib_decode(za + zb, verify=True) == (orc.ORC_OK, a + b)
    # deferred flush (quirk q7): junk after a valid stream drops the last member
    assert orc.zlib_decode(za + b"\x00\x00", verify=True) == (orc.ORC_FALSE, b"")
    assert orc.zlib_decode(za + zb + b"\x01\x02") == (orc.ORC_FALSE, a)
    bad = bytearray(za + zb)
    bad[-1] ^= 1
    assert orc.zlib_decode(bytes(bad), verify=True) == (orc.ORC_FALSE, a)
    assert orc.zlib_decode(bytes(bad), verify=False) == (orc.ORC_OK, a + b)
    assert orc.zlib_decode(streams.raw_deflate(a), raw=True) == (orc.ORC_OK, a)
    # cmf & 8 method test: 0x08 0x1d passes both checks
    assert (0x08 * 256 + 0x1d) % 31 == 0
    assert orc.zlib_decode(bytes([0x08, 0x1d]) + streams.raw_deflate(a) + zlib.adler32(a).to_bytes(4, "big"),
                           verify=True) == (orc.ORC_OK, a)
    assert orc.zlib_decode(za[:-2])[0] == orc.ORC_RANGE


def test_oracle_large_multimember():
    from tools import corpus
    comp, plain = corpus.make_gzip(n_members=64, want_plain=True)
    st, out = orc.gzip_decode(bytes(comp), cap=len(plain) + 16)
    assert st == 0 and out == bytes(plain)
    assert gzip.decompress(bytes(comp)) == bytes(plain)


def test_stream_position_after_decodestream():
    """Where decodeStream leaves its InputStream (what ahip_last_consumed is held against on the GPU): all of it when it
    returns true; on `false` where the failing check stood -- two header bytes in (method / FCHECK), behind the four bytes of
    a dictionary id, behind the Adler-32 that did not match (_zlib_decoder_web.dart:53-99); the gzip decoder rewinds to the
    start of what is not a gzip header before it hands over to the zlib one (_gzip_decoder_web.dart:31-37)."""
    import zlib

    from oracle import pyoracle as orc
    from tests import streams
    a, b = streams.text(3000, 3), streams.text(4000, 4)
    za, zb = zlib.compress(a), zlib.compress(b)
    assert orc.zlib_decode(za + zb, verify=True) == (0, a + b) and orc.last_position() == len(za + zb)
    assert orc.zlib_decode(za + b"\x77\x01" + zb)[0] == 1 and orc.last_position() == len(za) + 2      # method
    assert orc.zlib_decode(za + b"\x78\x02" + zb)[0] == 1 and orc.last_position() == len(za) + 2      # FCHECK
    assert orc.zlib_decode(za + b"\x78\x20" + zb)[0] == 1 and orc.last_position() == len(za) + 6      # FDICT: + readUint32
    bad = bytearray(za + zb)
    bad[len(za) - 1] ^= 1                                                                           # first member's Adler-32
    assert orc.zlib_decode(bytes(bad), verify=True)[0] == 1 and orc.last_position() == len(za)
    g = streams.gz_member(a) + streams.gz_member(b)
    assert orc.gzip_decode(g) == (0, a + b) and orc.last_position() == len(g)
    assert orc.gzip_decode(g + b"junk")[0] == 1 and orc.last_position() == len(g) + 2                 # zlib fallback: `ju` is no header
    assert orc.gzip_decode(g + za)[0] == 0 and orc.last_position() == len(g) + len(za)               # ... a zlib stream behind gzip members is decoded
