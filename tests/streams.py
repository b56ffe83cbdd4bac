"""Hand-built DEFLATE/gzip/zlib test streams shared by the oracle and GPU parity tests."""
import gzip
import random
import struct
import zlib


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    return c.compress(data) + c.flush()


def stored_block(data, final=True):
    assert len(data) <= 65535
    return bytes([1 if final else 0]) + struct.pack("<HH", len(data), len(data) ^ 0xffff) + data


def text(n, seed=1):
    rnd = random.Random(seed)
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu"]
    out = bytearray()
    while len(out) < n:
        out += rnd.choice(words).encode() + (b" " if rnd.random() < 0.8 else b"\n")
    return bytes(out[:n])


def gz_member(data, level=6, extra=None, name=None, comment=None, hcrc=False, mtime=0):
    flags = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    h = bytes([0x1f, 0x8b, 8, flags]) + struct.pack("<I", mtime) + bytes([0, 255])
    if extra is not None:
        h += struct.pack("<H", len(extra)) + extra
    if name is not None:
        h += name + b"\0"
    if comment is not None:
        h += comment + b"\0"
    if hcrc:
        h += struct.pack("<H", zlib.crc32(h) & 0xffff)
    return h + raw_deflate(data, level) + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


def bgzf_member(data, level=6):
    body = raw_deflate(data, level)
    total = 18 + len(body) + 8
    assert total <= 65536
    h = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", total - 1)
    return h + body + struct.pack("<II", zlib.crc32(data), len(data))


def valid_raw_streams():
    """(name, raw deflate bytes) covering stored / fixed / dynamic / mixed / empty / long matches."""
    rnd = random.Random(7)
    cases = []
    cases.append(("empty_fixed", raw_deflate(b"")))
    cases.append(("one_byte", raw_deflate(b"a")))
    cases.append(("stored_only", raw_deflate(text(70000), level=0)))
    cases.append(("fixed_small", raw_deflate(b"hello hello hello hello", strategy=zlib.Z_FIXED)))
    cases.append(("fixed_64k", raw_deflate(text(65536, 1), strategy=zlib.Z_FIXED)))
    cases.append(("dynamic_text", raw_deflate(text(200000, 2))))
    cases.append(("dynamic_l1", raw_deflate(text(100000, 3), level=1)))
    cases.append(("dynamic_l9", raw_deflate(text(100000, 4), level=9)))
    cases.append(("random_bytes", raw_deflate(bytes(rnd.getrandbits(8) for _ in range(50000)))))
    cases.append(("zeros_run", raw_deflate(bytes(300000))))
    cases.append(("period3_run", raw_deflate(b"abc" * 40000)))
    cases.append(("period300", raw_deflate(bytes(rnd.getrandbits(8) for _ in range(300)) * 500)))
    cases.append(("huffman_only", raw_deflate(text(60000, 5), strategy=zlib.Z_HUFFMAN_ONLY)))
    cases.append(("rle", raw_deflate(text(60000, 6), strategy=zlib.Z_RLE)))
    cases.append(("skewed_long_codes", raw_deflate(bytes(min(255, int(rnd.expovariate(0.08))) for _ in range(120000)))))
    # sync-flushed: stored empty blocks between dynamic ones
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = b""
    for i in range(5):
        parts += c.compress(text(20000, 10 + i)) + c.flush(zlib.Z_SYNC_FLUSH)
    parts += c.flush()
    cases.append(("sync_flush_mixed", parts))
    cases.append(("two_stored", stored_block(b"abc", final=False) + stored_block(b"defgh")))
    cases.append(("stored_zero_len_bad_nlen", bytes([1, 0, 0, 0x12, 0x34])))  # quirk q4: LEN==0 skips the NLEN check
    cases.append(("window_far_matches", raw_deflate((text(33000, 20) + text(500, 21)) * 3, level=9)))
    return cases


def malformed_raw_streams():
    good = raw_deflate(text(30000, 30))
    cases = []
    for cut in (1, 2, 5, 17, len(good) // 2, len(good) - 3, len(good) - 1):
        cases.append(("truncated_%d" % cut, good[:cut]))
    cases.append(("btype3", bytes([0x07, 0, 0])))
    cases.append(("stored_bad_nlen", bytes([1, 5, 0, 0, 0, 1, 2, 3, 4, 5])))
    cases.append(("stored_too_long", bytes([1, 10, 0, 0xf5, 0xff, 1, 2, 3])))
    cases.append(("no_final_block", stored_block(b"abc", final=False)))
    cases.append(("empty_input", b""))
    cases.append(("fixed_bad_litlen_286", _fixed_bits([(0b11000110, 8)])))  # code 286
    fixed_then_junk = raw_deflate(b"xyz", strategy=zlib.Z_FIXED) + b"\x55\xaa\x55"
    cases.append(("trailing_bytes_after_final", fixed_then_junk))
    cases.append(("far_distance_first_member", _fixed_far_distance()))
    return cases


def _pack_bits(fields):
    """fields: (value, nbits, msb_first) packed LSB-first like DEFLATE."""
    acc, n, out = 0, 0, bytearray()
    for val, nb, msb in fields:
        if msb:
            val = int(bin(val)[2:].zfill(nb)[::-1], 2)
        acc |= val << n
        n += nb
        while n >= 8:
            out.append(acc & 0xff)
            acc >>= 8
            n -= 8
    if n:
        out.append(acc & 0xff)
    return bytes(out)


def _fixed_bits(codes):
    f = [(1, 1, False), (1, 2, False)] + [(c, nb, True) for c, nb in codes]
    return _pack_bits(f)


def _fixed_far_distance():
    # literal 'a' (0x61 -> code 0x30+0x61, 8 bits), then length 3 (sym 257: 7-bit code 0000001), distance code 4
    # (5 bits) + 1 extra bit -> distance 5/6 > 1 byte of history
    f = [(1, 1, False), (1, 2, False), (0x30 + 0x61, 8, True), (1, 7, True), (4, 5, True), (1, 1, False), (0, 7, True)]
    return _pack_bits(f)


def oversubscribed_dynamic_block(data_bits=(0, 0, 1, 1, 1, 0)):
    """One final dynamic block whose literal/length code is OVER-subscribed: 'a', 'b', 'c' all have length 1 and the
    end-of-block code length 2.  The reference's HuffmanTable does not notice (lib/src/codecs/zlib/_huffman_table.dart:24-45):
    the third one-bit code wraps around and overwrites the first, the two-bit code overwrites half of 'b'.  With the
    default data bits the reference decodes b"ccbb"."""
    f = [(1, 1, False), (2, 2, False), (0, 5, False), (0, 5, False), (14, 4, False)]  # BFINAL, dynamic, HLIT 257, HDIST 1, HCLEN 18
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl_len = {0: 1, 1: 2, 2: 3, 18: 3}
    for sym in order[:18]:
        f.append((cl_len.get(sym, 0), 3, False))
    cl_code = {0: (0b0, 1), 1: (0b10, 2), 2: (0b110, 3), 18: (0b111, 3)}  # canonical, sent MSB first

    def cl(sym):
        c, n = cl_code[sym]
        f.append((c, n, True))

    cl(18); f.append((97 - 11, 7, False))     # symbols 0..96: zero
    cl(1); cl(1); cl(1)                       # 'a', 'b', 'c': length 1 each  (over-subscribed)
    cl(18); f.append((138 - 11, 7, False))    # 100..237
    cl(18); f.append((18 - 11, 7, False))     # 238..255
    cl(2)                                     # 256: length 2
    cl(1)                                     # the one distance code: length 1
    for b in data_bits:
        f.append((b, 1, False))
    return _pack_bits(f) + bytes(4)


def raw_far_reference(lit=b"a", length=3, dist_code=4, dist_extra=1):
    """A fixed-Huffman stream: the literals, then one match whose distance (code 4 + 1 extra bit: 5 or 6) reaches
    further back than the stream's own output -- legal in the reference when earlier gzip members put bytes there (q8)."""
    f = [(1, 1, False), (1, 2, False)]
    for ch in lit:
        f.append((0x30 + ch, 8, True) if ch < 144 else (0x190 + ch - 144, 9, True))
    f += [(length - 2, 7, True), (dist_code, 5, True), (dist_extra, 1, False), (0, 7, True)]
    return _pack_bits(f)


def gz_wrap(raw, data_for_trailer=b""):
    return bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255]) + raw + struct.pack("<II", zlib.crc32(data_for_trailer), len(data_for_trailer))
