"""ctypes loader for the CPU oracle (oracle/*_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (archive_amd) never does.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

ORC_OK, ORC_FALSE, ORC_RANGE, ORC_HANG, ORC_CAP = 0, 1, 2, 3, -1


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith("_oracle.c")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        u8p, szp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)
        L.orc_inflate_raw.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, szp, szp]
        L.orc_gzip_decode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_size_t, szp]
        L.orc_zlib_decode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_size_t, szp]
        L.orc_crc32.argtypes = [u8p, ctypes.c_size_t, ctypes.c_uint32]
        L.orc_crc32.restype = ctypes.c_uint32
        L.orc_adler32.argtypes = [u8p, ctypes.c_size_t, ctypes.c_uint32]
        L.orc_adler32.restype = ctypes.c_uint32
        L.orc_bzip2_decode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_size_t, szp]
        L.orc_last_position.argtypes = []
        L.orc_last_position.restype = ctypes.c_size_t
        L.orc_bzip2_block.argtypes = [u8p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int, u8p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), szp,
                                      ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int)]
        L.orc_deflate_raw.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_size_t, szp,
                                      ctypes.POINTER(ctypes.c_uint32)]
        L.orc_gzip_encode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, u8p, ctypes.c_size_t, szp]
        L.orc_zlib_encode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_size_t, szp]
        _lib = L
    return _lib


def _inbuf(data):
    data = bytes(data)
    return (ctypes.c_char * max(1, len(data))).from_buffer_copy(data or b"\0"), len(data)


def _run(fn, data, pre, cap):
    buf, n = _inbuf(data)
    cap = cap if cap is not None else max(1 << 16, 8 * n)
    while True:
        out = ctypes.create_string_buffer(cap)
        olen = ctypes.c_size_t(0)
        st, extra = fn(buf, n, out, cap, olen, *pre)
        if st != ORC_CAP:
            return st, out.raw[:olen.value], extra
        cap *= 4


def inflate_raw(data, cap=None):
    """Inflate(bytes).getBytes() -> (status, output, input_position)"""
    def fn(buf, n, out, cap, olen):
        pos = ctypes.c_size_t(0)
        st = lib().orc_inflate_raw(ctypes.addressof(buf), n, ctypes.addressof(out), cap, ctypes.byref(olen), ctypes.byref(pos))
        return st, pos.value
    return _run(fn, data, (), cap)


def gzip_decode(data, verify=False, raw=False, cap=None):
    """GZipDecoderWeb().decodeBytes -> (status, output)"""
    def fn(buf, n, out, cap, olen):
        return lib().orc_gzip_decode(ctypes.addressof(buf), n, int(verify), int(raw), ctypes.addressof(out), cap, ctypes.byref(olen)), None
    st, o, _ = _run(fn, data, (), cap)
    return st, o


def zlib_decode(data, verify=False, raw=False, cap=None):
    """ZLibDecoderWeb().decodeBytes -> (status, output)"""
    def fn(buf, n, out, cap, olen):
        return lib().orc_zlib_decode(ctypes.addressof(buf), n, int(verify), int(raw), ctypes.addressof(out), cap, ctypes.byref(olen)), None
    st, o, _ = _run(fn, data, (), cap)
    return st, o


def last_position():
    """where the calling thread's last gzip_decode / zlib_decode left the reference's InputStream"""
    return lib().orc_last_position()


def crc32(data, crc=0):
    buf, n = _inbuf(data)
    return lib().orc_crc32(ctypes.addressof(buf), n, crc)


def adler32(data, adler=1):
    buf, n = _inbuf(data)
    return lib().orc_adler32(ctypes.addressof(buf), n, adler)


def deflate_raw(data, level=6, truncate_heuristic=True, window_bits=15):
    """Deflate(bytes, level: L, windowBits: W).getBytes() -> (compressed bytes, crc32).  truncate_heuristic=False
    switches off the reference's block-truncation heuristic (= stock zlib behaviour)."""
    buf, n = _inbuf(data)
    cap = n + n // 8 + 1024
    out = ctypes.create_string_buffer(cap)
    olen = ctypes.c_size_t(0)
    crc = ctypes.c_uint32(0)
    L = lib()
    L.orc_deflate_raw_wb.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32)]
    L.orc_deflate_raw_wb.restype = ctypes.c_int
    rc = L.orc_deflate_raw_wb(ctypes.addressof(buf), n, level, window_bits, int(truncate_heuristic), ctypes.addressof(out), cap,
                              ctypes.byref(olen), ctypes.byref(crc))
    assert rc == 0
    return out.raw[:olen.value], crc.value


def gzip_encode(data, level=6, mtime=0):
    buf, n = _inbuf(data)
    cap = n + n // 8 + 1024
    out = ctypes.create_string_buffer(cap)
    olen = ctypes.c_size_t(0)
    assert lib().orc_gzip_encode(ctypes.addressof(buf), n, level, mtime, ctypes.addressof(out), cap, ctypes.byref(olen)) == 0
    return out.raw[:olen.value]


def zlib_encode(data, level=6):
    buf, n = _inbuf(data)
    cap = n + n // 8 + 1024
    out = ctypes.create_string_buffer(cap)
    olen = ctypes.c_size_t(0)
    assert lib().orc_zlib_encode(ctypes.addressof(buf), n, level, ctypes.addressof(out), cap, ctypes.byref(olen)) == 0
    return out.raw[:olen.value]


def bzip2_decode(data, verify=False, cap=None):
    """BZip2Decoder().decodeBytes -> (status, output)"""
    def fn(buf, n, out, cap, olen):
        return lib().orc_bzip2_decode(ctypes.addressof(buf), n, int(verify), ctypes.addressof(out), cap, ctypes.byref(olen)), None
    st, o, _ = _run(fn, data, (), cap if cap is not None else max(1 << 16, 64 * len(bytes(data))))
    return st, o


def bzip2_last_position():
    """where BZip2Decoder.decodeStream left its InputStream in the last bzip2_decode() of this thread"""
    L = lib()
    L.orc_bzip2_last_position.argtypes = []
    L.orc_bzip2_last_position.restype = ctypes.c_size_t
    return L.orc_bzip2_last_position()


def bzip2_block(data, bit, level, cap=None):
    """ONE candidate block of a bzip2 stream, read from bit position `bit` the way decodeStream reads it (block type,
    stored CRC, _readCompressed) -> dict(status, kind, end_bit, out, crc, stored).  status 17 = randomised flag set."""
    buf, n = _inbuf(data)
    cap = cap if cap is not None else level * 100000 * 52 + 1024
    out = ctypes.create_string_buffer(cap)
    end, olen = ctypes.c_uint64(0), ctypes.c_size_t(0)
    crc, stored, kind = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_int(0)
    st = lib().orc_bzip2_block(ctypes.addressof(buf), n, bit, level, ctypes.addressof(out), cap, ctypes.byref(end), ctypes.byref(olen),
                               ctypes.byref(crc), ctypes.byref(stored), ctypes.byref(kind))
    return dict(status=st, kind=kind.value, end_bit=end.value, out=out.raw[:olen.value], crc=crc.value, stored=stored.value)


def bzip2_block_bits(data):
    """Bit positions of the blocks of a VALID stream, the end-of-stream marker's last."""
    level = data[3] - 0x30
    bits = [32]
    while True:
        r = bzip2_block(data, bits[-1], level)
        assert r["status"] == 0, r["status"]
        if r["kind"] == 2:
            return bits
        bits.append(r["end_bit"])
