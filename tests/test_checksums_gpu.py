"""CRC-32 / Adler-32 kernels (archive_amd/csrc/checksum_kernels.hpp) against zlib and the reference's
known-answer vectors (test/crc32_test.dart, test/adler32_test.dart via tests/golden/manifest.json)."""
import ctypes
import json
import os
import random
import zlib

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _dev(data):
    import torch
    pad = 3  # start the buffer at an odd address: the kernels must not assume alignment
    t = torch.empty(len(data) + pad + 8, dtype=torch.uint8, device="cuda")
    if data:
        t[pad:pad + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    return t, t.data_ptr() + pad


def _crc(L, data, init=0):
    t, p = _dev(data)
    out = ctypes.c_uint32()
    assert L.ahip_crc32_device(p, len(data), init, ctypes.byref(out), None) == 0
    return out.value


def _adler(L, data, init=1):
    t, p = _dev(data)
    out = ctypes.c_uint32()
    assert L.ahip_adler32_device(p, len(data), init, ctypes.byref(out), None) == 0
    return out.value


def test_known_answers(native_built):
    from archive_amd import _native as N
    L = N.lib()
    assert L.ahip_init(0) == 0
    kats = json.load(open(os.path.join(HERE, "golden", "manifest.json")))["checksum_kat"]

    def expand(spec):  # "hex" or "hex*repeat"
        h, _, rep = spec.partition("*")
        return bytes.fromhex(h) * (int(rep) if rep else 1)
    for spec, want in kats["crc32"]:
        assert _crc(L, expand(spec)) == int(want, 16), spec
    for spec, want in kats["adler32"]:
        assert _adler(L, expand(spec)) == int(want, 16), spec


def test_sizes_alignments_and_chaining(native_built):
    from archive_amd import _native as N
    L = N.lib()
    assert L.ahip_init(0) == 0
    rnd = random.Random(5)
    sizes = [0, 1, 3, 4, 5, 255, 256, 257, 1023, 4096, 65535, 65536, 65537, 65536 * 3 + 700, 1 << 20, (1 << 22) + 12345,
             40 * 65536 * 4 + 99]
    for n in sizes:
        data = rnd.randbytes(n) if n < (1 << 21) else (rnd.randbytes(1 << 16) * ((n >> 16) + 1))[:n]
        assert _crc(L, data) == zlib.crc32(data), n
        assert _adler(L, data) == zlib.adler32(data), n
    # chaining: the second call continues the first
    a, b = rnd.randbytes(100001), rnd.randbytes(77777)
    assert _crc(L, b, _crc(L, a)) == zlib.crc32(a + b)
    assert _adler(L, b, _adler(L, a)) == zlib.adler32(a + b)
    # worst case for the Adler sums: all 0xff
    ff = b"\xff" * (3 << 20)
    assert _adler(L, ff) == zlib.adler32(ff) and _crc(L, ff) == zlib.crc32(ff)
