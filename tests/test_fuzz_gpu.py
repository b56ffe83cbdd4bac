"""Damaged streams, many at a time: verdict and bytes of every mutant equal the CPU oracle's.

The mutants go through `ahip_inflate_batch` in one call, so the wave-parallel decoder, its serial fallback and
the per-stream input bound are all on the path; the oracle is the restated reference (oracle/inflate_oracle.c)."""
import ctypes
import random
import zlib

import pytest

from tests import streams

pytestmark = pytest.mark.gpu


def _mutants(seed, n):
    rnd = random.Random(seed)
    bases = [streams.raw_deflate(streams.text(40000, 40)),                         # one long dynamic block
             streams.raw_deflate(streams.text(12000, 41), level=1),
             streams.raw_deflate(streams.text(9000, 42), strategy=zlib.Z_FIXED),
             streams.raw_deflate(bytes(rnd.getrandbits(8) for _ in range(6000))),  # nearly incompressible
             streams.raw_deflate(streams.text(5000, 43), level=0) + b""]           # stored
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    multi = b"".join(c.compress(streams.text(15000, 50 + i)) + c.flush(zlib.Z_FULL_FLUSH) for i in range(4)) + c.flush()
    bases.append(multi)                                                            # several blocks + empty stored ones
    out = []
    for k in range(n):
        b = bytearray(bases[k % len(bases)])
        kind = rnd.randrange(5)
        if kind == 0:      # single bit flip
            p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
        elif kind == 1:    # a few flips
            for _ in range(rnd.randrange(2, 6)):
                p = rnd.randrange(len(b)); b[p] ^= 1 << rnd.randrange(8)
        elif kind == 2:    # truncation
            b = b[:rnd.randrange(1, len(b))]
        elif kind == 3:    # overwritten bytes
            p = rnd.randrange(len(b)); b[p:p + rnd.randrange(1, 9)] = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 9)))
        else:              # intact, with trailing junk
            b += bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0, 7)))
        out.append(bytes(b))
    return out


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_mutants_match_oracle(native_built, seed):
    from archive_amd import _native as N
    from oracle import pyoracle as orc
    L = N.lib()
    assert L.ahip_init(0) == 0
    muts = _mutants(seed, 360)
    blob = b"".join(muts)
    k = len(muts)
    u64s = ctypes.c_uint64 * k
    offs, o = [], 0
    for m in muts:
        offs.append(o); o += len(m)
    in_off, in_size = u64s(*offs), u64s(*[len(m) for m in muts])
    out_off, out_len, status, total = u64s(), u64s(), (ctypes.c_int32 * k)(), ctypes.c_size_t()
    rc = L.ahip_inflate_batch(blob, len(blob), k, in_off, in_size, None, None, 0, out_off, out_len, status, ctypes.byref(total))
    assert rc in (0, N.AHIP_E_CAP), N.last_error()
    obuf = ctypes.create_string_buffer(max(1, total.value))
    assert L.ahip_inflate_batch(blob, len(blob), k, in_off, in_size, None, obuf, total.value, out_off, out_len, status,
                                ctypes.byref(total)) == 0, N.last_error()
    raw = obuf.raw
    bad = []
    for i, m in enumerate(muts):
        ost, oout, _ = orc.inflate_raw(m, cap=1 << 20)
        got = raw[out_off[i]:out_off[i] + out_len[i]]
        if ost == 2:
            ok = status[i] == N.AHIP_RANGE
        elif ost == 3:
            ok = status[i] == N.AHIP_HANG
        else:
            ok = status[i] == ost and got == oout
        if not ok:
            bad.append((i, i % 6, ost, status[i], len(oout), len(got)))
    assert not bad, bad[:10]
