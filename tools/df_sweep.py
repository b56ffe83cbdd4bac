"""Dev tool (GPU box): random inputs of many shapes and sizes through the HIP Deflate at every level and window size; every
stream must inflate to its input through zlib (with that window) and through the HIP Inflate, and stay within the stored
bound.  Prints the failures (none expected).

    python tools/df_sweep.py [budget seconds]"""
import os
import random
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import archive_amd  # noqa: E402
from archive_amd import _native as N  # noqa: E402
from tests import streams  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
assert N.lib().ahip_init(0) == 0
rnd = random.Random(23)


kind_of = [0]


def make(n):
    k = rnd.randrange(7)
    kind_of[0] = k
    if k == 0:
        return streams.text(n, rnd.randrange(1000))
    if k == 1:
        return bytes(rnd.getrandbits(8) for _ in range(n))
    if k == 2:
        return bytes([rnd.randrange(256)]) * n
    if k == 3:
        p = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
        return (p * (n // len(p) + 1))[:n]
    if k == 4:
        out = bytearray()
        while len(out) < n:
            out += bytes([rnd.randrange(4)]) * rnd.randrange(1, 700)
        return bytes(out[:n])
    if k == 5:
        t = streams.text(max(1, n // 3), rnd.randrange(1000))
        return (t + bytes(rnd.getrandbits(8) for _ in range(n // 3)) + t)[:n]
    return (streams.text(5000, 5) * (n // 5000 + 1))[:n]


t0 = time.time()
n = bad = 0
while time.time() - t0 < budget:
    size = rnd.choice([0, 1, 2, 3, 100, 32767, 32768, 32769, 65536, rnd.randrange(1, 300000), rnd.randrange(1, 5000)])
    data = make(size)
    level = rnd.randrange(10)
    wb = rnd.choice([15, 15, 15, 9, 10, 11, 12, 13, 14])
    comp = archive_amd.Deflate(data, level=level, window_bits=wb).get_bytes() if "window_bits" in archive_amd.Deflate.__init__.__code__.co_varnames else archive_amd.Deflate(data, level=level).get_bytes()
    n += 1
    ok = True
    why = ""
    try:
        if zlib.decompress(comp, -wb) != data:
            ok = False; why = "zlib inflates to other bytes"
    except zlib.error as e:
        ok = False; why = "zlib: %s" % e
        try:
            if zlib.decompress(comp, -15) == data:
                why += " (fine with a 32 KiB window: a match reaches beyond 2^wb - 262)"
        except zlib.error:
            pass
    if ok:
        # (a raw stream that ENDS with its last code makes the reference's reader fail for want of maxCodeLength bits, quirk q2
        #  -- gzip / zlib trailers always follow in practice: four bytes behind the stream stand in for them)
        z = archive_amd.Inflate(comp + bytes(4))
        if not (z.status == 0 and z.get_bytes() == data):
            ok = False; why = "HIP inflate: status %d, %d bytes" % (z.status, len(z.get_bytes()))
    if ok and len(comp) > len(data) + 5 * (len(data) // 32768 + 1) + 16:
        ok = False; why = "larger than stored"
    if not ok:
        bad += 1
        print("FAIL size", size, "level", level, "wb", wb, "comp", len(comp), "kind", kind_of[0], why, flush=True)
        if bad > 10:
            break
print("cases %d, failures %d, %.0f s" % (n, bad, time.time() - t0))
