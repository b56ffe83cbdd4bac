"""Dev tool (GPU box): damaged multi-member gzip / zlib / raw streams through the HIP decoders against the oracle -- status,
bytes and the InputStream position decodeStream leaves.  Shapes the test suite has less of: members WITHOUT the BC subfield
next to members with it, a stored member, a long member (chunked path), damage that hits two members at once.
Prints the mismatches (none expected).

    python tools/gz_sweep.py [budget seconds]"""
import os
import random
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import archive_amd  # noqa: E402
from archive_amd import _native as N  # noqa: E402
from archive_amd import errors  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from tests import streams  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
assert N.lib().ahip_init(0) == 0
rnd = random.Random(17)


def gz(buf, **kw):
    d = archive_amd.GZipDecoder()
    try:
        out = d.decode_bytes(buf, **kw)
        return d.last_status, out, d.input_position
    except errors.RangeError:
        return 2, None, None
    except errors.ReferenceWouldHang:
        return 3, None, None


def zl(buf, **kw):
    d = archive_amd.ZLibDecoder()
    try:
        out = d.decode_bytes(buf, **kw)
        return d.last_status, out, d.input_position
    except errors.RangeError:
        return 2, None, None
    except errors.ReferenceWouldHang:
        return 3, None, None


a, b, c = streams.text(9000, 21), streams.text(70000, 22), bytes(rnd.getrandbits(8) for _ in range(3000))
base_gz = [
    streams.bgzf_member(a) + streams.gz_member(b) + streams.bgzf_member(c) + streams.gz_member(a, level=1),
    streams.gz_member(a, name=b"n", comment=b"c") + streams.gz_member(c, level=0) + streams.gz_member(b, level=9),
    streams.gz_member(a) * 5,
]
base_zl = [zlib.compress(a) + zlib.compress(b, 1), zlib.compress(c, 0) + zlib.compress(a, 9)]
t0 = time.time()
n = bad = 0
hist = {}
while time.time() - t0 < budget:
    kind = rnd.randrange(3)
    src = rnd.choice(base_gz if kind < 2 else base_zl)
    buf = bytearray(src)
    for _ in range(rnd.choice([1, 1, 1, 2, 3])):
        bit = rnd.randrange(len(buf) * 8)
        buf[bit >> 3] ^= 1 << (bit & 7)
    if rnd.randrange(5) == 0:
        buf = buf[:rnd.randrange(1, len(buf))]
    buf = bytes(buf)
    verify = bool(rnd.randrange(2))
    if kind < 2:
        got = gz(buf, verify=verify)
        st, out = orc.gzip_decode(buf, verify=verify)
    else:
        got = zl(buf, verify=verify)
        st, out = orc.zlib_decode(buf, verify=verify)
    pos = orc.last_position()
    want = (st, None, None) if st in (2, 3) else (st, out, pos)
    n += 1
    hist[st] = hist.get(st, 0) + 1
    if got != want:
        bad += 1
        print("MISMATCH kind", kind, "verify", verify, "got", got[0], None if got[1] is None else len(got[1]), got[2], "want", want[0],
              None if want[1] is None else len(want[1]), want[2], flush=True)
        if bad > 20:
            break
print("cases %d, mismatches %d, oracle verdicts %s, %.0f s" % (n, bad, sorted(hist.items()), time.time() - t0))
