"""Dev tool (GPU box): a wide sweep of damaged bzip2 streams through the HIP decoder against the oracle -- every 37th bit of
a two-block level-1 stream, every 101st bit of a level-9 stream with several tables, every cut of a small stream, random
double flips.  Status, bytes and -- round 6 -- the stream position after `true` and `false` are held against the oracle.  Prints the mismatches (none expected); not part of the test suite (minutes of one-lane inverse transforms).

    python tools/bz_sweep.py [budget seconds]"""
import bz2
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import archive_amd  # noqa: E402
from archive_amd import _native as N  # noqa: E402
from archive_amd import errors  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from tests import streams  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
assert N.lib().ahip_init(0) == 0


def run(buf, verify):
    d = archive_amd.BZip2Decoder()
    try:
        out = d.decode_bytes(buf, verify=verify)
        run.position = d.input_position   # (ahip_last_consumed: where decodeStream leaves its InputStream)
        return d.last_status, out
    except errors.RangeError:
        return 2, None
    except errors.ArchiveHipError as e:
        return (None, None) if "randomised" in str(e) else (-9, str(e).encode())


rnd = random.Random(3)
two = bz2.compress(streams.text(110000, 9) + bytes(3000) + streams.text(2000, 10), 1)
nine = bz2.compress(streams.text(24000, 4) + bytes(rnd.getrandbits(8) for _ in range(6000)) + bytes(500) + streams.text(9000, 6), 9)
small = bz2.compress(streams.text(3000, 7) + b"aaaa\x05" * 30, 1)
cases = []
for bit in range(0, len(two) * 8, 37):
    cases.append(("two", two, (bit,)))
for bit in range(0, len(nine) * 8, 101):
    cases.append(("nine", nine, (bit,)))
for bit in range(0, len(small) * 8, 3):
    cases.append(("small", small, (bit,)))
for _ in range(600):
    s = rnd.choice([two, nine, small])
    cases.append(("double", s, (rnd.randrange(len(s) * 8), rnd.randrange(len(s) * 8))))
rnd.shuffle(cases)
t0 = time.time()
bad = n = 0
hist = {}
for name, s, bits in cases:
    if time.time() - t0 > budget:
        break
    b = bytearray(s)
    for bit in bits:
        b[bit >> 3] ^= 0x80 >> (bit & 7)
    if name == "small" and bits[0] % 5 == 0:
        b = b[:max(1, (bits[0] >> 3))]  # ... and cuts
    b = bytes(b)
    for verify in (False, True):
        got = run(b, verify)
        if got[0] is None:
            continue
        st, out = orc.bzip2_decode(b, verify=verify)
        want = (2, None) if st == 2 else (st, out)
        n += 1
        hist[st] = hist.get(st, 0) + 1
        if got == want and st in (0, 1) and run.position != orc.bzip2_last_position():
            bad += 1
            print("POSITION", name, bits, "verify", verify, "status", st, "got", run.position, "want", orc.bzip2_last_position(), flush=True)
        if got != want:
            bad += 1
            print("MISMATCH", name, bits, "verify", verify, "got", got[0], None if got[1] is None else len(got[1]), "want", want[0],
                  None if want[1] is None else len(want[1]), flush=True)
print("cases %d, mismatches %d, oracle verdicts %s, %.0f s" % (n, bad, sorted(hist.items()), time.time() - t0))
