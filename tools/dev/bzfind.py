import bz2, sys
sys.path.insert(0, '/root/repo')
import archive_amd
from archive_amd import _native as N, errors
from oracle import pyoracle as orc
from tests import streams
assert N.lib().ahip_init(0) == 0
data = streams.text(110000, 9) + bytes(3000) + streams.text(2000, 10)
c = bz2.compress(data, 1)
def run(buf, verify):
    d = archive_amd.BZip2Decoder()
    try:
        out = d.decode_bytes(buf, verify=verify)
        return d.last_status, out
    except errors.RangeError:
        return 2, None
    except errors.ArchiveHipError as e:
        return None
bad = 0
for bit in range(0, len(c) * 8, 37):
    buf = bytearray(c); buf[bit >> 3] ^= 0x80 >> (bit & 7); buf = bytes(buf)
    for verify in (False, True):
        got = run(buf, verify)
        if got is None: continue
        st, out = orc.bzip2_decode(buf, verify=verify)
        want = (2, None) if st == 2 else (st, out)
        if got != want:
            bad += 1
            print("bit", bit, "verify", verify, "got", got[0], None if got[1] is None else len(got[1]), "want", want[0], None if want[1] is None else len(want[1]))
            if bad > 12: sys.exit(0)
print("mismatches", bad)
