"""Dev tool (GPU box): compressed sizes of the HIP Deflate against the oracle (the reference's algorithm) on the test corpora."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import archive_amd
from archive_amd import _native as N
from oracle import pyoracle
from tests import streams
from tools import corpus
assert N.lib().ahip_init(0) == 0
C = {"text12": streams.text(200000, 2), "log": bytes(corpus.text(corpus.LOG, 1234, 0, 1 << 20)), "wiki": bytes(corpus.text(corpus.WIKI, 8, 0, 1 << 19)),
     "log4M": bytes(corpus.text(corpus.LOG, 1234, 0, 4 << 20))}
for name, d in C.items():
    for level in (1, 6, 9):
        ours = archive_amd.Deflate(d, level=level).get_bytes()
        assert zlib.decompress(ours, -15) == d
        ref = len(pyoracle.deflate_raw(d, level)[0])
        print("%-7s L%d  ours %8d  reference %8d  %+6.2f %%" % (name, level, len(ours), ref, 100.0 * (len(ours) / ref - 1)))
