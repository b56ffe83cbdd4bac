"""Dev tool: bzip2 decode throughput (config 5: N x 900k blocks of wiki-like text) through the host-pointer API."""
import bz2, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import archive_amd
from archive_amd import _native as N
from oracle import pyoracle as orc
from tools import corpus
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N.lib().ahip_init(0)
data = bytes(corpus.text(corpus.WIKI, 8, 0, mb << 20))
t = time.time(); comp = bz2.compress(data, 9); print("bz2 -9: %d -> %d bytes (%.1f s on the host)" % (len(data), len(comp), time.time() - t))
dec = archive_amd.BZip2Decoder()
for it in range(3):
    t = time.perf_counter(); out = dec.decode_bytes(comp, verify=True); dt = time.perf_counter() - t
    print("GPU decode: status %d, %.1f ms, %.3f GB/s out (host buffers, PCIe included), ok=%s" % (dec.last_status, dt * 1e3, len(out) / dt / 1e9, out == data))
t = time.perf_counter(); st, o = orc.bzip2_decode(comp, verify=True, cap=len(data) + 64); dt = time.perf_counter() - t
print("CPU oracle (1 thread): %.3f GB/s" % (len(o) / dt / 1e9))
