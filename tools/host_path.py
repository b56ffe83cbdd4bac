"""Dev tool (GPU box): the host-pointer gzip entry point end to end on a synthetic BGZF stream: python tools/host_path.py <members>"""
import ctypes, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from archive_amd import _native as N
from tools import corpus
members = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = N.lib(); assert L.ahip_init(0) == 0
t = time.time(); comp, plain = corpus.make_gzip(n_members=members, want_plain=True); print("gen %.1fs" % (time.time() - t), flush=True)
out = np.empty(len(plain) + 64, dtype=np.uint8); olen = ctypes.c_size_t()
for it in range(3):
    t = time.perf_counter()
    rc = L.ahip_gzip_decode(comp.ctypes.data, len(comp), 0, 0, out.ctypes.data, len(out), ctypes.byref(olen))
    dt = time.perf_counter() - t
    print("rc %d  %d bytes  %.1f ms  %.2f GB/s out  shards %d  ok=%s" % (rc, olen.value, dt * 1e3, olen.value / dt / 1e9, L.ahip_debug_last_shards(),
          bool(np.array_equal(out[:len(plain)], plain))), flush=True)
print("done", flush=True)
