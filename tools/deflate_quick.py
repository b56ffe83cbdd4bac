import ctypes, os, sys, time, zlib
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from archive_amd import _native as N
from tools import corpus
L = N.lib(); L.ahip_init(0)
n = 256 << 20
buf = np.empty(n, dtype=np.uint8)
for c in range(n >> 20):
    corpus.lib().corpus_log_text(1234, c * 16, buf[c << 20:].ctypes.data, 1 << 20)
d_in = torch.from_numpy(buf).cuda(); d_out = torch.empty(L.ahip_deflate_bound(n), dtype=torch.uint8, device="cuda"); olen = ctypes.c_size_t()
for lvl in (1, 6, 9):
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = L.ahip_deflate_raw_device(d_in.data_ptr(), n, lvl, 15, d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(os.environ.get("AHIP_LIB"), "level %d, 256 MiB: %.2f ms  %.2f GB/s  ratio %.4f" % (lvl, dt * 1e3, n / dt / 1e9, n / olen.value))
