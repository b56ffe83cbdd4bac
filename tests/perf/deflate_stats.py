"""Measurement harness (lives under tests/ because it times the CPU oracle beside the GPU path): compressed size of the HIP Deflate vs the reference oracle / zlib, and device throughput."""
import ctypes, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import archive_amd
from archive_amd import _native as N
from oracle import pyoracle as orc
from tools import corpus
from tests import streams
L = N.lib(); L.ahip_init(0)
cs = {"text": streams.text(200000, 2), "log": bytes(corpus.text(corpus.LOG, 1234, 0, 1 << 20)), "wiki": bytes(corpus.text(corpus.WIKI, 8, 0, 1 << 20))}
for name, d in cs.items():
    for lvl in (1, 6, 9):
        ours = len(archive_amd.Deflate(d, level=lvl).get_bytes()); ref = len(orc.deflate_raw(d, lvl)[0])
        print("%-5s L%d ours %8d ref %8d  %+.1f%%  ratio %.3f vs %.3f" % (name, lvl, ours, ref, 100.0 * (ours - ref) / ref, len(d) / ours, len(d) / ref))
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mb << 20
buf = np.empty(n, dtype=np.uint8)
for c in range(n >> 20):
    corpus.lib().corpus_log_text(1234, c * 16, buf[c << 20:].ctypes.data, 1 << 20)
d_in = torch.from_numpy(buf).cuda(); d_out = torch.empty(L.ahip_deflate_bound(n), dtype=torch.uint8, device="cuda"); olen = ctypes.c_size_t()
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = L.ahip_deflate_raw_device(d_in.data_ptr(), n, 6, 15, d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("deflate L6 %d MiB: rc %d  %.1f ms  %.2f GB/s in  ratio %.3f" % (mb, rc, dt * 1e3, n / dt / 1e9, n / max(1, olen.value)))
comp = d_out[:olen.value].cpu().numpy().tobytes()
print("round trip:", zlib.decompress(comp, -15) == buf.tobytes())
