"""Dev tool: throughput of the CRC-32 / Adler-32 kernels on device-resident data."""
import ctypes, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archive_amd import _native as N
L = N.lib(); L.ahip_init(0)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
rng = np.random.default_rng(1)
buf = rng.integers(0, 256, size=n, dtype=np.uint8)
d = torch.from_numpy(buf).cuda()
out = ctypes.c_uint32()
for name, fn, init, ref in (("crc32", L.ahip_crc32_device, 0, zlib.crc32), ("adler32", L.ahip_adler32_device, 1, zlib.adler32)):
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = fn(d.data_ptr(), n, init, ctypes.byref(out), None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter(); want = ref(buf.tobytes()); cpu = time.perf_counter() - t
    print("%s %d MiB: %.3f ms  %.1f GB/s  ok=%s   (zlib on one host core: %.1f GB/s)" % (name, n >> 20, dt * 1e3, n / dt / 1e9, out.value == want, n / cpu / 1e9))
