"""Dev tool (GPU box): device-resident decode time of ONE long gzip member (config 2a) -- the knobs AHIP_SM_CHUNK / AHIP_SM_SPLIT /
AHIP_SM_TBLOCKS are read from the environment by the library:  python tools/sm_time.py [MiB] [wiki|log]"""
import ctypes, gzip, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from archive_amd import _native as N
from tools import corpus
N.lib().ahip_init(0)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = corpus.LOG if (len(sys.argv) > 2 and sys.argv[2] == "log") else corpus.WIKI
cache = "/tmp/sm_time_%d_%d.gz" % (mb, kind)
if os.path.exists(cache):
    gz = open(cache, "rb").read()
else:
    gz = gzip.compress(bytes(corpus.text(kind, 8, 0, mb << 20)), 6)
    open(cache, "wb").write(gz)
d_in = torch.frombuffer(bytearray(gz), dtype=torch.uint8).cuda()
d_out = torch.empty((mb << 20) + 64, dtype=torch.uint8, device="cuda")
olen = ctypes.c_size_t()
ts = []
for it in range(7):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = N.lib().ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
ts = sorted(ts[2:])
crc = zlib.crc32(bytes(d_out[:olen.value].cpu().numpy()))
print("chunk %s split %s: rc %d  median %.2f ms  %.2f GB/s out  (chunks %d, crc %08x)" % (os.environ.get("AHIP_SM_CHUNK", "-"), os.environ.get("AHIP_SM_SPLIT", "-"),
      rc, ts[len(ts) // 2] * 1e3, olen.value / ts[len(ts) // 2] / 1e9, N.lib().ahip_debug_last_chunks(), crc))
