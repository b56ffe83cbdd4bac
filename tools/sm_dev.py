"""Dev tool (GPU box): device-resident decode of ONE long gzip member, timed: python tools/sm_dev.py <MiB> [wiki]"""
import ctypes, gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from archive_amd import _native as N
from tools import corpus
N.lib().ahip_init(0)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = corpus.WIKI if (len(sys.argv) > 2 and sys.argv[2] == "wiki") else corpus.LOG
data = bytes(corpus.text(kind, 8, 0, mb << 20))
gz = gzip.compress(data, 6)
d_in = torch.frombuffer(bytearray(gz), dtype=torch.uint8).cuda(); d_out = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda"); olen = ctypes.c_size_t()
for it in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = N.lib().ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("gzip_decode_device: rc %d %.2f ms  %.2f GB/s out" % (rc, dt * 1e3, olen.value / dt / 1e9), flush=True)
print("device bytes ok:", bytes(d_out[:olen.value].cpu().numpy()) == data)
