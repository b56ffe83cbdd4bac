"""Dev tool: known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this box.

Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (or WRITE_SIZE): a 2 GiB fill (pure 16-B/lane
streaming write) and a 2 GiB device-to-device copy (streaming read + write), both far past the 256 MiB
Infinity Cache.  tools/prof_summary.py reads their counters to derive the read / write scale factors.
"""
import torch

N = 2 << 30
a = torch.empty(N, dtype=torch.uint8, device="cuda")
b = torch.empty(N, dtype=torch.uint8, device="cuda")
for _ in range(3):
    a.fill_(7)       # vectorized_elementwise_kernel<..FillFunctor<unsigned char>..>: writes N
    b.copy_(a)       # elementwise copy kernel or blit: reads N, writes N
torch.cuda.synchronize()
print("calib bytes", N)
