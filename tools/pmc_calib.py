"""Dev tool: known-byte-count kernels for calibrating rocprofv3 memory-side counters on this box.

Run under `rocprofv3 --kernel-trace --pmc <counters>`: a 2 GiB fill (pure 16-B/lane streaming write), a 2 GiB
device-to-device copy (streaming read + write) and a GATHER of 2^26 random 16-byte rows out of a 2 GiB table (the
resolver's access pattern: one far source per token), all far past the 256 MiB Infinity Cache.
tools/prof_summary.py reads their counters to check how FETCH_SIZE / WRITE_SIZE and the TCC_EA0 request counters
translate into bytes for each pattern.
"""
import torch

N = 2 << 30
a = torch.empty(N, dtype=torch.uint8, device="cuda")
b = torch.empty(N, dtype=torch.uint8, device="cuda")
rows = a.view(torch.int64).view(-1, 2)                     # 2^27 rows of 16 bytes
g = torch.Generator(device="cuda"); g.manual_seed(1)
idx = torch.randint(0, rows.shape[0], (1 << 26,), device="cuda", generator=g)
for _ in range(3):
    a.fill_(7)       # vectorized_elementwise_kernel<..FillFunctor<unsigned char>..>: writes N
    b.copy_(a)       # elementwise copy kernel or blit: reads N, writes N
    out = rows.index_select(0, idx)   # index_select kernel: reads 2^26 x 16 B at random (+ 2^26 x 8 B of indices), writes 2^30
torch.cuda.synchronize()
print("calib bytes", N, "gather rows", idx.numel())
