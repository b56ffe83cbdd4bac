"""Measurement harness (lives under tests/ because it times the CPU oracle beside the GPU path): bzip2 decode throughput (config 5: N x 900k blocks of wiki-like text) through the host-pointer API."""
import bz2, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
# device-resident: input already in HBM, output stays there (the figure DESIGN.md quotes)
import ctypes, torch
d_in = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
d_out = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
olen = ctypes.c_size_t()
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = N.lib().ahip_bzip2_decode_device(d_in.data_ptr(), d_in.numel(), 1, d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("GPU decode, device-resident: status %d, %.1f ms, %.3f GB/s out, (C+U)/t = %.3f GB/s" % (rc, dt * 1e3, olen.value / dt / 1e9, (olen.value + len(comp)) / dt / 1e9))
print("ok=%s" % (bytes(d_out[:olen.value].cpu().numpy()) == data))
t = time.perf_counter(); st, o = orc.bzip2_decode(comp, verify=True, cap=len(data) + 64); dt = time.perf_counter() - t
print("CPU oracle (1 thread): %.3f GB/s" % (len(o) / dt / 1e9))
