"""Dev tool: decode a synthetic stream once and print the parallel decoder's per-member diagnostics."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from archive_amd import _native as N
from tools import corpus

members = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = corpus.WIKI if (len(sys.argv) > 2 and sys.argv[2] == "wiki") else corpus.LOG
bc = not (len(sys.argv) > 3 and sys.argv[3] == "nobc")
L = N.lib(); L.ahip_init(0)
cache = "/tmp/ahip_corpus_%d_%d_%d.npz" % (members, kind, bc)
if os.path.exists(cache):
    z = np.load(cache); comp, plain = z["comp"], z["plain"]
else:
    comp, plain = corpus.make_gzip(kind=kind, seed=1234 if kind == corpus.LOG else 8, n_members=members, bc=bc, want_plain=True)
    np.savez(cache, comp=comp, plain=plain)
d_in = torch.from_numpy(comp).cuda(); d_out = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
plan = ctypes.c_void_p()
assert L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), None, ctypes.byref(plan)) == 0
for _ in range(2):
    assert L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), None) == 0
torch.cuda.synchronize()
times = []
for _ in range(5):
    d_out.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
ms = sorted(times)[len(times) // 2]
print("plan_run ms: " + " ".join("%.3f" % t for t in times))
buf = np.zeros(members * 20, dtype=np.uint32); n = ctypes.c_size_t()
assert L.ahip_debug_plan_results(plan, buf.ctypes.data, members, ctypes.byref(n)) == 0
r = buf.reshape(-1, 20)
print("kernel %.3f ms  %.1f GB/s out  ok=%s" % (ms, len(plain) / ms / 1e6, bool(np.array_equal(d_out[:len(plain)].cpu().numpy(), plain))))
print("per member: blocks %.2f windows %.2f rounds %.2f (%.3f per window) fallbacks %.3f partial %.2f" % (
    r[:, 5].mean(), r[:, 6].mean(), r[:, 7].mean(), r[:, 7].sum() / max(1, r[:, 6].sum()), r[:, 8].mean(), r[:, 9].mean()))
print("status histogram", np.bincount(r[:, 4]))
print("debug codes", sorted(set(hex(v) for v in r[:, 17] if (v >> 16) == 0xdead)))
cyc = r[:, 10:18].astype(np.float64).mean(axis=0) * 16
if cyc.sum() > 0:
    names = ["tok header+tables", "tok stage", "tok decode steps", "tok scan+assign", "tok retire", "tok header decode", "tok litlen table", "tok serial"]
    tot = cyc.sum()
    if os.environ.get("AHIP_KSTATS_RES"):  # a -DAHIP_PROFILE_RES build: the resolver's phases
        names = ["res look setup", "res gather", "res prep", "res classify+lit+late", "res rounds", "res hard", "res flush", "res whole member"]
        if os.environ.get("AHIP_KSTATS_RES") == "wg":  # the workgroup-per-member resolver: phases summed over its waves, [7] wave 0's whole member
            names = ["wg look setup+barriers", "wg gather+prep", "wg wait for room", "wg literals+early", "wg wait chunk in front", "wg late rounds", "wg publish+flush", "wg whole member (wave 0)"]
        tot = cyc[7]
    print("cycles per member: total %.0f" % tot)
    for nme, c in zip(names, cyc):
        print("  %-18s %10.0f  %5.1f%%" % (nme, c, 100 * c / tot))
got = d_out[:len(plain)].cpu().numpy()
if not np.array_equal(got, plain):
    bad = np.nonzero(got != plain)[0]
    print("mismatches:", len(bad), "first at", bad[:10], "member", bad[0] // 65536, "offset", bad[0] % 65536)
    i = bad[0]
    print("got ", bytes(got[i - 20:i + 40]))
    print("want", bytes(plain[i - 20:i + 40]))
    print("mismatch offsets mod 64 histogram:", np.bincount((bad % 65536) % 64, minlength=64))
