"""Dev tool: config 4 WITHOUT the BGZF BC subfield through the plan API -- time of ahip_gzip_plan_create (index + sizing run
that keeps its tokens) and of ahip_gzip_plan_run (resolve the kept tokens + the listed members), twice per plan.

    python tools/nobc_time.py 65536        (AHIP_NO_TOKEN_REUSE=1: the two-pass form)
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from archive_amd import _native as N
from tools import corpus
members = int(sys.argv[1])
L = N.lib(); L.ahip_init(0)
cache = "/tmp/ahip_corpus_nobc_%d.npz" % members
if os.path.exists(cache):
    z = np.load(cache); comp, plain = z["comp"], z["plain"]
else:
    comp, plain = corpus.make_gzip(kind=corpus.LOG, seed=1234, n_members=members, bc=False, want_plain=True)
    np.savez(cache, comp=comp, plain=plain)
d_in = torch.from_numpy(comp).cuda(); d_out = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
for it in range(4):
    plan = ctypes.c_void_p()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), None, ctypes.byref(plan)) == 0
    torch.cuda.synchronize(); t1 = time.perf_counter()
    assert L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), None) == 0
    torch.cuda.synchronize(); t2 = time.perf_counter()
    assert L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), None) == 0
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("create %.2f ms  run %.2f ms  run again %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    L.ahip_gzip_plan_destroy(plan)
print("ok", bool(np.array_equal(d_out[:len(plain)].cpu().numpy(), plain)))
