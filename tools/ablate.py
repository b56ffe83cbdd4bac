"""Dev tool: kernel times of the headline decode for one build of the library (AHIP_LIB=archive_amd/lib/var_<name>.so),
nothing checked -- ablation builds (-DAHIP_ABL_*) write wrong bytes on purpose.  The corpus is cached in /tmp.

    AHIP_LIB=... AHIP_KTIME=1 python tools/ablate.py [members]      (prints the library's own ktime lines)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from archive_amd import _native as N  # noqa: E402
from tools import corpus  # noqa: E402

members = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cache = "/tmp/ablate_%d.bin" % members
if os.path.exists(cache):
    comp = np.fromfile(cache, dtype=np.uint8)
else:
    comp, _ = corpus.make_gzip(n_members=members)
    comp.tofile(cache)
L = N.lib()
assert L.ahip_init(0) == 0
d_in = torch.from_numpy(comp).cuda()
d_out = torch.empty(members * 65536 + 64, dtype=torch.uint8, device="cuda")
for _ in range(3):
    plan = ctypes.c_void_p()
    assert L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), None, ctypes.byref(plan)) == 0
    L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), None)
    torch.cuda.synchronize()
    L.ahip_gzip_plan_destroy(plan)
print("done", os.environ.get("AHIP_LIB", "production"))
