"""Dev tool: one small decode with the AHIP_FLOW_DEBUG build; prints the scheduler state of members whose flow stalled."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from archive_amd import _native as N
from tools import corpus
members = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = N.lib(); L.ahip_init(0)
comp, plain = corpus.make_gzip(n_members=members, want_plain=True)
d_in = torch.from_numpy(comp).cuda(); d_out = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
plan = ctypes.c_void_p()
assert L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), None, ctypes.byref(plan)) == 0
assert L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), None) == 0
torch.cuda.synchronize()
buf = np.zeros(members * 20, dtype=np.uint32); n = ctypes.c_size_t()
assert L.ahip_debug_plan_results(plan, buf.ctypes.data, members, ctypes.byref(n)) == 0
r = buf.reshape(-1, 20)
print("ok=%s" % bool(np.array_equal(d_out[:len(plain)].cpu().numpy(), plain)))
for i in range(min(members, 8)):
    print(i, "status", r[i, 4], "blocks/dbg", hex(r[i, 5]), "epochs", r[i, 6], "mispred", r[i, 7], "fallbacks", r[i, 8], "lost", r[i, 9],
          "V %d retired %d next_fix %d next_spec %d g %d n_items %d faV %s misc %s" % (r[i, 10], r[i, 11], r[i, 12], r[i, 13], r[i, 14], r[i, 15], hex(r[i, 16]), hex(r[i, 17])))
