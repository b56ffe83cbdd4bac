import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from archive_amd import _native as N
from tools import corpus
L = N.lib(); L.ahip_init(0)
comp, plain = corpus.make_gzip(kind=corpus.LOG, seed=1234, n_members=2048, bc=False, want_plain=True)
d_in = torch.from_numpy(comp).cuda(); d_out = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
olen = ctypes.c_size_t()
rc = L.ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None)
print("rc", rc, olen.value, bool(np.array_equal(d_out[:len(plain)].cpu().numpy(), plain)))
buf = np.zeros(1, dtype=np.uint32)
