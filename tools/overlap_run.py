"""Dev experiment: do the two inflate kernels overlap when two decodes run side by side on two HIP streams?

The tokenizer is bound by VALU issue (71 %), the resolver by L2 misses and LDS cycles: run back to back each leaves the
other's resource idle.  Two host threads, each with its own stream, its own copy of the stream and its own output, decode
in a loop; AHIP_TOK_WGS_PER_CU / AHIP_RES_WGS_PER_CU cap the waves each kernel puts on a CU so that both fit side by side.

    AHIP_TOK_WGS_PER_CU=6 AHIP_RES_WGS_PER_CU=9 python tools/overlap_run.py [threads] [members] [steps]"""
import ctypes
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from archive_amd import _native as N  # noqa: E402
from tools import corpus  # noqa: E402


def main():
    nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    members = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    cache = "/tmp/ablate_%d.bin" % members
    if os.path.exists(cache):
        comp = np.fromfile(cache, dtype=np.uint8)
    else:
        comp, _ = corpus.make_gzip(n_members=members)
        comp.tofile(cache)
    L = N.lib()
    assert L.ahip_init(0) == 0
    out_bytes = members * 65536
    bar = threading.Barrier(nthreads)
    times = [0.0] * nthreads


    def work(t):
        stream = torch.cuda.Stream()
        sh = ctypes.c_void_p(stream.cuda_stream)
        with torch.cuda.stream(stream):
            d_in = torch.from_numpy(comp).cuda()
            d_out = torch.empty(out_bytes + 64, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()

        def step():
            plan = ctypes.c_void_p()
            assert L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), sh, ctypes.byref(plan)) == 0
            assert L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), sh) == 0
            olen = ctypes.c_size_t()
            rc = L.ahip_gzip_plan_status(plan, ctypes.byref(olen))
            L.ahip_gzip_plan_destroy(plan)
            assert rc == 0 and olen.value == out_bytes, (rc, olen.value)
        step()
        step()
        bar.wait()
        if t == 1:
            time.sleep(0.004)  # half a decode out of phase: one thread's tokenizer under the other's resolver
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        stream.synchronize()
        times[t] = time.perf_counter() - t0


    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    tot = max(times)
    print("threads %d members %d steps %d: %.2f ms per decode and thread, aggregate %.1f GB/s out (caps tok %s res %s)" % (
        nthreads, members, steps, tot / steps * 1e3, nthreads * steps * out_bytes / tot / 1e9,
        os.environ.get("AHIP_TOK_WGS_PER_CU", "-"), os.environ.get("AHIP_RES_WGS_PER_CU", "-")))


if __name__ == "__main__":
    main()
