#!/usr/bin/env python3
"""bench.py -- headline benchmark: Inflate GB/s (uncompressed out) on a multi-member gzip stream.

One "step" = one complete decode of the rank's device-resident stream: member index build
(candidate scan, header parse, chain), the inflate kernel, verification, and -- for N > 1 -- the
one collective the path has: an all-gather of per-rank output sizes (RCCL) whose exclusive scan
is each shard's offset in the logical concatenated output.  Inputs and outputs stay in HBM.

Workload (config.workload): BASELINE.json configs[3] -- 65 536 gzip members x 64 KiB of
synthetic log text (4 GiB out, ~1.7 GiB in) PER GPU (weak scaling: rank r holds members
[r*M, (r+1)*M) of a stream of N*M members).  --members shrinks it for quick runs.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--members", type=int, default=65536, help="gzip members per GPU")
    ap.add_argument("--member-bytes", type=int, default=65536)
    ap.add_argument("--kind", default="log", choices=["log", "wiki"])
    ap.add_argument("--no-bc", action="store_true", help="omit the BGZF BC subfield (forces sizing runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--check", action="store_true", help="verify the decoded bytes against the generator's plain text")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`" % (args.gpus, args.gpus),
                  file=sys.stderr)
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from archive_amd import _native as N
    from tools import corpus
    L = N.lib()
    if L.ahip_init(local_rank) != 0:
        raise SystemExit("ahip_init failed: " + N.last_error())

    # ---- synthetic workload, generated on the host and made resident in HBM before timing ----
    kind = corpus.LOG if args.kind == "log" else corpus.WIKI
    seed = 1234 if kind == corpus.LOG else 8
    threads = max(1, (os.cpu_count() or 1) // max(1, local_world))
    t0 = time.time()
    comp, plain = corpus.make_gzip(kind=kind, seed=seed, n_members=args.members, member_bytes=args.member_bytes,
                                   level=6, bc=not args.no_bc, threads=threads, first_chunk=rank * args.members,
                                   want_plain=args.check)
    gen_s = time.time() - t0
    out_bytes = args.members * args.member_bytes
    d_in = torch.from_numpy(comp).to(dev)
    d_out = torch.empty(out_bytes + 64, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    sh = ctypes.c_void_p(stream.cuda_stream)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        plan = ctypes.c_void_p()
        rc = L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), sh, ctypes.byref(plan))
        if rc != 0:
            raise SystemExit("plan_create: %d %s" % (rc, N.last_error()))
        if i is not None:
            ev0[i].record(stream)
        rc = L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), sh)
        if i is not None:
            ev1[i].record(stream)
        if rc != 0:
            raise SystemExit("plan_run: %d %s" % (rc, N.last_error()))
        olen = ctypes.c_size_t()
        rc = L.ahip_gzip_plan_status(plan, ctypes.byref(olen))
        L.ahip_gzip_plan_destroy(plan)
        if rc != 0 or olen.value != out_bytes:
            raise SystemExit("decode verdict %d, %d bytes (expected %d): %s" % (rc, olen.value, out_bytes, N.last_error()))
        if world > 1:  # the path's one exchange: output-size all-gather -> shard offsets (RCCL)
            from archive_amd.sharding import exchange_output_offsets
            exchange_output_offsets(olen.value, device=dev)
        return olen.value

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(out_bytes)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_out = float(tot.item())

    if args.check:
        got = d_out[:out_bytes].cpu().numpy()
        if not np.array_equal(got, plain):
            raise SystemExit("decoded bytes differ from the generator's plain text")

    if rank == 0:
        kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps
        algo_bytes = float(d_in.numel() + out_bytes)  # C + U: compressed read once + output written once
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        value = total_out * args.steps / elapsed / 1e9
        line = {
            "metric": "Inflate GB/s (uncompressed out) on multi-member gzip",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d gzip members x %d B %s text per GPU (%s), zlib level 6" % (
                args.members, args.member_bytes, args.kind, "BGZF BC subfield" if not args.no_bc else "no BC"),
                "members_per_gpu": args.members, "member_bytes": args.member_bytes,
                "compressed_bytes_per_gpu": int(d_in.numel()), "ratio": round(out_bytes / d_in.numel(), 4),
                "sharding": "members, one process per GPU" if world > 1 else "single GPU",
                "gen_seconds": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None,
                         "kernel": "inflate_tokenize_kernel + inflate_resolve_kernel (one launch each per decode; "
                                   "HIP events on the launch stream around ahip_gzip_plan_run)",
                         "kernel_ms": round(kern_ms, 4), "algorithmic_bytes": int(algo_bytes)},
        }
        # HBM-side traffic of the inflate stage: PMC counters cannot be read from inside this
        # process; the figure measured by rocprofv3 (separate --pmc passes, profiles/) is attached
        # when this run uses the profiled workload.
        try:
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))[-1]
            with open(latest) as f:
                pmc = json.load(f)
            w = pmc["workload"]
            if (w["members"], w["member_bytes"], w["kind"], w["bc"]) == (args.members, args.member_bytes, args.kind, not args.no_bc):
                line["roofline"]["traffic"] = round(pmc["traffic_bytes_per_launch"] / 1e9, 2)
                line["roofline"]["traffic_unit"] = ("GB per decode (rocprofv3 FETCH_SIZE+WRITE_SIZE, calibrated, profiles/%s)" % os.path.basename(latest))
        except Exception:
            pass
        if args.cpu_seconds > 0 and world == 1:  # the CPU baseline is reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(comp, args, out_bytes)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(comp, args, out_bytes):
    """The CPU oracle (C restatement of the reference's Dart algorithm) on a bounded sample of the
    same stream, members spread over host threads.  Reported, never part of the product path."""
    import concurrent.futures as cf

    from oracle import pyoracle
    lib = pyoracle.lib()
    ncores = os.cpu_count() or 1
    # member boundaries of the sample via the BC subfield / a quick serial probe of the oracle itself
    buf = comp
    base = buf.ctypes.data
    # calibrate: one member on one thread
    probe_members = min(args.members, 16)
    offs = member_offsets(buf, probe_members + 1)
    out = ctypes.create_string_buffer(args.member_bytes + 64)
    olen = ctypes.c_size_t()
    t0 = time.perf_counter()
    for i in range(probe_members):
        lib.orc_gzip_decode(base + offs[i], offs[i + 1] - offs[i], 0, 0, out, len(out), ctypes.byref(olen))
    per_member = (time.perf_counter() - t0) / probe_members
    n = int(min(args.members, max(ncores, args.cpu_seconds * ncores / max(per_member, 1e-9))))
    offs = member_offsets(buf, n + 1)
    chunks = [list(range(k, n, ncores)) for k in range(ncores)]

    def work(idx):
        o = ctypes.create_string_buffer(args.member_bytes + 64)
        ol = ctypes.c_size_t()
        tot = 0
        for i in idx:
            st = lib.orc_gzip_decode(base + offs[i], offs[i + 1] - offs[i], 0, 0, o, len(o), ctypes.byref(ol))
            assert st == 0
            tot += ol.value
        return tot
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(ncores) as ex:
        total = sum(ex.map(work, chunks))
    dt = time.perf_counter() - t0
    return {"value": round(total / dt / 1e9, 4), "unit": "GB/s", "cores": ncores, "kind": "port",
            "sample": "%d of %d members (%.0f MiB out), oracle/inflate_oracle.c, %d threads, %.1f s" % (
                n, args.members, total / 2**20, ncores, dt)}


def member_offsets(buf, count):
    """Start offsets of the first `count` members (walks BC subfields; falls back to zlib for no-BC)."""
    import zlib
    offs, p, n = [0], 0, len(buf)
    while len(offs) < count and p < n:
        if buf[p + 3] & 4 and buf[p + 12] == 66 and buf[p + 13] == 67:
            p += (int(buf[p + 16]) | (int(buf[p + 17]) << 8)) + 1
        else:
            d = zlib.decompressobj(31)
            d.decompress(bytes(buf[p:p + 4 * 65536 + 4096]))
            p += min(n - p, 4 * 65536 + 4096) - len(d.unused_data)
        offs.append(p)
    while len(offs) < count:
        offs.append(n)
    return offs


if __name__ == "__main__":
    main()
