#!/usr/bin/env python3
"""bench.py -- headline benchmark: Inflate GB/s (uncompressed out) on a multi-member gzip stream.

One "step" = one complete decode of the rank's device-resident stream: member index build
(candidate scan, header parse, chain), the inflate kernels, verification, and -- for N > 1 -- the
one collective the path has: an all-gather of per-rank output sizes (RCCL) whose exclusive scan
is each shard's offset in the logical concatenated output.  Inputs and outputs stay in HBM.

Workload (config.workload): BASELINE.json configs[3] -- ONE stream of 65 536 gzip members x 64 KiB of
synthetic log text (4 GiB out, ~1.7 GiB in).  N = 1: that stream on one GPU.  N > 1: the SAME stream, cut into N
contiguous member ranges balanced on compressed bytes (archive_amd.sharding.partition_members); every rank indexes and
decodes only its slice -- `scaling: "strong"`, which is what BASELINE.json's "4 GiB ... 1/2/4/8 GPU" metric asks for.
--members shrinks it for quick runs.

The JSON line also carries
  check     CRC-32 of the decoded bytes taken on the device vs the CRC-32 the generator's gzip trailers imply
            (per-member CRCs combined with the GF(2) shift x^(8 len)) -- the number carries its own proof; for N > 1
            the shard CRCs are combined the same way and must equal the whole stream's;
  weak      (N > 1, unless --no-extras) every rank decoding its OWN 65 536 members (rank r holds members
            [r*M, (r+1)*M) of a stream of N*M members): per-GPU work fixed, reported next to the headline;
  one_member (N > 1, unless --no-extras) ONE gzip member of --one-member-mib MiB decoded by all ranks together
            (ahip_stream_split_*: a rank's range of the stream's blocks, three all-gathers a step), with the same member's
            single-device time beside it;
  extras    (N = 1) the other BASELINE configs, device-resident unless said otherwise: 2a one 256 MiB member (and one of 1 GiB), 2b
            4 096 members of wiki-like text, 3 Deflate level 6 on 1 GiB, 4 without the BGZF BC subfield, 5 bzip2,
            and the host-pointer entry point end to end (PCIe included).
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


# ---- CRC-32 of a concatenation from the CRCs of its pieces (zlib's crc32_combine, vectorised over many pairs) ----
def _gf2_times(mat, vec):
    """mat: (32,) uint32 columns; vec: (k,) uint32 -> (k,) uint32"""
    import numpy as np
    out = np.zeros_like(vec)
    for b in range(32):
        out ^= np.where((vec >> np.uint32(b)) & np.uint32(1), mat[b], np.uint32(0)).astype(np.uint32)
    return out


def _gf2_square(mat):
    return _gf2_times(mat, mat)


def _shift_matrix(nbytes):
    """operator that appends `nbytes` zero bytes to a CRC-32 register"""
    import numpy as np
    odd = np.zeros(32, dtype=np.uint32)
    odd[0] = 0xEDB88320
    for n in range(1, 32):
        odd[n] = np.uint32(1) << np.uint32(n - 1)
    even = _gf2_square(odd)  # two zero bits
    odd = _gf2_square(even)  # four
    ident = np.array([np.uint32(1) << np.uint32(b) for b in range(32)], dtype=np.uint32)
    result, n, cur = ident, int(nbytes), None
    # apply len2 zero BYTES: first squaring of `odd` is 8 bits = one byte
    cur = _gf2_square(odd)
    while n:
        if n & 1:
            result = _gf2_times(cur, result)
        n >>= 1
        if n:
            cur = _gf2_square(cur)
    return result


def crc32_of_concat(crcs, lens):
    """CRC-32 of piece_0 + piece_1 + ... from (crc_i, len_i); equal lengths are folded as a tree."""
    import numpy as np
    crcs = np.asarray(crcs, dtype=np.uint32)
    lens = [int(v) for v in lens]
    if len(set(lens)) == 1 and len(lens) & (len(lens) - 1) == 0 and len(lens) > 1:
        step = lens[0]
        while len(crcs) > 1:
            m = _shift_matrix(step)
            crcs = _gf2_times(m, crcs[0::2]) ^ crcs[1::2]
            step *= 2
        return int(crcs[0])
    total = np.uint32(crcs[0])
    for c, n in zip(crcs[1:], lens[1:]):
        total = _gf2_times(_shift_matrix(n), np.array([total], dtype=np.uint32))[0] ^ np.uint32(c)
    return int(total)


def member_table(buf, count):
    """(start offsets[count+1], crc32[count], isize[count]) of the first `count` members (BC subfields, else zlib)."""
    import zlib
    offs, crcs, sizes, p, n = [0], [], [], 0, len(buf)
    while len(crcs) < count and p < n:
        if buf[p + 3] & 4 and buf[p + 12] == 66 and buf[p + 13] == 67:
            q = p + (int(buf[p + 16]) | (int(buf[p + 17]) << 8)) + 1
        else:
            d = zlib.decompressobj(31)
            d.decompress(bytes(buf[p:p + 4 * 65536 + 4096]))
            q = p + min(n - p, 4 * 65536 + 4096) - len(d.unused_data)
        crcs.append(int.from_bytes(bytes(buf[q - 8:q - 4]), "little"))
        sizes.append(int.from_bytes(bytes(buf[q - 4:q]), "little"))
        offs.append(q)
        p = q
    return offs, crcs, sizes


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
    ap.add_argument("--check", action="store_true", help="also compare the decoded bytes with the generator's plain text")
    ap.add_argument("--one-member-mib", type=int, default=512, help="N > 1: size of the single member the ranks decode together (the `one_member` leg)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configs (N = 1) / the strong-scaling leg (N > 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collectives of the N > 1 run: nccl (= RCCL over xGMI, GPU tensors) or gloo (CPU tensors; for boxes with fewer GPUs than ranks)")
    ap.add_argument("--one-device", action="store_true",
                    help="every rank decodes on cuda:0 (with --backend gloo: runs the whole N > 1 path on a one-GPU box; a functional run, not a scaling figure)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # the bare command `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, exactly the way the
        # driver does it (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...)
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    dev_index = 0 if args.one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the tensors of the collectives live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from archive_amd import _native as N
    from tools import corpus
    L = N.lib()
    if L.ahip_init(dev_index) != 0:
        raise SystemExit("ahip_init failed: " + N.last_error())

    # ---- synthetic workload, generated on the host and made resident in HBM before timing ----
    kind = corpus.LOG if args.kind == "log" else corpus.WIKI
    seed = 1234 if kind == corpus.LOG else 8
    threads = max(1, (os.cpu_count() or 1) // max(1, local_world))
    stream = torch.cuda.current_stream()
    sh = ctypes.c_void_p(stream.cuda_stream)

    def decode_loop(d_src, d_dst, expect_bytes, steps, warmup, exchange, with_index=False):
        """times `steps` whole decodes of d_src; returns (elapsed seconds, mean ms of the inflate stage, bytes out).
        The HIP events bracket ahip_gzip_plan_run -- the inflate kernels -- or, with_index, plan_create + plan_run:
        members without the BC subfield are sized by a run of the tokenizer INSIDE plan_create, which the stage time
        must then include."""
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]

        def step(i=None):
            plan = ctypes.c_void_p()
            if i is not None and with_index:
                ev0[i].record(stream)
            rc = L.ahip_gzip_plan_create(d_src.data_ptr(), d_src.numel(), sh, ctypes.byref(plan))
            if rc != 0:
                raise SystemExit("plan_create: %d %s" % (rc, N.last_error()))
            ex = None
            if exchange and world > 1:
                # the path's one exchange: output-size all-gather -> shard offsets (RCCL).  The index knows the shard's size (the
                # ISIZE trailers, verified by the decode below), so the all-gather is started here and runs UNDER the inflate kernels
                from archive_amd.sharding import exchange_output_offsets_begin
                ob = ctypes.c_uint64()
                L.ahip_gzip_plan_info(plan, None, ctypes.byref(ob), None)
                ex = exchange_output_offsets_begin(ob.value, device=cdev)
            if i is not None and not with_index:
                ev0[i].record(stream)
            rc = L.ahip_gzip_plan_run(plan, d_dst.data_ptr(), d_dst.numel(), sh)
            if i is not None:
                ev1[i].record(stream)
            if rc != 0:
                raise SystemExit("plan_run: %d %s" % (rc, N.last_error()))
            olen = ctypes.c_size_t()
            rc = L.ahip_gzip_plan_status(plan, ctypes.byref(olen))
            L.ahip_gzip_plan_destroy(plan)
            if rc != 0 or (expect_bytes is not None and olen.value != expect_bytes):
                raise SystemExit("decode verdict %d, %d bytes (expected %s): %s" % (rc, olen.value, expect_bytes, N.last_error()))
            if ex is not None:
                _, _, sizes = ex.result()
                if sizes[rank] != olen.value:
                    raise SystemExit("rank %d: the index said %d bytes, the decode made %d" % (rank, sizes[rank], olen.value))
            return olen.value

        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            n_out = step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / steps
        return float(el.item()), kern_ms, n_out

    t0 = time.time()
    # N = 1: the stream.  N > 1: rank 0 makes THE stream (chunks 0 .. M-1); the others receive it below.
    comp, plain = (None, None)
    if world == 1 or rank == 0:
        comp, plain = corpus.make_gzip(kind=kind, seed=seed, n_members=args.members, member_bytes=args.member_bytes,
                                       level=6, bc=not args.no_bc, threads=threads if world == 1 else (os.cpu_count() or 1),
                                       first_chunk=0, want_plain=args.check and world == 1)
    gen_s = time.time() - t0
    out_bytes = args.members * args.member_bytes

    if world == 1:
        d_in = torch.from_numpy(comp).to(dev)
        d_out = torch.empty(out_bytes + 64, dtype=torch.uint8, device=dev)
        elapsed, kern_ms, _ = decode_loop(d_in, d_out, out_bytes, args.steps, args.warmup, True, with_index=args.no_bc)
        total_out = float(out_bytes)
        in_bytes_rank0 = int(d_in.numel())
        out_bytes_rank0 = out_bytes
        # ---- the proof: CRC-32 of what sits in d_out (device kernel) vs what the gzip trailers say it must be ----
        offs, crcs, sizes = member_table(comp, args.members)
        want_crc = crc32_of_concat(crcs, sizes)
        got = ctypes.c_uint32()
        if L.ahip_crc32_device(d_out.data_ptr(), out_bytes, 0, ctypes.byref(got), sh) != 0:
            raise SystemExit("crc32_device: " + N.last_error())
        crc_ok = got.value == want_crc and sum(sizes) == out_bytes
        if args.check:
            if not np.array_equal(d_out[:out_bytes].cpu().numpy(), plain):
                raise SystemExit("decoded bytes differ from the generator's plain text")
        if not crc_ok:
            raise SystemExit("device CRC-32 %08x != %08x expected from the member trailers" % (got.value, want_crc))
        check = {"crc32_device": "%08x" % got.value, "crc32_expected": "%08x" % want_crc, "ok": True,
                 "what": "CRC-32 of the decoded bytes (ahip_crc32_device) vs the member trailers' CRCs combined over GF(2)"}
        sharding, members_per_rank = "single GPU", None
    else:
        sr = strong_scaling(args, L, N, corpus, dist, torch, np, dev, cdev, sh, rank, world, comp, decode_loop)
        elapsed, kern_ms, total_out = sr["elapsed"], sr["kern_ms"], float(sr["total"])
        in_bytes_rank0, out_bytes_rank0 = sr["slice_in"], sr["slice_out"]
        check = sr["check"]
        members_per_rank = sr["members_per_rank"]
        sharding = ("ONE stream cut into %d contiguous member ranges balanced on compressed bytes, one process per GPU" % world
                    if not args.one_device else "ONE stream cut into %d member ranges, %d processes on ONE GPU (functional run)" % (world, world))

    weak = one = None
    if world > 1 and not args.no_extras:
        weak = weak_scaling(args, L, N, corpus, dist, torch, np, dev, cdev, sh, rank, world, kind, seed, threads, decode_loop)
        try:  # (rides along: whatever goes wrong in it must not cost the headline its line)
            one = one_member_split(args, L, N, corpus, dist, torch, np, dev, dev_index, cdev, sh, rank, world)
        except Exception as e:  # noqa: BLE001
            one = {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None

    if rank == 0:
        # C + U of what rank 0's kernels handled: compressed read once + output written once (N = 1: the whole stream)
        algo_bytes = float(in_bytes_rank0 + out_bytes_rank0)
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        step_ms = elapsed / args.steps * 1e3
        value = total_out * args.steps / elapsed / 1e9
        stage = ("ahip_gzip_plan_create + ahip_gzip_plan_run (the sizing run of the tokenizer sits inside plan_create)" if args.no_bc
                 else "ahip_gzip_plan_run")
        line = {
            "metric": "Inflate GB/s (uncompressed out) on multi-member gzip",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "ONE stream of %d gzip members x %d B %s text (%s), zlib level 6%s" % (
                args.members, args.member_bytes, args.kind, "BGZF BC subfield" if not args.no_bc else "no BC",
                "" if world == 1 else ", decoded once by %d ranks" % world),
                "members": args.members, "member_bytes": args.member_bytes,
                "compressed_bytes": int(len(comp)), "ratio": round(out_bytes / len(comp), 4),
                "sharding": sharding, "members_per_rank": members_per_rank,
                "collectives": ("RCCL (nccl backend), GPU tensors: one all-gather of 8 bytes per rank and step" if args.backend == "nccl" else "gloo, CPU tensors") if world > 1 else None,
                "gen_seconds": round(gen_s, 1)},
            "check": check,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "frac_step": round(algo_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": None,
                         "kernel": "inflate_tokenize_kernel + inflate_resolve_kernel (one launch each per decode; "
                                   "HIP events on the launch stream around %s)%s" % (stage, "" if world == 1 else "; rank 0's shard"),
                         "what": "frac = algorithmic bytes / kernel_ms (the inflate stage alone); frac_step = the same bytes / ms_per_step "
                                 "(index build, verification and verdict read-back included): the figure of record is the smaller one",
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
            if world == 1 and (w["members"], w["member_bytes"], w["kind"], w["bc"]) == (args.members, args.member_bytes, args.kind, not args.no_bc):
                line["roofline"]["traffic"] = round(pmc["traffic_bytes_per_launch"] / 1e9, 2)
                line["roofline"]["traffic_unit"] = "GB per decode"
                if "per_kernel" in pmc:
                    line["roofline"]["traffic_per_kernel"] = pmc["per_kernel"]
                line["roofline"]["traffic_source"] = ("NOT measured in this run: read from the committed profile profiles/%s (rocprofv3 --pmc passes over "
                                                      "this same command, L2 request counters by request size; tools/run_prof.sh)" % os.path.basename(latest))
        except Exception:
            pass
        if weak is not None:
            line["weak"] = weak
        if one is not None:
            line["one_member"] = one
        if world == 1 and not args.no_extras:
            del d_out
            line["extras"] = extras(args, L, N, corpus, torch, np, dev, sh, comp, d_in, out_bytes, decode_loop)
        if args.cpu_seconds > 0 and world == 1:  # the CPU baseline is reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(comp, args, out_bytes)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n):
    """Re-run this command line under torch.distributed.run with n local ranks (rendezvous on 127.0.0.1, a free port);
    rank 0's JSON line goes to our stdout, the exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def weak_scaling(args, L, N, corpus, dist, torch, np, dev, cdev, sh, rank, world, kind, seed, threads, decode_loop):
    """Per-GPU work fixed: rank r decodes its own members [r*M, (r+1)*M) of a stream of N*M members, with the same
    per-step size exchange.  Reported next to the strong-scaled headline, never as `value`."""
    comp, _ = corpus.make_gzip(kind=kind, seed=seed, n_members=args.members, member_bytes=args.member_bytes,
                               level=6, bc=not args.no_bc, threads=threads, first_chunk=rank * args.members)
    out_bytes = args.members * args.member_bytes
    d_in = torch.from_numpy(comp).to(dev)
    d_out = torch.empty(out_bytes + 64, dtype=torch.uint8, device=dev)
    steps = max(3, args.steps // 2)
    elapsed, kern_ms, _ = decode_loop(d_in, d_out, out_bytes, steps, 1, True, with_index=args.no_bc)
    offs, crcs, sizes = member_table(comp, args.members)
    want_crc = crc32_of_concat(crcs, sizes)
    got = ctypes.c_uint32()
    if L.ahip_crc32_device(d_out.data_ptr(), out_bytes, 0, ctypes.byref(got), sh) != 0:
        raise SystemExit("crc32_device: " + N.last_error())
    ok_all = torch.tensor([1.0 if (got.value == want_crc and sum(sizes) == out_bytes) else 0.0], dtype=torch.float64, device=cdev)
    dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
    if float(ok_all.item()) != 1.0:
        raise SystemExit("rank %d (weak leg): device CRC-32 %08x != %08x expected from the member trailers" % (rank, got.value, want_crc))
    if rank != 0:
        return None
    return {"scaling": "weak", "value": round(out_bytes * world * steps / elapsed / 1e9, 3), "unit": "GB/s", "steps": steps,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "n_gpus": world, "kernel_ms": round(kern_ms, 4),
            "workload": "%d members x %d B PER GPU (rank r: members r*M .. (r+1)*M of one stream of N*M)" % (args.members, args.member_bytes),
            "check": {"ok": True, "what": "per-rank device CRC-32 vs the member trailers, every rank"}}


def one_member_split(args, L, N, corpus, dist, torch, np, dev, dev_index, cdev, sh, rank, world):
    """ONE gzip member (BASELINE config 2a's kind of stream, --one-member-mib of wiki text, compressed pigz-style on rank 0
    and broadcast untimed) decoded by ALL ranks: archive_amd.sharding.ShardedStreamDecoder -- every rank finds, sizes and
    resolves its range of the stream's blocks and writes its slice; three all-gathers per step.  Rides along like `weak`."""
    import zlib
    from archive_amd.sharding import ShardedStreamDecoder
    nbytes = args.one_member_mib << 20
    if rank == 0:
        gz, want_crc = corpus.make_one_member(kind=corpus.WIKI, seed=8, nbytes=nbytes, level=6, threads=os.cpu_count() or 1)
    n_in = torch.tensor([len(gz) if rank == 0 else 0, want_crc if rank == 0 else 0], dtype=torch.int64, device=cdev)
    dist.broadcast(n_in, 0)
    whole = torch.from_numpy(gz.copy()).to(cdev) if rank == 0 else torch.empty(int(n_in[0].item()), dtype=torch.uint8, device=cdev)
    dist.broadcast(whole, 0)
    d_in = whole.to(dev)
    want_crc = int(n_in[1].item())
    dec = ShardedStreamDecoder(device_index=dev_index, collective_device=None if cdev.type == "cuda" else "cpu")
    steps = max(3, args.steps // 2)

    def step():
        return dec.decode(d_in, 10)
    step()
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        d_out, n, offset, total, end_pos = step()
    torch.cuda.synchronize(); dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    got = ctypes.c_uint32()
    if L.ahip_crc32_device(d_out.data_ptr(), n, 0, ctypes.byref(got), sh) != 0:
        raise SystemExit("crc32_device: " + N.last_error())
    mine = torch.tensor([got.value, n, offset, 1 if dec.last_handled else 0], dtype=torch.int64, device=cdev)
    allv = torch.zeros(4 * world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(allv, mine)
    single_ms = None
    if rank == 0:  # the same member on one device, for the ratio
        o1 = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
        olen = ctypes.c_size_t()
        for i in range(4):
            if i == 1:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            if L.ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), o1.data_ptr(), o1.numel(), ctypes.byref(olen), sh) != 0:
                raise SystemExit("gzip_decode_device: " + N.last_error())
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / 3 * 1e3
        del o1
    dist.barrier()
    if rank != 0:
        return None
    rows = [[int(v) for v in allv[4 * r:4 * r + 4].tolist()] for r in range(world)]
    combined = crc32_of_concat([r[0] for r in rows], [r[1] for r in rows])
    ok = combined == want_crc and sum(r[1] for r in rows) == nbytes == total and end_pos == d_in.numel() - 8
    if not ok:
        return {"error": "one member on %d ranks: slices combine to %08x (%d bytes), expected %08x (%d)" % (world, combined, sum(r[1] for r in rows), want_crc, nbytes)}
    sec = float(el.item()) / steps
    return {"value": round(nbytes / sec / 1e9, 3), "unit": "GB/s", "ms_per_step": round(sec * 1e3, 4), "steps": steps, "n_gpus": world,
            "single_device_ms": round(single_ms, 4), "speedup_vs_single_device": round(single_ms / (sec * 1e3), 3),
            "split_path_on_every_rank": all(r[3] == 1 for r in rows), "slice_MiB": [round(r[1] / 2 ** 20, 1) for r in rows],
            "workload": "ONE gzip member of %d MiB wiki text (pigz-style level 6, %.1f MiB compressed) decoded by %d ranks: ahip_stream_split_*, three all-gathers per step" % (
                args.one_member_mib, d_in.numel() / 2 ** 20, world),
            "check": {"ok": True, "what": "per-slice device CRC-32 combined over GF(2) in rank order vs the member's trailer; slices back to back; end position in front of the trailer"}}


def strong_scaling(args, L, N, corpus, dist, torch, np, dev, cdev, sh, rank, world, comp0, decode_loop):
    """The headline of N > 1 -- BASELINE config 4 as written: ONE stream (rank 0's, broadcast untimed), partitioned over
    the ranks on compressed bytes (archive_amd.sharding.partition_members); a rank indexes and decodes only its slice;
    the exchange is the size all-gather of every step; the shard CRCs (device kernel) must combine to the whole
    stream's.  W warm-up and exactly K timed steps, like the N = 1 run."""
    from archive_amd.sharding import exchange_output_offsets, partition_members
    n_in = torch.tensor([len(comp0) if rank == 0 else 0], dtype=torch.int64, device=cdev)
    dist.broadcast(n_in, 0)
    whole = torch.from_numpy(comp0).to(cdev) if rank == 0 else torch.empty(int(n_in.item()), dtype=torch.uint8, device=cdev)
    dist.broadcast(whole, 0)  # setup, untimed: afterwards every rank only touches its own slice
    whole = whole.to(dev)
    if rank == 0:
        offs0, crcs0, sizes0 = member_table(comp0, args.members)
    meta = torch.tensor(offs0, dtype=torch.int64, device=cdev) if rank == 0 else torch.empty(args.members + 1, dtype=torch.int64, device=cdev)
    dist.broadcast(meta, 0)
    offs = [int(v) for v in meta.tolist()]
    csize = [offs[i + 1] - offs[i] for i in range(args.members)]
    parts = partition_members(csize, world)
    lo, hi = parts[rank]
    d_slice = whole[offs[lo]:offs[hi]]
    expect = (hi - lo) * args.member_bytes
    d_dst = torch.empty(expect + 64, dtype=torch.uint8, device=dev)
    elapsed, kern_ms, n_out = decode_loop(d_slice, d_dst, expect, args.steps, args.warmup, True, with_index=args.no_bc)
    offset, total, all_sizes = exchange_output_offsets(n_out, device=cdev)
    got = ctypes.c_uint32()
    if L.ahip_crc32_device(d_dst.data_ptr(), n_out, 0, ctypes.byref(got), sh) != 0:
        raise SystemExit("crc32_device: " + N.last_error())
    shard = torch.tensor([got.value], dtype=torch.int64, device=cdev)
    allc = torch.zeros(world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(allc, shard)
    res = {"elapsed": elapsed, "kern_ms": kern_ms, "total": total, "slice_in": offs[hi] - offs[lo], "slice_out": expect,
           "members_per_rank": [b - a for a, b in parts], "check": None}
    if rank != 0:
        return res
    combined = crc32_of_concat([int(v) for v in allc.tolist()], all_sizes)
    want = crc32_of_concat(crcs0, sizes0)
    if combined != want or total != args.members * args.member_bytes:
        raise SystemExit("strong scaling: shard CRCs combine to %08x, expected %08x (total %d bytes)" % (combined, want, total))
    res["check"] = {"crc32_combined": "%08x" % combined, "crc32_expected": "%08x" % want, "ok": True,
                    "what": "per-shard CRC-32 of the decoded bytes (ahip_crc32_device), combined over GF(2) in rank order, vs the "
                            "member trailers' CRCs of the whole stream"}
    return res



def bz2_repeat(cz, times):
    """One bzip2 stream whose blocks are the blocks of the stream `cz`, `times` times over -- a long stream for the price of
    compressing a short one on the host.  Blocks are not byte aligned: the block section is spliced bit by bit; the stream's
    CRC is the rotate-and-xor fold of the block CRCs (each sits behind its block's 48-bit magic).  Returns (stream, blocks)."""
    import numpy as np
    total = len(cz) * 8
    big = int.from_bytes(cz, "big")
    eos = None
    for pad in range(8):  # end-of-stream magic, 32 bits of CRC, 0..7 bits of padding
        p = total - pad - 32 - 48
        if (big >> (total - p - 48)) & ((1 << 48) - 1) == 0x177245385090:
            eos = p
            break
    assert eos is not None, "not a bzip2 stream"
    n_b = eos - 32
    blocks = (big >> (total - eos)) & ((1 << n_b) - 1)
    a = np.frombuffer(cz, dtype=np.uint8)
    w = np.zeros(len(a) - 7, dtype=np.uint64)
    for k in range(7):
        w = (w << np.uint64(8)) | a[k:len(a) - 7 + k].astype(np.uint64)
    pos = []
    for sft in range(8):  # the block magic at every bit offset
        win = (w >> np.uint64(8 - sft)) & np.uint64((1 << 48) - 1)
        pos += [int(h) * 8 + sft for h in np.nonzero(win == np.uint64(0x314159265359))[0]]
    pos.sort()
    crcs = [(big >> (total - q - 48 - 32)) & 0xffffffff for q in pos]
    comb = 0
    for _ in range(times):
        for c in crcs:
            comb = (((comb << 1) | (comb >> 31)) & 0xffffffff) ^ c
    acc, nbits = int.from_bytes(cz[:4], "big"), 32
    for _ in range(times):
        acc = (acc << n_b) | blocks
        nbits += n_b
    acc = (((acc << 48) | 0x177245385090) << 32) | comb
    nbits += 80
    acc <<= (-nbits) % 8
    nbits += (-nbits) % 8
    return acc.to_bytes(nbits // 8, "big"), len(crcs) * times


def extras(args, L, N, corpus, torch, np, dev, sh, comp, d_in, out_bytes, decode_loop):
    """The other BASELINE configs on this GPU (short runs; every figure with its own (C+U)/t fraction of HBM peak)."""
    import bz2
    import zlib
    res = {}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    def frac(c_bytes, u_bytes, sec):
        return round((c_bytes + u_bytes) / sec / 1e9 / HBM_PEAK_GBPS, 5)

    try:  # config 4 without the BC subfield: the index has to measure every member first
        c2, _ = corpus.make_gzip(kind=corpus.LOG, seed=1234, n_members=args.members, member_bytes=args.member_bytes, level=6, bc=False,
                                 threads=os.cpu_count() or 1)
        d2 = torch.from_numpy(c2).to(dev)
        o2 = torch.empty(out_bytes + 64, dtype=torch.uint8, device=dev)
        el, _, _ = decode_loop(d2, o2, out_bytes, 3, 1, False)
        res["config4_no_bc"] = {"value": round(out_bytes * 3 / el / 1e9, 2), "unit": "GB/s out", "ms": round(el / 3 * 1e3, 2),
                                "hbm_frac": frac(len(c2), out_bytes, el / 3), "what": "same members without BGZF BC: sizing run + decode"}
        del d2, o2
    except SystemExit as e:
        res["config4_no_bc"] = {"error": str(e)}

    try:  # host-pointer entry point, end to end (what a dart:ffi caller gets): H2D + decode + D2H
        host_out = np.empty(out_bytes + 64, dtype=np.uint8)
        olen = ctypes.c_size_t()

        def call():
            rc = L.ahip_gzip_decode(comp.ctypes.data, len(comp), 0, 0, host_out.ctypes.data, len(host_out), ctypes.byref(olen))
            assert rc == 0 and olen.value == out_bytes, (rc, olen.value, N.last_error())
        sec = timed(call, reps=2)
        res["config4_host_pointers"] = {"value": round(out_bytes / sec / 1e9, 2), "unit": "GB/s out", "ms": round(sec * 1e3, 1),
                                        "what": "ahip_gzip_decode(host in, host out): PCIe both ways included; bound ~ 63 GB/s x (C+U)/U",
                                        "crc_ok": bool(zlib.crc32(host_out[:1 << 24].tobytes()) == zlib.crc32(corpus_plain_head(corpus, args, 1 << 24)))}
        del host_out
    except AssertionError as e:
        res["config4_host_pointers"] = {"error": str(e)}

    try:  # config 4's kind of member compressed by the REFERENCE's own Deflate (the oracle's restatement, level 6, with its
        # 8 192-symbol block truncation, deflate.dart:549-562: more and shorter blocks per member than zlib's) -- what a Dart user's
        # own GZipEncoder output looks like to this decoder.  4 096 members; the oracle only makes the input here.
        import struct
        from concurrent.futures import ThreadPoolExecutor
        from oracle import pyoracle
        n_or, mb = 4096, 65536
        plain = np.empty(n_or * mb, dtype=np.uint8)
        for c in range(n_or * mb >> 20):
            corpus.lib().corpus_log_text(1234, c * 16, plain[c << 20:].ctypes.data, 1 << 20)

        def member(i):
            d = plain[i * mb:(i + 1) * mb].tobytes()
            body, crc = pyoracle.deflate_raw(d, 6, True)
            total = 18 + len(body) + 8
            assert total <= 65536
            return bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", total - 1) + body + struct.pack("<II", crc, mb), body[0] & 1
        pyoracle.lib()
        with ThreadPoolExecutor(min(64, os.cpu_count() or 1)) as ex:
            made = list(ex.map(member, range(n_or)))
        c7 = np.frombuffer(b"".join(m for m, _ in made), dtype=np.uint8)
        d7 = torch.from_numpy(c7.copy()).to(dev)
        o7 = torch.empty(n_or * mb + 64, dtype=torch.uint8, device=dev)
        el, km, _ = decode_loop(d7, o7, n_or * mb, 5, 2, False)
        got = ctypes.c_uint32()
        L.ahip_crc32_device(o7.data_ptr(), n_or * mb, 0, ctypes.byref(got), None)
        res["config4_reference_deflate_4096_members"] = {
            "value": round(n_or * mb * 5 / el / 1e9, 2), "unit": "GB/s out", "ms": round(el / 5 * 1e3, 3), "kernel_ms": round(km, 3),
            "ratio": round(n_or * mb / len(c7), 4), "members_whose_first_block_is_not_their_last": int(sum(1 for _, f in made if not f)),
            "hbm_frac": frac(len(c7), n_or * mb, km * 1e-3), "crc_ok": bool(got.value == zlib.crc32(plain.tobytes())),
            "what": "4 096 x 64 KiB of log text compressed by the oracle's restatement of the reference's Deflate (level 6, block truncation on) instead of zlib"}
        del d7, o7, plain
    except (AssertionError, SystemExit) as e:
        res["config4_reference_deflate_4096_members"] = {"error": str(e)}

    try:  # config 2b: 4 096 members of wiki-like text
        c3, _ = corpus.make_gzip(kind=corpus.WIKI, seed=8, n_members=4096, member_bytes=65536, level=6, bc=True, threads=os.cpu_count() or 1)
        d3 = torch.from_numpy(c3).to(dev)
        o3 = torch.empty(4096 * 65536 + 64, dtype=torch.uint8, device=dev)
        el, km, _ = decode_loop(d3, o3, 4096 * 65536, 5, 2, False)
        res["config2b_wiki_4096_members"] = {"value": round(4096 * 65536 * 5 / el / 1e9, 2), "unit": "GB/s out", "ms": round(el / 5 * 1e3, 3),
                                             "kernel_ms": round(km, 3), "hbm_frac": frac(len(c3), 4096 * 65536, km * 1e-3)}
        del d3, o3
    except SystemExit as e:
        res["config2b_wiki_4096_members"] = {"error": str(e)}

    try:  # config 2a: ONE 256 MiB member (chunked single-stream path)
        data = bytes(corpus.text(corpus.WIKI, 8, 0, 256 << 20))
        gz = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255])
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        gz += co.compress(data) + co.flush() + zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")
        d4 = torch.frombuffer(bytearray(gz), dtype=torch.uint8).to(dev)
        o4 = torch.empty(len(data) + 64, dtype=torch.uint8, device=dev)
        olen = ctypes.c_size_t()

        def call():
            rc = L.ahip_gzip_decode_device(d4.data_ptr(), d4.numel(), o4.data_ptr(), o4.numel(), ctypes.byref(olen), None)
            assert rc == 0 and olen.value == len(data), (rc, olen.value, N.last_error())
        sec = timed(call)
        got = ctypes.c_uint32()
        L.ahip_crc32_device(o4.data_ptr(), len(data), 0, ctypes.byref(got), None)
        res["config2a_one_256MiB_member"] = {"value": round(len(data) / sec / 1e9, 2), "unit": "GB/s out", "ms": round(sec * 1e3, 2),
                                             "hbm_frac": frac(len(gz), len(data), sec), "crc_ok": bool(got.value == zlib.crc32(data))}
        del d4, o4, data, gz
    except AssertionError as e:
        res["config2a_one_256MiB_member"] = {"error": str(e)}

    try:  # config 2a at four times the size, compressed the way pigz does it (pieces primed with the 32 KiB in front, sync-flush markers between)
        gz1, crc1 = corpus.make_one_member(kind=corpus.WIKI, seed=8, nbytes=1 << 30, level=6, threads=os.cpu_count() or 1)
        d8 = torch.from_numpy(gz1.copy()).to(dev)
        o8 = torch.empty((1 << 30) + 64, dtype=torch.uint8, device=dev)
        olen = ctypes.c_size_t()

        def call8():
            rc = L.ahip_gzip_decode_device(d8.data_ptr(), d8.numel(), o8.data_ptr(), o8.numel(), ctypes.byref(olen), None)
            assert rc == 0 and olen.value == 1 << 30, (rc, olen.value, N.last_error())
        sec = timed(call8)
        got = ctypes.c_uint32()
        L.ahip_crc32_device(o8.data_ptr(), 1 << 30, 0, ctypes.byref(got), None)
        res["config2a_one_1GiB_member_pigz"] = {"value": round((1 << 30) / sec / 1e9, 2), "unit": "GB/s out", "ms": round(sec * 1e3, 2),
                                                "hbm_frac": frac(len(gz1), 1 << 30, sec), "crc_ok": bool(got.value == crc1),
                                                "chunks": int(L.ahip_debug_last_chunks()),
                                                "what": "ONE gzip member of 1 GiB wiki text (not a BASELINE config: the 256 MiB member above is): a member of this size fills the chip's waves several times over"}
        del d8, o8, gz1
    except AssertionError as e:
        res["config2a_one_1GiB_member_pigz"] = {"error": str(e)}

    try:  # config 3: Deflate level 6 on 1 GiB of log text
        n = 1 << 30
        buf = np.empty(n, dtype=np.uint8)
        for c in range(n >> 20):
            corpus.lib().corpus_log_text(1234, c * 16, buf[c << 20:].ctypes.data, 1 << 20)
        d5 = torch.from_numpy(buf).to(dev)
        o5 = torch.empty(L.ahip_deflate_bound(n), dtype=torch.uint8, device=dev)
        olen = ctypes.c_size_t()

        def call():
            rc = L.ahip_deflate_raw_device(d5.data_ptr(), n, 6, 15, o5.data_ptr(), o5.numel(), ctypes.byref(olen), None)
            assert rc == 0, (rc, N.last_error())
        sec = timed(call)
        from oracle import pyoracle  # the checker: reference size on a 4 MiB sample
        sample = buf[:4 << 20].tobytes()
        ref = len(pyoracle.deflate_raw(sample, 6)[0])
        so = ctypes.c_size_t()
        L.ahip_deflate_raw_device(d5.data_ptr(), len(sample), 6, 15, o5.data_ptr(), o5.numel(), ctypes.byref(so), None)
        res["config3_deflate_L6_1GiB"] = {"value": round(n / sec / 1e9, 2), "unit": "GB/s in", "ms": round(sec * 1e3, 2), "ratio": round(n / olen.value, 4),
                                          "size_vs_reference": round(so.value / ref, 4), "hbm_frac": frac(olen.value, n, sec),
                                          "what": "ahip_deflate_raw_device; size / oracle size on the first 4 MiB"}
        del d5, o5, buf
    except AssertionError as e:
        res["config3_deflate_L6_1GiB"] = {"error": str(e)}

    try:  # config 5: bzip2 -9, 64 blocks of 900 k
        data = bytes(corpus.text(corpus.WIKI, 8, 0, 64 * 900000))
        cz = bz2.compress(data, 9)
        d6 = torch.frombuffer(bytearray(cz), dtype=torch.uint8).to(dev)
        o6 = torch.empty(len(data) + 64, dtype=torch.uint8, device=dev)
        olen = ctypes.c_size_t()

        def call():
            rc = L.ahip_bzip2_decode_device(d6.data_ptr(), d6.numel(), 1, o6.data_ptr(), o6.numel(), ctypes.byref(olen), None)
            assert rc == 0 and olen.value == len(data), (rc, olen.value, N.last_error())
        sec = timed(call)
        got = ctypes.c_uint32()
        L.ahip_crc32_device(o6.data_ptr(), len(data), 0, ctypes.byref(got), None)
        res["config5_bzip2_64x900k"] = {"value": round(len(data) / sec / 1e9, 3), "unit": "GB/s out", "ms": round(sec * 1e3, 1),
                                        "hbm_frac": frac(len(cz), len(data), sec), "crc_ok": bool(got.value == zlib.crc32(data)),
                                        "what": "ahip_bzip2_decode_device on ONE stream of 64 blocks; a third of this time is steps that are serial "
                                                "per block, so longer streams run faster: the next entry"}
        # the same blocks seven times over in ONE stream: the steps that are serial per block are paid once per batch
        big, nblk = bz2_repeat(cz, 7)
        del d6, o6
        d7 = torch.frombuffer(bytearray(big), dtype=torch.uint8).to(dev)
        o7 = torch.empty(7 * len(data) + 64, dtype=torch.uint8, device=dev)

        def call7():
            rc = L.ahip_bzip2_decode_device(d7.data_ptr(), d7.numel(), 1, o7.data_ptr(), o7.numel(), ctypes.byref(olen), None)
            assert rc == 0 and olen.value == 7 * len(data), (rc, olen.value, N.last_error())
        sec = timed(call7)
        L.ahip_crc32_device(o7.data_ptr(), 7 * len(data), 0, ctypes.byref(got), None)
        want = 0
        for _ in range(7):
            want = zlib.crc32(data, want)
        res["config5_bzip2_%dx900k" % nblk] = {"value": round(7 * len(data) / sec / 1e9, 3), "unit": "GB/s out", "ms": round(sec * 1e3, 1),
                                                "hbm_frac": frac(len(big), 7 * len(data), sec), "crc_ok": bool(got.value == want),
                                                "what": "the 64 x 900k stream's blocks seven times over in one stream (bz2_repeat): %d blocks, one batch" % nblk}
        del d7, o7
    except AssertionError as e:
        res["config5_bzip2_64x900k"] = {"error": str(e)}
    return res


def corpus_plain_head(corpus, args, n):
    """first n bytes of the headline workload's plain text (member 0.. of the log corpus)"""
    import zlib
    comp, plain = corpus.make_gzip(kind=corpus.LOG if args.kind == "log" else corpus.WIKI, seed=1234 if args.kind == "log" else 8,
                                   n_members=(n + args.member_bytes - 1) // args.member_bytes, member_bytes=args.member_bytes, level=6, want_plain=True)
    return plain[:n].tobytes()


def cpu_baseline(comp, args, out_bytes):
    """The CPU oracle (C restatement of the reference's Dart algorithm) on a bounded sample of the
    same stream, members spread over host threads.  Reported, never part of the product path."""
    import concurrent.futures as cf

    from oracle import pyoracle
    lib = pyoracle.lib()
    ncores = os.cpu_count() or 1
    buf = comp
    base = buf.ctypes.data
    # calibrate: one member on one thread
    probe_members = min(args.members, 16)
    offs, _, _ = member_table(buf, probe_members)
    out = ctypes.create_string_buffer(args.member_bytes + 64)
    olen = ctypes.c_size_t()
    t0 = time.perf_counter()
    for i in range(probe_members):
        lib.orc_gzip_decode(base + offs[i], offs[i + 1] - offs[i], 0, 0, out, len(out), ctypes.byref(olen))
    per_member = (time.perf_counter() - t0) / probe_members

    def run(nthreads, seconds):
        n = int(min(args.members, max(nthreads, seconds * nthreads / max(per_member, 1e-9))))
        offs, _, _ = member_table(buf, n)
        chunks = [list(range(k, n, nthreads)) for k in range(nthreads)]

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
        with cf.ThreadPoolExecutor(nthreads) as ex:
            total = sum(ex.map(work, chunks))
        dt = time.perf_counter() - t0
        return total / dt / 1e9, n, total, dt

    v, n, total, dt = run(ncores, args.cpu_seconds)
    line = {"value": round(v, 4), "unit": "GB/s", "cores": ncores, "kind": "port",
            "sample": "%d of %d members (%.0f MiB out), oracle/inflate_oracle.c, %d threads, %.1f s" % (
                n, args.members, total / 2**20, ncores, dt)}
    # the same restatement on 1 and on 8 threads (the reference itself is single-threaded)
    v1, n1, _, d1 = run(1, min(3.0, args.cpu_seconds / 3))
    v8, n8, _, d8 = run(min(8, ncores), min(3.0, args.cpu_seconds / 3))
    line["one_thread"] = {"value": round(v1, 4), "unit": "GB/s", "cores": 1, "sample": "%d members, %.1f s" % (n1, d1)}
    line["eight_threads"] = {"value": round(v8, 4), "unit": "GB/s", "cores": min(8, ncores), "sample": "%d members, %.1f s" % (n8, d8)}
    return line


if __name__ == "__main__":
    main()
