#!/usr/bin/env python3
"""ONE 256 MiB gzip member (bench.py's config 2a stream) decoded by 1 / 2 / 4 / 8 ranks -- all of them handles in THIS
process, one after another on the one GPU there is, so a rank's phases are timed as they would run on a GPU of its own;
the gathers are numpy concatenations here (a real launch: two all-gathers of a few KB and one of 64 KiB a rank).

    python tools/split_bench.py [MiB [ranks,ranks,...]] > profiles/r06_stream_split.md

Per world size: for every phase the slowest rank's time (a step of a real job is the sum of those, plus the three
exchanges), next to the single-device decode of the same stream (ahip_gzip_decode_device)."""
import ctypes
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from archive_amd import _native as N
    from archive_amd.sharding import StreamSplit
    from tools import corpus
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    L = N.lib()
    assert L.ahip_init(0) == 0
    data = bytes(corpus.text(corpus.WIKI, 8, 0, mib << 20))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    gz = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255]) + co.compress(data) + co.flush() + zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")
    d_in = torch.frombuffer(bytearray(gz), dtype=torch.uint8).cuda()
    d_all = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
    olen = ctypes.c_size_t()

    def single():
        rc = L.ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), d_all.data_ptr(), d_all.numel(), ctypes.byref(olen), None)
        assert rc == 0 and olen.value == len(data)
    one = float("nan")
    if not os.environ.get("SPLIT_BENCH_NO_SINGLE"):  # (a kernel trace of the split alone: no single-device decodes in it)
        single(); single()
        t = time.perf_counter()
        for _ in range(5):
            single()
        one = (time.perf_counter() - t) / 5
    want_crc = zlib.crc32(data)

    print("# Round 6 -- one %d MiB gzip member (wiki text, level 6; %.1f MiB compressed) decoded by several ranks\n" % (mib, len(gz) / 2 ** 20))
    print("    python tools/split_bench.py %d\n" % mib)
    print("All ranks are handles in one process, run one after another on one MI355X: a rank's phases as they would run on a GPU of")
    print("its own.  Per phase the SLOWEST rank (ms); `step` = their sum = what a real job spends between its three exchanges")
    print("(two all-gathers of 8 / 32 B per block start, one of 64 KiB per rank).  Single-device decode of the same stream")
    print("(`ahip_gzip_decode_device`, the chunked path of section 10): **%.2f ms = %.1f GB/s**.\n" % (one * 1e3, len(data) / one / 1e9))
    print("| ranks | find | size | chain (host) | resolve + maps | finish (link, windows, translate) | step | GB/s out | x single device | slices (MiB, min .. max) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for world in ([int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
        best = None
        for rep in range(3):
            sps = [StreamSplit(d_in, 10, r, world) for r in range(world)]
            ph = np.zeros((world, 5))

            def timed(r, k, f):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                v = f()
                torch.cuda.synchronize()
                ph[r, k] += time.perf_counter() - t0
                return v
            cands = [timed(r, 0, sp.candidates) for r, sp in enumerate(sps)]
            all_cand = np.concatenate(cands)
            sized = [timed(r, 1, lambda sp=sp: sp.size(all_cand)) for r, sp in enumerate(sps)]
            assert all(h for h, _ in sized)
            all_res = np.concatenate([r for _, r in sized])
            chains = [timed(r, 2, lambda sp=sp: sp.chain(all_res)) for r, sp in enumerate(sps)]
            assert all(c[0] for c in chains)
            maps = torch.cat([timed(r, 3, sp.resolve) for r, sp in enumerate(sps)])
            outs = [torch.empty(c[2] + 64, dtype=torch.uint8, device="cuda") for c in chains]
            for r, sp in enumerate(sps):
                handled, n = timed(r, 4, lambda sp=sp, r=r: sp.finish(maps, outs[r]))
                assert handled and n == chains[r][2]
            crc = 0
            for r, c in enumerate(chains):  # the slices side by side are the member: CRC-32 of the trailer
                got = ctypes.c_uint32()
                L.ahip_crc32_device(outs[r].data_ptr(), c[2], crc, ctypes.byref(got), None)
                crc = got.value
            assert crc == want_crc and chains[0][3] == len(data) and chains[0][4] == len(gz) - 8
            for sp in sps:
                sp.close()
            step = ph.max(axis=0).sum()
            if best is None or step < best[0]:
                best = (step, ph.max(axis=0), [c[2] for c in chains], len(all_cand))
        step, mx, sizes, ncand = best
        print("| %d | %.2f | %.2f | %.2f | %.2f | %.2f | **%.2f** | %.1f | %.2f | %.1f .. %.1f (%d block starts) |" % (
            world, mx[0] * 1e3, mx[1] * 1e3, mx[2] * 1e3, mx[3] * 1e3, mx[4] * 1e3, step * 1e3, len(data) / step / 1e9, one / step,
            min(sizes) / 2 ** 20, max(sizes) / 2 ** 20, ncand))


if __name__ == "__main__":
    main()
