#!/usr/bin/env python3
"""One-off sweep on the GPU: random long DEFLATE streams through the stream split (ahip_stream_split_*) with random numbers of
ranks, all ranks as handles of this process; the slices side by side must be the input (zlib made the stream), and a damaged
copy of every stream must either be refused at the same step by every rank or -- when the damage still decodes -- give what
the single-device path (archive_amd.Inflate: the reference's verdicts, tests/test_single_stream_gpu.py) gives.

    python tools/split_sweep.py [seconds] [seed]"""
import os
import random
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_data(rnd, corpus, n):
    """n bytes of a random mix: text, noise (stored blocks), zeros / short periods (long self-overlapping matches), copies from far
    back (matches across chunk and rank edges)."""
    out = bytearray()
    while len(out) < n:
        kind = rnd.randrange(7)
        ln = rnd.choice([20000, 150000, 600000, 2000000])
        if kind <= 1:
            out += bytes(corpus.text(corpus.LOG if kind == 0 else corpus.WIKI, rnd.randrange(1 << 16), rnd.randrange(64), ln))
        elif kind == 2:
            out += rnd.randbytes(ln // 2)
        elif kind == 3:
            out += bytes(ln)
        elif kind == 4:
            p = rnd.randbytes(rnd.choice([1, 2, 3, 7, 258, 1000]))
            out += (p * (ln // len(p) + 1))[:ln]
        elif kind == 5 and len(out) > 40000:
            d = rnd.randrange(1, 32768)
            for _ in range(ln // 4096):
                out += out[-d:-d + min(d, 4096)] if d > 4096 else out[-d:] * (4096 // d)
        else:
            out += bytes(rnd.choice(b"abcdefgh ") for _ in range(ln // 8))
    return bytes(out[:n])


def compress(rnd, data):
    level = rnd.choice([1, 1, 3, 6, 6, 9])
    mem = rnd.choice([1, 4, 8, 9])          # small memLevel: short blocks (many block starts)
    strat = rnd.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strat)
    flush = rnd.choice([0, 0, 0, 100000, 1000000])
    kindf = rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])
    if not flush:
        return co.compress(data) + co.flush(), (level, mem, strat, 0)
    out = b""
    for o in range(0, len(data), flush):
        out += co.compress(data[o:o + flush]) + co.flush(kindf)
    return out + co.flush(), (level, mem, strat, flush)


def split_decode(torch, StreamSplit, stream, data_off, world):
    d_in = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
    sps = [StreamSplit(d_in, data_off, r, world) for r in range(world)]
    try:
        all_cand = np.concatenate([sp.candidates() for sp in sps])
        sized = [sp.size(all_cand) for sp in sps]
        assert len({h for h, _ in sized}) == 1
        if not sized[0][0]:
            return None, None
        all_res = np.concatenate([r for _, r in sized])
        chains = [sp.chain(all_res) for sp in sps]
        assert len({c[0] for c in chains}) == 1
        if not chains[0][0]:
            return None, None
        total, end_pos = chains[0][3], chains[0][4]
        maps = torch.cat([sp.resolve() for sp in sps])
        out, at = bytearray(total), 0
        for sp, (_, off, n, _, _) in zip(sps, chains):
            assert off == at
            d_out = torch.full((n + 64,), 0x5A, dtype=torch.uint8, device="cuda")
            handled, got = sp.finish(maps, d_out)
            if not handled:
                return None, None
            assert got == n and bool((d_out[n:] == 0x5A).all())
            out[off:off + n] = bytes(d_out[:n].cpu().numpy())
            at += n
        assert at == total
        return bytes(out), end_pos
    finally:
        for sp in sps:
            sp.close()


def main():
    import torch
    import archive_amd
    from archive_amd import _native as N, errors
    from archive_amd.sharding import StreamSplit
    from tools import corpus
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    assert N.lib().ahip_init(0) == 0
    t0 = time.time()
    n_ok = n_unhandled = n_dam = n_dam_decoded = 0
    worlds = {}
    while time.time() - t0 < seconds:
        data = make_data(rnd, corpus, rnd.choice([3, 5, 8, 12]) << 20)
        raw, how = compress(rnd, data)
        off = rnd.choice([0, 0, 10, 3])
        stream = bytes(off) + raw + rnd.randbytes(rnd.choice([0, 8, 100]))
        world = rnd.choice([1, 2, 2, 3, 4, 5, 8])
        two_pass = rnd.random() < 0.25
        if two_pass:
            os.environ["AHIP_SM_TWO_PASS"] = "1"
        else:
            os.environ.pop("AHIP_SM_TWO_PASS", None)
        got, end_pos = split_decode(torch, StreamSplit, stream, off, world)
        if got is None:
            n_unhandled += 1  # (short compressed size, Z_HUFFMAN_ONLY / Z_RLE streams of few blocks, ...)
        else:
            if got != data or end_pos != off + len(raw):
                print("MISMATCH: seed %d, %d bytes, world %d, how %s, two_pass %s" % (seed, len(data), world, how, two_pass))
                sys.exit(1)
            n_ok += 1
            worlds[world] = worlds.get(world, 0) + 1
        # the same stream, damaged
        b = bytearray(stream)
        p = rnd.randrange(off + len(raw) // 8, off + len(raw))
        b[p] ^= 1 << rnd.randrange(8)
        got, _ = split_decode(torch, StreamSplit, bytes(b), off, world)
        n_dam += 1
        if got is not None:
            n_dam_decoded += 1
            try:
                z = archive_amd.Inflate(bytes(b[off:]))
                want = (z.status, z.get_bytes())
            except errors.ArchiveHipError:
                want = None
            if want is None or want[0] != 0 or want[1] != got:
                print("MISMATCH on a damaged stream: seed %d, world %d, how %s, byte %d" % (seed, world, how, p))
                sys.exit(1)
    print("split sweep, seed %d, %.0f s: %d streams decoded by the split and equal to the input (ranks: %s), %d not taken by the path; "
          "%d damaged copies: %d refused by every rank at the same step, %d decoded -- to what the single-device path makes of them" % (
              seed, time.time() - t0, n_ok, ", ".join("%d x %d" % (v, k) for k, v in sorted(worlds.items())), n_unhandled, n_dam, n_dam - n_dam_decoded, n_dam_decoded))


if __name__ == "__main__":
    main()
