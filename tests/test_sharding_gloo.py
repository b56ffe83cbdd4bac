"""CPU-only, world_size 2 over gloo: the N>1 path of the benchmark -- member partitioning and the
output-size all-gather/prefix-scan -- and that shards concatenated at the exchanged offsets equal
the whole-stream decode (checked with the CPU oracle; the GPU decode itself is covered by -m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import streams


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, members, ranges, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from archive_amd.sharding import exchange_output_offsets
        from oracle import pyoracle
        lo, hi = ranges[rank]
        shard = b"".join(members[lo:hi])
        st, out = pyoracle.gzip_decode(shard) if shard else (0, b"")
        assert st == 0
        off, total, sizes = exchange_output_offsets(len(out))
        q.put((rank, off, total, sizes, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_partition_is_contiguous_and_balanced():
    from archive_amd.sharding import partition_members
    sizes = [100] * 1000
    for w in (1, 2, 4, 8):
        r = partition_members(sizes, w)
        assert r[0][0] == 0 and r[-1][1] == 1000 and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1
    r = partition_members([10, 1000, 10, 10], 2)
    assert r[0][0] == 0 and r[-1][1] == 4 and r[0][1] == r[1][0]
    assert partition_members([], 2) == [(0, 0), (0, 0)]
    assert partition_members([5], 4)[-1][1] == 1


def test_two_rank_offsets_and_concatenation():
    from archive_amd.sharding import partition_members
    from oracle import pyoracle
    payloads = [streams.text(1000 + 997 * i, i) for i in range(11)] + [b""]
    members = [streams.bgzf_member(p) if i % 2 else streams.gz_member(p) for i, p in enumerate(payloads)]
    ranges = partition_members([len(m) for m in members], 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, members, ranges, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    st, whole = pyoracle.gzip_decode(b"".join(members))
    assert st == 0 and whole == b"".join(payloads)
    buf = bytearray(res[0][2])
    for rank, off, total, sizes, out in res:
        assert total == len(whole) and sum(sizes) == total
        buf[off:off + len(out)] = out
    assert bytes(buf) == whole
    assert res[0][1] == 0 and res[1][1] == res[0][3][0]


def test_partition_properties_random():
    """Contiguous, covering, and no rank is more than one member over its fair share of compressed bytes."""
    import random
    from archive_amd.sharding import partition_members
    rnd = random.Random(12)
    for _ in range(300):
        n = rnd.randrange(0, 200)
        sizes = [rnd.choice((1, 20, 300, 65536, rnd.randrange(1, 100000))) for _ in range(n)]
        w = rnd.randrange(1, 17)
        r = partition_members(sizes, w)
        assert len(r) == w and r[0][0] == 0 and r[-1][1] == n
        assert all(lo <= hi for lo, hi in r) and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        if n:
            fair, biggest = sum(sizes) / w, max(sizes)
            for lo, hi in r:
                assert sum(sizes[lo:hi]) <= fair + biggest + 1e-9, (sizes, w, r)
