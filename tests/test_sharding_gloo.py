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


def _deflate_worker(rank, world, port, data, ranges, q):
    """One rank of a sharded Deflate: its byte range compressed on its own, closed by the byte-aligning empty stored
    block (Z_SYNC_FLUSH: the marker the reference itself writes, deflate.dart:219) unless it is the last; sizes exchanged."""
    import zlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from archive_amd.sharding import exchange_output_offsets
        lo, hi = ranges[rank]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        piece = co.compress(data[lo:hi])
        piece += co.flush(zlib.Z_FINISH if rank == world - 1 else zlib.Z_SYNC_FLUSH)
        if hi == lo and rank != world - 1:
            piece = b""  # an empty shard in the middle adds nothing
        off, total, sizes = exchange_output_offsets(len(piece))
        q.put((rank, off, total, sizes, piece, zlib.crc32(data[lo:hi]), hi - lo))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_deflate_splices_into_one_stream():
    """ahip_deflate_shards' contract on CPU: byte ranges cut on chunk multiples, every piece but the last ends on the
    flush marker, pieces laid end to end at the exchanged offsets inflate (through the reference's Inflate: the oracle) to
    the whole input; the per-shard CRCs combine to the whole input's."""
    import zlib
    from archive_amd.sharding import crc32_combine, partition_bytes
    from oracle import pyoracle
    data = streams.text(700000, 3) + bytes(100000) + streams.text(300001, 4)
    for n_bytes in (len(data), 40000, 0):
        d = data[:n_bytes]
        ranges = partition_bytes(len(d), 2)
        assert ranges[0][0] == 0 and ranges[-1][1] == len(d) and ranges[0][1] == ranges[1][0] and ranges[0][1] % 32768 == 0
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_deflate_worker, args=(r, 2, port, d, ranges, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        buf = bytearray(res[0][2])
        crc = 0
        for rank, off, total, sizes, piece, c, n in res:
            buf[off:off + len(piece)] = piece
            crc = crc32_combine(crc, c, n)
        st, out, pos = pyoracle.inflate_raw(bytes(buf), cap=len(d) + 64)
        assert st == 0 and out == d and pos == len(buf)
        assert crc == zlib.crc32(d)


def _bz_candidates(stream):
    """(bit position, kind) of every block / end-of-stream magic of a bzip2 stream, and the 32-bit field behind it."""
    bits = int.from_bytes(stream, "big")
    nbits = len(stream) * 8
    out = []
    for p in range(32, nbits - 80 + 1):
        v = (bits >> (nbits - p - 48)) & ((1 << 48) - 1)
        if v in (0x314159265359, 0x177245385090):
            crc = (bits >> (nbits - p - 80)) & 0xffffffff
            out.append((p, 1 if v == 0x314159265359 else 2, crc))
    return out


def _bz_worker(rank, world, port, cands, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from archive_amd.sharding import fold_block_crcs, partition_blocks
        lo, hi = partition_blocks(len(cands), world)[rank]
        mine = [c for c in cands[lo:hi] if c[1] == 1]
        rep = torch.tensor([len(mine), fold_block_crcs([c[2] for c in mine])], dtype=torch.int64)
        allr = torch.zeros(2 * world, dtype=torch.int64)
        dist.all_gather_into_tensor(allr, rep)
        q.put((rank, [int(v) for v in allr.tolist()]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_bzip2_block_ranges_and_crc_fold():
    """ahip_bzip2_decode_shards' merge rule on CPU: block magics found by every rank alike, candidate ranges
    [K r / n, K (r + 1) / n), and the stream CRC (rotl-xor over the blocks, bzip2_decoder.dart:77-78) recovered from the
    ranks' own folds -- it has to equal the combined CRC the stream stores behind its end-of-stream magic."""
    import bz2
    from archive_amd.sharding import merge_block_folds, partition_blocks
    data = streams.text(950000, 6)
    stream = bz2.compress(data, 1)  # 100 k blocks: ten of them
    cands = _bz_candidates(stream)
    assert [c[1] for c in cands].count(1) >= 9 and cands[-1][1] == 2 and cands[0][0] == 32
    for w in (2, 3):
        r = partition_blocks(len(cands), w)
        assert r[0][0] == 0 and r[-1][1] == len(cands) and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bz_worker, args=(r, 2, port, cands, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]
    flat = res[0][1]
    assert merge_block_folds([(flat[0], flat[1]), (flat[2], flat[3])]) == cands[-1][2]


# ---- one long member across ranks: the host side of ahip_stream_split_* ----
def test_stream_split_chain_walk_on_plain_arrays(native_built):
    """ahip_stream_split_chain's walk (no device): six block starts of which two are false finds (they decode garbage and
    nothing ends on them), three ranks owning two candidates each.  The chain 0 -> 2 -> 3 -> 5 gives every rank its slice of
    the output: offsets are the sums in front, a rank whose candidates are all off the chain gets an empty slice where the
    chain passes its range, the total and the end position are everybody's."""
    import ctypes
    from archive_amd import _native as N
    L = N.lib()
    OK, CHUNK_END = 0, 19
    cand = np.array([80, 1000, 2000, 3000, 3500, 4000], dtype=np.uint64)
    res = np.array([[CHUNK_END, 70000, 2000, 3], [7, 123, 1111, 1], [CHUNK_END, 50000, 3000, 2], [CHUNK_END, 10, 4000, 1],
                    [1, 5, 3600, 1], [OK, 900, 520, 2]], dtype=np.uint64)

    def walk(c0, c1, r=res, c=cand):
        out = np.zeros(6, dtype=np.uint64)
        rc = L.ahip_debug_stream_split_chain(c.ctypes.data, np.ascontiguousarray(r).ctypes.data, len(c), c0, c1, out.ctypes.data)
        return rc, [int(v) for v in out]
    assert walk(0, 2) == (0, [1, 0, 70000, 120910, 520, 1])
    assert walk(2, 4) == (0, [1, 70000, 50010, 120910, 520, 2])
    assert walk(4, 6) == (0, [1, 120010, 900, 120910, 520, 1])
    assert walk(4, 5) == (0, [1, 120010, 0, 120910, 520, 0])      # only the false find: an empty slice where the chain passes
    assert walk(6, 6) == (0, [1, 120910, 0, 120910, 520, 0])      # no candidates at all (a range inside one long block)
    assert walk(0, 6) == (0, [1, 0, 120910, 120910, 520, 4])      # one rank: everything
    bad = res.copy(); bad[3, 0] = 4                                # a chunk of the chain ends with an error status: not for this path
    assert walk(2, 4, bad)[1][0] == 0 and walk(0, 2, bad)[1][0] == 0
    broken = res.copy(); broken[2, 2] = 3001                       # ends where nothing starts: cannot happen (ends ARE candidates)
    assert walk(0, 2, broken)[0] == N.AHIP_E_DEVICE
    back = res.copy(); back[2, 2] = 1000                           # ... or on an earlier candidate
    assert walk(0, 2, back)[0] == N.AHIP_E_DEVICE
    assert walk(0, 2, res[:3], cand[:3])[1][0] == 0                # fewer than four block starts: the one-wave path's


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from archive_amd.sharding import _all_gather_u64
        mine = [np.array([80, 2 ** 63 + 5, 7], dtype=np.uint64), np.zeros(0, dtype=np.uint64)][rank]
        a = _all_gather_u64(mine, "cpu", None)
        b = _all_gather_u64(np.array([rank * 10 + 1, rank * 10 + 2], dtype=np.uint64), "cpu", None)
        c = _all_gather_u64(np.zeros(0, dtype=np.uint64), "cpu", None)
        # the size exchange started ahead of the decode (exchange_output_offsets_begin) gives what the blocking one gives
        from archive_amd.sharding import exchange_output_offsets, exchange_output_offsets_begin
        ex = exchange_output_offsets_begin(1000 + rank, device="cpu")
        assert ex.result() == exchange_output_offsets(1000 + rank, device="cpu") == ([0, 1000][rank], 2001, [1000, 1001])
        q.put((rank, a.tolist(), b.tolist(), c.tolist()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_of_ragged_lists():
    """The two list exchanges of the split decode (block starts, sizing results): lists of different lengths -- one rank's may
    be empty -- concatenated in rank order, values above 2^63 intact."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, a, b, c in got:
        assert a == [80, 2 ** 63 + 5, 7] and b == [1, 2, 11, 12] and c == []


def test_gzip_data_offset_follows_the_reference_header_walk():
    """ShardedStreamDecoder.gzip_data_offset = `_readHeader` (_gzip_decoder_web.dart:59-139): every combination of the optional
    fields (extra, name, comment, header CRC), checked by inflating from the offset it returns."""
    import zlib
    from archive_amd.sharding import ShardedStreamDecoder as D
    payload = b"the quick brown fox " * 50
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(payload) + co.flush()
    for flags in range(32):
        if flags & 1 and flags & ~1 == 0:
            pass  # (FTEXT alone: a hint, nothing to skip)
        h = bytes([0x1f, 0x8b, 8, flags, 1, 2, 3, 4, 0, 255])
        if flags & 4:
            h += (8).to_bytes(2, "little") + b"\0extra\0\1"       # (zero bytes inside the extra field do not end anything)
        if flags & 8:
            h += b"name.txt\0"
        if flags & 16:
            h += b"a comment\0"
        if flags & 2:
            h += b"\xaa\xbb"
        off = D.gzip_data_offset(h + raw + bytes(8))
        assert off == len(h), (flags, off, len(h))
        assert zlib.decompress((h + raw)[off:], -15) == payload
    assert D.gzip_data_offset(b"\x1f\x8b\x07" + bytes(20)) is None        # not deflate
    assert D.gzip_data_offset(b"\x78\x9c" + bytes(20)) is None            # a zlib stream: the reference falls back to ZLibDecoder
    assert D.gzip_data_offset(bytes([0x1f, 0x8b, 8, 8, 0, 0, 0, 0, 0, 3]) + b"never ends") is None
    assert D.gzip_data_offset(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 3, 200, 0]) + bytes(50)) is None  # extra field runs past the buffer


def test_bzip2_rank_merge_follows_the_block_loop():
    """merge_bzip2_ranks = the merge of ahip_bzip2_decode_shards on gathered rows (decodeStream's loop, bzip2_decoder.dart:20-88):
    ranks in stream order, a rank whose chain began somewhere else than the ranks in front expect goes again, the first
    verdict that is not OK or the end-of-stream block ends the stream (what lies behind counts for nothing), the stream CRC is
    the rotl-xor fold of the ranks' folds."""
    from archive_amd import _native as N
    from archive_amd.sharding import fold_block_crcs, merge_bzip2_ranks
    crcs = [0x11111111, 0x22222222, 0x80000001, 0x44444444, 0x55555555]
    whole = fold_block_crcs(crcs)

    def row(st, nbytes, blk, eos, stored, stopped, first, nxt):
        return (st, nbytes, len(blk), fold_block_crcs(blk), eos, stored, stopped, first, nxt)
    # three ranks: 2 + 2 + 1 blocks, the last meets the end-of-stream block
    rows = [row(0, 200, crcs[0:2], 0, 0, 0, 0, 2), row(0, 150, crcs[2:4], 0, 0, 0, 2, 4), row(0, 70, crcs[4:5], 1, whole, 1, 4, 6)]
    assert merge_bzip2_ranks(rows, True) == (None, None, 0, [200, 150, 70])
    bad = list(rows); bad[2] = row(0, 70, crcs[4:5], 1, whole ^ 1, 1, 4, 6)
    assert merge_bzip2_ranks(bad, True)[2] == N.AHIP_FALSE and merge_bzip2_ranks(bad, False)[2] == 0   # the stored CRC only matters with verify
    # rank 1 began on a false magic (candidate 2, the chain expects 3): it must go again from 3
    off = list(rows); off[0] = row(0, 260, crcs[0:2], 0, 0, 0, 0, 3)
    assert merge_bzip2_ranks(off, True)[:2] == (1, 3)
    # the stream ends inside rank 1 (`false` in its second block): rank 2 counts for nothing
    stop = [rows[0], row(1, 90, crcs[2:3], 0, 0, 1, 2, 4), rows[2]]
    assert merge_bzip2_ranks(stop, True) == (None, None, 1, [200, 90, 0])
    # the end-of-stream block inside rank 0: everything behind is another stream (the reference decodes ONE)
    early = [row(0, 200, crcs[0:2], 1, fold_block_crcs(crcs[0:2]), 1, 0, 3), rows[1], rows[2]]
    assert merge_bzip2_ranks(early, True) == (None, None, 0, [200, 0, 0])
    # an error on a rank is the stream's verdict
    assert merge_bzip2_ranks([rows[0], row(N.AHIP_E_CAP, 0, [], 0, 0, 0, 2, 2), rows[2]], False)[2] == N.AHIP_E_CAP
