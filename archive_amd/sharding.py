"""Multi-GPU sharding of a multi-member gzip stream: one process per GPU, members partitioned
into contiguous ranges, no data-path collective.  The path's only exchange step is the output
size prefix-scan: every rank all-gathers its decoded byte count (8 bytes per rank; RCCL over xGMI
when the tensors live on the GPU, gloo on CPU) and takes the exclusive sum as the offset of its
shard in the logical concatenated output.

The reference has no counterpart (it is single-threaded); what must hold is that the
concatenation of the shards at those offsets equals `GZipDecoder().decodeBytes(whole stream)`.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _native as N
from .codecs import _check


def partition_members(member_sizes, world_size):
    """Contiguous member ranges [(lo, hi), ...] per rank, balanced on compressed bytes.

    member_sizes: compressed size of every member, in stream order."""
    n = len(member_sizes)
    total = sum(member_sizes)
    bounds, acc, r = [0], 0, 1
    for i, s in enumerate(member_sizes):
        acc += s
        while r < world_size and acc * world_size >= total * r and len(bounds) < world_size:
            bounds.append(i + 1)
            r += 1
    while len(bounds) < world_size:
        bounds.append(n)
    bounds.append(n)
    return [(bounds[i], max(bounds[i], bounds[i + 1])) for i in range(world_size)]


def exchange_output_offsets(local_out_bytes, device=None, group=None):
    """(offset of this rank's shard, total bytes, per-rank sizes) via one all-gather of int64."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0, int(local_out_bytes), [int(local_out_bytes)]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([int(local_out_bytes)], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    excl = torch.cumsum(sizes, 0) - sizes
    sizes_l = [int(v) for v in sizes.tolist()]
    return int(excl[rank].item()), int(sum(sizes_l)), sizes_l


def partition_bytes(n, world_size, align=32768):
    """Contiguous byte ranges [(lo, hi), ...] of an n-byte Deflate input, one per rank, cut on multiples of `align`
    (the encoder's chunk size: a cut there loses nothing; only the 32 KiB of history behind a cut are not used)."""
    units = (n + align - 1) // align
    cuts = [min(n, (units * r // world_size) * align) for r in range(world_size)] + [n]
    return [(cuts[r], max(cuts[r], cuts[r + 1])) for r in range(world_size)]


def partition_blocks(n_candidates, world_size):
    """Candidate (block magic) ranges per rank: [K r / n, K (r + 1) / n) -- the rule ahip_bzip2_decode_shards applies."""
    return [(n_candidates * r // world_size, n_candidates * (r + 1) // world_size) for r in range(world_size)]


def fold_block_crcs(crcs, start=0):
    """bzip2's stream CRC over block CRCs: combined = rotl(combined, 1) ^ crc (bzip2_decoder.dart:77-78)."""
    c = start
    for v in crcs:
        c = (((c << 1) | (c >> 31)) & 0xffffffff) ^ v
    return c


def merge_block_folds(folds):
    """Stream CRC from per-shard (number of blocks, fold started at 0) in stream order: the fold is linear over XOR, a
    shard of m blocks rotates what came before it by m."""
    c = 0
    for m, f in folds:
        r = m & 31
        c = (((c << r) | (c >> (32 - r))) & 0xffffffff if r else c) ^ f
    return c


_GF2_POLY = 0xEDB88320


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def crc32_combine(crc1, crc2, len2):
    """CRC-32 of A + B from crc32(A), crc32(B), len(B) (zlib's crc32_combine): what a gzip trailer over sharded Deflate
    output is built from (ahip_deflate_shards returns the per-shard CRCs)."""
    if len2 <= 0:
        return crc1
    odd = [_GF2_POLY] + [1 << n for n in range(31)]
    even = [_gf2_times(odd, odd[n]) for n in range(32)]
    odd = [_gf2_times(even, even[n]) for n in range(32)]
    while True:
        even = [_gf2_times(odd, odd[n]) for n in range(32)]
        if len2 & 1:
            crc1 = _gf2_times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = [_gf2_times(even, even[n]) for n in range(32)]
        if len2 & 1:
            crc1 = _gf2_times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


class ShardedGZipDecoder:
    """Rank-local decode of one shard (a byte range of whole gzip members) on this rank's GPU."""

    def __init__(self, device_index=None):
        self.device_index = torch.cuda.current_device() if device_index is None else device_index
        rc = N.lib().ahip_init(self.device_index)
        if rc != 0:
            _check(rc)

    def decode_shard(self, d_in, d_out=None, group=None):
        """d_in: uint8 CUDA tensor holding this rank's members.  Returns (d_out, n, offset, total)."""
        L = N.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        plan = ctypes.c_void_p()
        _check(L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), stream, ctypes.byref(plan)))
        try:
            members, out_bytes, payload = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
            L.ahip_gzip_plan_info(plan, ctypes.byref(members), ctypes.byref(out_bytes), ctypes.byref(payload))
            if d_out is None or d_out.numel() < out_bytes.value:
                d_out = torch.empty(out_bytes.value + 64, dtype=torch.uint8, device=d_in.device)
            _check(L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), stream))
            n = ctypes.c_size_t()
            _check(L.ahip_gzip_plan_status(plan, ctypes.byref(n)))
        finally:
            L.ahip_gzip_plan_destroy(plan)
        offset, total, _ = exchange_output_offsets(n.value, device=d_in.device, group=group)
        return d_out, n.value, offset, total
