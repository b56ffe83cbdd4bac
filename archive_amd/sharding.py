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
