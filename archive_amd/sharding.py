"""Multi-GPU sharding of a multi-member gzip stream: one process per GPU, members partitioned
into contiguous ranges, no data-path collective.  The path's only exchange step is the output
size prefix-scan: every rank all-gathers its decoded byte count (8 bytes per rank; RCCL over xGMI
when the tensors live on the GPU, gloo on CPU) and takes the exclusive sum as the offset of its
shard in the logical concatenated output.

The reference has no counterpart (it is single-threaded); what must hold is that the
concatenation of the shards at those offsets equals `GZipDecoder().decodeBytes(whole stream)`.
"""
import ctypes

import numpy as np
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


class _OffsetsExchange:
    """The size all-gather in flight (exchange_output_offsets_begin): `result()` waits for it."""

    def __init__(self, local, sizes, work, rank):
        self.local, self.sizes, self.work, self.rank = int(local), sizes, work, rank

    def result(self):
        if self.sizes is None:
            return 0, self.local, [self.local]
        if self.work is not None:
            self.work.wait()
        sizes_l = [int(v) for v in self.sizes.tolist()]
        return sum(sizes_l[:self.rank]), sum(sizes_l), sizes_l


def exchange_output_offsets_begin(local_out_bytes, device=None, group=None):
    """Starts the size all-gather and returns at once: a shard's output size is known from the member index (the ISIZE
    trailers: trusted, then verified by the decode) BEFORE its members are decoded, so the exchange can run under the inflate
    kernels instead of behind them.  `.result()` = what exchange_output_offsets returns."""
    if not (dist.is_available() and dist.is_initialized()):
        return _OffsetsExchange(local_out_bytes, None, None, 0)
    world = dist.get_world_size(group)
    mine = torch.tensor([int(local_out_bytes)], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    work = dist.all_gather_into_tensor(sizes, mine, group=group, async_op=True)
    return _OffsetsExchange(local_out_bytes, sizes, work, dist.get_rank(group))


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
            # the index already knows the shard's size (ISIZE trailers): the exchange runs under the inflate kernels
            ex = exchange_output_offsets_begin(out_bytes.value, device=d_in.device, group=group)
            _check(L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), stream))
            n = ctypes.c_size_t()
            _check(L.ahip_gzip_plan_status(plan, ctypes.byref(n)))
        finally:
            L.ahip_gzip_plan_destroy(plan)
        offset, total, _ = ex.result()
        if n.value != out_bytes.value:  # (cannot happen with status 0: plan_status reports an index that disagrees with the data)
            offset, total, _ = exchange_output_offsets(n.value, device=d_in.device, group=group)
        return d_out, n.value, offset, total


# ---- ONE long member across ranks (include/archive_hip.h: ahip_stream_split_*) ----
class StreamSplit:
    """One rank's part of the chunked decode of ONE long DEFLATE stream: a thin wrapper of the C-ABI handle.  Every rank
    holds the whole compressed stream (`d_in`, uint8 CUDA tensor; the DEFLATE data starts at byte `data_off`); the phases
    alternate with three all-gathers the caller performs (ShardedStreamDecoder below; the tests also drive several
    handles from one process).  The loop it spreads out is the reference's block loop, zlib/inflate.dart:104-156."""

    def __init__(self, d_in, data_off, rank, world, stream=None):
        self.L = N.lib()
        self.d_in = d_in  # (kept alive: the handle points into it)
        self.h = ctypes.c_void_p()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if stream is None and d_in.is_cuda else ctypes.c_void_p(stream)
        _check(self.L.ahip_stream_split_create(d_in.data_ptr(), d_in.numel(), int(data_off), int(rank), int(world), st, ctypes.byref(self.h)))
        self.n_own = 0

    def candidates(self):
        n = ctypes.c_size_t()
        buf = np.zeros(65536, dtype=np.uint64)
        _check(self.L.ahip_stream_split_candidates(self.h, buf.ctypes.data, buf.size, ctypes.byref(n)))
        self.n_own = n.value
        return buf[:n.value].copy()

    def size(self, all_cand):
        all_cand = np.ascontiguousarray(all_cand, dtype=np.uint64)
        res = np.zeros(4 * max(self.n_own, 1), dtype=np.uint64)
        handled = ctypes.c_int32()
        _check(self.L.ahip_stream_split_size(self.h, all_cand.ctypes.data, all_cand.size, res.ctypes.data, res.size, ctypes.byref(handled)))
        return bool(handled.value), res[:4 * self.n_own].copy()

    def chain(self, all_results):
        all_results = np.ascontiguousarray(all_results, dtype=np.uint64)
        handled = ctypes.c_int32()
        off, ln, total, end = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(self.L.ahip_stream_split_chain(self.h, all_results.ctypes.data, all_results.size // 4, ctypes.byref(handled), ctypes.byref(off),
                                              ctypes.byref(ln), ctypes.byref(total), ctypes.byref(end)))
        return bool(handled.value), off.value, ln.value, total.value, end.value

    def resolve(self):
        d_map = torch.empty(self.L.ahip_stream_split_map_bytes(), dtype=torch.uint8, device=self.d_in.device)
        _check(self.L.ahip_stream_split_resolve(self.h, d_map.data_ptr()))
        return d_map

    def finish(self, d_maps, d_out):
        n, handled = ctypes.c_size_t(), ctypes.c_int32()
        _check(self.L.ahip_stream_split_finish(self.h, d_maps.data_ptr(), d_out.data_ptr(), d_out.numel(), ctypes.byref(n), ctypes.byref(handled)))
        return bool(handled.value), n.value

    def close(self):
        if self.h:
            self.L.ahip_stream_split_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _all_gather_u64(arr, device, group):
    """Variable-length uint64 lists of all ranks, concatenated in rank order (two all-gathers: the counts, the padded lists)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    if not (dist.is_available() and dist.is_initialized()):
        return arr
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, torch.tensor([arr.size], dtype=torch.int64, device=device), group=group)
    counts = [int(c) for c in counts.tolist()]
    width = max(max(counts), 1)
    mine = torch.zeros(width, dtype=torch.int64, device=device)
    if arr.size:
        mine[:arr.size] = torch.from_numpy(arr.view(np.int64)).to(device)
    everybody = torch.zeros(world * width, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(everybody, mine, group=group)
    rows = everybody.cpu().numpy().view(np.uint64).reshape(world, width)
    return np.concatenate([rows[r, :counts[r]] for r in range(world)]) if sum(counts) else np.zeros(0, dtype=np.uint64)


class ShardedStreamDecoder:
    """ONE long DEFLATE stream (a single gzip member, a zlib stream, a raw stream) decoded by all ranks of a process group:
    every rank passes the whole compressed stream and receives its slice of the output and where it lies.  The collectives run
    on `collective_device` (default: the stream's device = RCCL under the `nccl` backend; "cpu" for gloo): two gathers of a few
    words per block start and one of 64 KiB per rank.  Not handled (too short, damaged, ...) = every rank learns so at the same
    step and rank 0 decodes the stream alone with the exact single-device path; the other ranks' slices are empty."""

    def __init__(self, device_index=None, collective_device=None):
        self.device_index = torch.cuda.current_device() if device_index is None else device_index
        self.collective_device = collective_device
        rc = N.lib().ahip_init(self.device_index)
        if rc != 0:
            _check(rc)
        self.last_handled = None

    def decode(self, d_in, data_off=0, group=None, fallback=True):
        """Returns (d_out, n, offset, total, end_pos): this rank's n bytes at `offset` of the stream's `total`; end_pos = the
        reference's stream position behind the DEFLATE data (inflate.dart:104-156), None on the fallback (fallback=False:
        returns None instead of decoding on one rank)."""
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        world = dist.get_world_size(group) if on else 1
        cdev = self.collective_device or d_in.device
        sp = StreamSplit(d_in, data_off, rank, world)
        try:
            all_cand = _all_gather_u64(sp.candidates(), cdev, group)
            handled, res = sp.size(all_cand)
            if handled:
                handled, offset, n, total, end_pos = sp.chain(_all_gather_u64(res, cdev, group))
            if handled:
                d_map = sp.resolve()
                if on:
                    mine = d_map.to(cdev)
                    maps = torch.empty(world * mine.numel(), dtype=torch.uint8, device=cdev)
                    dist.all_gather_into_tensor(maps, mine, group=group)
                    maps = maps.to(d_in.device)
                else:
                    maps = d_map
                d_out = torch.empty(n + 64, dtype=torch.uint8, device=d_in.device)
                handled, n = sp.finish(maps, d_out)
            self.last_handled = handled
            if handled:
                return d_out, n, offset, total, end_pos
        finally:
            sp.close()
        if not fallback:
            return None
        return self._one_rank(d_in, cdev, group, rank, raw_from=data_off)

    def _one_rank(self, d_in, cdev, group, rank, raw_from=None):
        """The exact single-device path on rank 0 (every malformed input keeps the reference's verdict there); the other ranks'
        slices are empty.  raw_from: a raw DEFLATE stream from that byte on; None: the buffer is a gzip stream."""
        if rank != 0:
            total = exchange_output_offsets(0, device=cdev, group=group)[1]
            return torch.empty(0, dtype=torch.uint8, device=d_in.device), 0, total, total, None
        from .codecs import GZipDecoder, Inflate
        host = bytes(d_in.cpu().numpy())
        out = Inflate(host[raw_from:]).get_bytes() if raw_from is not None else GZipDecoder().decode_bytes(host)
        exchange_output_offsets(len(out), device=cdev, group=group)
        d_out = torch.frombuffer(bytearray(out), dtype=torch.uint8).to(d_in.device) if out else torch.empty(0, dtype=torch.uint8, device=d_in.device)
        return d_out, len(out), 0, len(out), None

    @staticmethod
    def gzip_data_offset(head):
        """Where the DEFLATE data of a gzip member starts -- the reference's `_readHeader`, _gzip_decoder_web.dart:59-139:
        signature, method 8, flags, then the optional extra field, name, comment and header CRC are skipped.  None: no gzip
        header here (the reference falls back to zlib) or it does not end inside `head`."""
        if len(head) < 10 or head[0] != 0x1f or head[1] != 0x8b or head[2] != 8:
            return None
        flags, pos = head[3], 10
        if flags & 4:
            if pos + 2 > len(head):
                return None
            pos += 2 + (head[pos] | (head[pos + 1] << 8))
        for bit in (8, 16):  # name, comment: zero-terminated
            if flags & bit:
                end = head.find(b"\0", pos)
                if end < 0:
                    return None
                pos = end + 1
        if flags & 2:
            pos += 2
        return pos if pos <= len(head) else None

    def decode_gzip(self, d_in, group=None):
        """`GZipDecoder().decodeBytes` (_gzip_decoder_web.dart:19-55) of a stream that is ONE member, by all ranks together:
        the header is skipped like `_readHeader` does, the DEFLATE data split over the ranks, and the member must end eight
        bytes (CRC-32, ISIZE: read, never checked by the reference) in front of the end of the buffer.  Anything else -- no gzip
        header, more members or other bytes behind the first, a stream the split does not take -- is decoded by rank 0 alone
        with the reference's exact member loop.  Returns what `decode` returns."""
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        cdev = self.collective_device or d_in.device
        data_off = self.gzip_data_offset(bytes(d_in[:1 << 17].cpu().numpy()))
        if data_off is not None and data_off < d_in.numel():
            res = self.decode(d_in, data_off, group, fallback=False)
            if res is not None and res[4] == d_in.numel() - 8:
                return res
        self.last_handled = False
        return self._one_rank(d_in, cdev, group, rank)


# ---- Deflate and BZip2, one process per GPU (the one-process forms are ahip_deflate_shards / ahip_bzip2_decode_shards) ----
class ShardedDeflate:
    """`Deflate(bytes, level: L, windowBits: W).getBytes()` (deflate.dart:39-48) of an input cut over the ranks
    (`partition_bytes`): every rank compresses its piece on its own GPU; the pieces, laid end to end at the offsets of one
    all-gather of sizes, are ONE raw DEFLATE stream of the whole input (every piece but the last ends with the reference's own
    flush marker, deflate.dart:219)."""

    def __init__(self, device_index=None, collective_device=None):
        self.device_index = torch.cuda.current_device() if device_index is None else device_index
        self.collective_device = collective_device
        rc = N.lib().ahip_init(self.device_index)
        if rc != 0:
            _check(rc)

    def encode_piece(self, d_piece, level=6, window_bits=15, group=None):
        """d_piece: this rank's bytes (uint8 CUDA tensor, may be empty).  Returns (d_out, n, offset, total, crc32 of the WHOLE
        input): n compressed bytes that belong at `offset` of the stream's `total`."""
        L = N.lib()
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        world = dist.get_world_size(group) if on else 1
        cdev = self.collective_device or d_piece.device
        n_in = d_piece.numel()
        d_out = torch.empty(L.ahip_deflate_bound(n_in) + 64, dtype=torch.uint8, device=d_piece.device)
        olen, crc = ctypes.c_size_t(), ctypes.c_uint32()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _check(L.ahip_deflate_piece_device(d_piece.data_ptr() if n_in else None, n_in, level, window_bits, 1 if rank == world - 1 else 0,
                                           d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), ctypes.byref(crc), stream))
        offset, total, _ = exchange_output_offsets(olen.value, device=cdev, group=group)
        whole_crc = crc.value
        if on:  # the gzip trailer's CRC: the pieces' CRCs combined over GF(2) in rank order
            mine = torch.tensor([crc.value, n_in], dtype=torch.int64, device=cdev)
            rows = torch.zeros(2 * world, dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(rows, mine, group=group)
            rows = rows.view(world, 2).tolist()
            whole_crc = 0
            for c, ln in rows:
                whole_crc = crc32_combine(whole_crc, int(c), int(ln))
        return d_out, olen.value, offset, total, whole_crc


def merge_bzip2_ranks(rows, verify):
    """The merge of ahip_bzip2_decode_shards on gathered rows, rank order = stream order (bzip2_decoder.dart:20-88).
    rows[r] = (status, bytes, blocks folded, fold, saw_eos, eos_stored, stopped, first, next).
    Returns (rank that must run again or None, candidate it must start at, verdict, bytes counted per rank)."""
    worst, ended, saw_eos, eos_stored, combined = 0, False, False, 0, 0
    stands = None
    counted = []
    for r, (st, nbytes, nblocks, fold, eos, stored, stopped, first, nxt) in enumerate(rows):
        if ended:
            counted.append(0)
            continue
        if r > 0 and stands is not None and st != N.AHIP_E_DEVICE and first != stands:
            return r, stands, None, None  # what rank r decoded began at a non-block: again from where the chain stands
        stands = nxt
        counted.append(nbytes)
        rot = nblocks & 31
        combined = ((((combined << rot) | (combined >> (32 - rot))) & 0xffffffff) if rot else combined) ^ fold
        if st != 0:
            worst, ended = st, True
        elif stopped:
            ended, saw_eos, eos_stored = True, bool(eos), stored
    if saw_eos and verify and eos_stored != combined and worst == 0:
        worst = N.AHIP_FALSE
    return None, None, worst, counted


class ShardedBZip2Decoder:
    """`BZip2Decoder().decodeBytes(data, verify)` (bzip2_decoder.dart:13-88) with the blocks of the stream spread over the ranks:
    every rank holds the whole compressed stream and decodes the blocks among its range of the block-magic candidates; one
    all-gather of nine words per rank carries what the merge needs (the chain of blocks, the CRC folds, the verdicts); a rank
    whose range began on a false magic (inside another block's data) runs once more from where the chain really stands."""

    def __init__(self, device_index=None, collective_device=None):
        self.device_index = torch.cuda.current_device() if device_index is None else device_index
        self.collective_device = collective_device
        rc = N.lib().ahip_init(self.device_index)
        if rc != 0:
            _check(rc)
        self.reruns = 0

    def decode(self, d_in, out_cap, verify=False, group=None):
        """Returns (d_out, n, offset, total, status): this rank's n bytes at `offset` of the `total` the stream decodes to;
        status = what ahip_bzip2_decode_device returns for the stream (0 true, 1 false, 2 RangeError)."""
        L = N.lib()
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        world = dist.get_world_size(group) if on else 1
        cdev = self.collective_device or d_in.device
        d_out = torch.empty(out_cap + 64, dtype=torch.uint8, device=d_in.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        start, row = 0xffffffffffffffff, None
        for _ in range(world + 1):
            if row is None:  # (the first time, and again only on the rank the merge sends back)
                olen = ctypes.c_size_t()
                info = (ctypes.c_uint64 * 8)()
                st = L.ahip_bzip2_decode_range_device(d_in.data_ptr(), d_in.numel(), 1 if verify else 0, rank, world, start, d_out.data_ptr(), out_cap,
                                                      ctypes.byref(olen), info, stream)
                self.last_error = N.last_error() if st < 0 else ""
                row = [st, olen.value, info[0], info[1], info[2], info[3], info[4], info[5] & 0x7fffffffffffffff, info[6] & 0x7fffffffffffffff]
            if on:
                mine = torch.tensor(row, dtype=torch.int64, device=cdev)
                rows = torch.zeros(9 * world, dtype=torch.int64, device=cdev)
                dist.all_gather_into_tensor(rows, mine, group=group)
                rows = [tuple(int(v) for v in r) for r in rows.view(world, 9).tolist()]
            else:
                rows = [tuple(row)]
            again, at, verdict, counted = merge_bzip2_ranks(rows, verify)
            if again is None:
                if verdict < 0:  # (every rank sees the same verdict: all raise)
                    _check(verdict)
                return d_out, counted[rank], sum(counted[:rank]), sum(counted), verdict
            self.reruns += 1
            if again == rank:
                start, row = at, None
        raise RuntimeError("bzip2 merge did not settle")
