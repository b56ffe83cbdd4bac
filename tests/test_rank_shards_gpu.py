"""gzip members, Deflate and BZip2 sharded one process per GPU (archive_amd/sharding.py: ShardedGZipDecoder -- member ranges, the
size exchange started under the decode --, ShardedDeflate over ahip_deflate_piece_device, ShardedBZip2Decoder over
ahip_bzip2_decode_range_device; the one-process forms are ahip_deflate_shards /
ahip_bzip2_decode_shards, tests/test_multidevice_gpu.py).  Two real processes on the one GPU of this box, collectives on gloo --
the code a launch on two GPUs runs with RCCL in their place -- and the same calls without a process group (a world of one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import bz2, os, sys, zlib
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from tests import streams
from archive_amd import _native as N
from archive_amd.sharding import ShardedBZip2Decoder, ShardedDeflate, partition_bytes

if "RANK" in os.environ:
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
else:
    rank, world = 0, 1
torch.cuda.set_device(0)


def gather(obj):
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


# ---- gzip members: contiguous member ranges, the size exchange started under the decode ----
from archive_amd.sharding import ShardedGZipDecoder, partition_members
plains = [streams.text(20000 + 977 * i, 100 + i) for i in range(48)]
members = [streams.gz_member(p, level=6) for p in plains]
lo, hi = partition_members([len(m) for m in members], world)[rank]
shard = b"".join(members[lo:hi])
gdec = ShardedGZipDecoder(device_index=0)
if world > 1:
    import archive_amd.sharding as _sh
    _begin = _sh.exchange_output_offsets_begin
    _sh.exchange_output_offsets_begin = lambda n, device=None, group=None: _begin(n, device="cpu", group=group)  # (gloo: CPU tensors)
d_sh = torch.frombuffer(bytearray(shard), dtype=torch.uint8).cuda()
d_o, n, off, total = gdec.decode_shard(d_sh)
want = b"".join(plains[lo:hi])
assert bytes(d_o[:n].cpu().numpy()) == want and total == sum(len(p) for p in plains) and off == sum(len(p) for p in plains[:lo]), (rank, n, off, total)
print("gzip ok") if rank == 0 else None

# ---- Deflate: the pieces at their offsets are ONE stream of the whole input ----
data = streams.text(1500000, 11) + bytes(200000) + streams.text(700001, 12)
enc = ShardedDeflate(device_index=0, collective_device="cpu")
for level, cuts in ((6, partition_bytes(len(data), world)), (1, [(0, 100001)] + [(100001, len(data))] * (world - 1) if world > 1 else [(0, len(data))]), (0, partition_bytes(len(data), world))):
    lo, hi = cuts[rank]
    if world > 1 and level == 1 and rank > 1:
        lo = hi  # (more than two ranks: the others' pieces are empty)
    piece = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8).cuda() if hi > lo else torch.empty(0, dtype=torch.uint8, device="cuda")
    d_out, n, off, total, crc = enc.encode_piece(piece, level=level)
    rows = gather((off, n, bytes(d_out[:n].cpu().numpy())))
    assert crc == zlib.crc32(data), (level, crc)
    if rank == 0:
        whole, at = bytearray(total), 0
        for o, ln, b in rows:
            assert o == at
            whole[o:o + ln] = b
            at += ln
        assert at == total and zlib.decompress(bytes(whole), -15) == data, level
print("deflate ok") if rank == 0 else None

# ---- BZip2: the blocks of one stream over the ranks ----
text = streams.text(2600000, 5)
bz = bz2.compress(text, 1)                      # 100k blocks: 26 of them
dec = ShardedBZip2Decoder(device_index=0, collective_device="cpu")
d_in = torch.frombuffer(bytearray(bz), dtype=torch.uint8).cuda()
d_out, n, off, total, status = dec.decode(d_in, len(text) + 4096, verify=True)
rows = gather((off, n, bytes(d_out[:n].cpu().numpy())))
assert status == 0 and total == len(text), (status, total)
if rank == 0:
    assert b"".join(b for _, _, b in rows) == text and all(ln > 0 for _, ln, _ in rows), [(o, l) for o, l, _ in rows]
# damaged: a bit flipped in the second half -- the verdict and the bytes of the unsharded call
bad = bytearray(bz)
bad[len(bad) * 3 // 4] ^= 0x04
d_bad = torch.frombuffer(bad, dtype=torch.uint8).cuda()
import ctypes
ref_out = torch.empty(len(text) + 4096, dtype=torch.uint8, device="cuda")
ref_len = ctypes.c_size_t()
ref_status = N.lib().ahip_bzip2_decode_device(d_bad.data_ptr(), d_bad.numel(), 1, ref_out.data_ptr(), ref_out.numel(), ctypes.byref(ref_len), None)
if ref_status >= 0:
    d_out, n, off, total, status = dec.decode(d_bad, len(text) + 4096, verify=True)
    rows = gather((off, n, bytes(d_out[:n].cpu().numpy())))
    assert status == ref_status and total == ref_len.value, (status, ref_status, total, ref_len.value)
    if rank == 0:
        assert b"".join(b for _, _, b in rows) == bytes(ref_out[:ref_len.value].cpu().numpy())
# truncated in the middle of a block; and two streams back to back (the reference decodes ONE)
for case in (bz[:len(bz) // 2], bz + bz2.compress(b"second stream")):
    d_c = torch.frombuffer(bytearray(case), dtype=torch.uint8).cuda()
    ref_status = N.lib().ahip_bzip2_decode_device(d_c.data_ptr(), d_c.numel(), 0, ref_out.data_ptr(), ref_out.numel(), ctypes.byref(ref_len), None)
    d_out, n, off, total, status = dec.decode(d_c, len(text) + 4096, verify=False)
    rows = gather((off, n, bytes(d_out[:n].cpu().numpy())))
    assert status == ref_status, (status, ref_status)
    if status in (0, 1):  # (2 = the reference throws RangeError: no output is defined)
        assert total == ref_len.value, (total, ref_len.value)
        if rank == 0:
            assert b"".join(b for _, _, b in rows) == bytes(ref_out[:ref_len.value].cpu().numpy())
print("bzip2 ok (second tries: %%d)" %% dec.reruns) if rank == 0 else None
if world > 1:
    dist.destroy_process_group()
"""


@pytest.mark.parametrize("ranks", [1, 2, 3])
def test_deflate_and_bzip2_one_process_per_rank(native_built, tmp_path, ranks):
    script = tmp_path / "ranks.py"
    script.write_text(_SCRIPT % {"root": ROOT})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if ranks == 1:
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1",
               "--master-port", str(33500 + os.getpid() % 2000), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "gzip ok" in r.stdout and "deflate ok" in r.stdout and "bzip2 ok" in r.stdout, r.stdout[-2000:]
