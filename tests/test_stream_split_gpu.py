"""ONE long DEFLATE stream decoded by several ranks (include/archive_hip.h: ahip_stream_split_*; host:
archive_amd/sharding.py::StreamSplit / ShardedStreamDecoder).  The ranks' slices, laid side by side at the offsets the
chain reports, must be the bytes the reference's Inflate produces (zlib here: the streams are valid; the damaged ones are
compared with the oracle through the fallback), whatever the number of ranks.

Two forms: several handles driven from ONE process (the gathers are numpy concatenations: every world size on the one GPU
there is, deterministic), and two real processes on one device with the collectives on gloo -- the code a multi-GPU
launch runs, RCCL in place of gloo."""
import gzip
import os
import random
import subprocess
import sys
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for o in range(0, len(data), flush_every):
        out += c.compress(data[o:o + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)
    return out + c.flush()


@pytest.fixture(scope="module")
def corpora():
    from tools import corpus
    rnd = random.Random(33)
    log = bytes(corpus.text(corpus.LOG, 4321, 0, 10 << 20))
    # 6 000 fresh bytes, then the 14 000 that lie 20 000 back, again and again: every match reaches 20 000 bytes back, across
    # every chunk and rank edge, and what it copies is itself a copy from further back (chains through the window maps)
    buf = bytearray(rnd.randbytes(20000))
    while len(buf) < (12 << 20):
        buf += rnd.randbytes(6000)
        buf += buf[-20000:-6000]
    period = bytes(buf)
    mixed = b"".join(bytes(corpus.text(corpus.LOG, 78, i, 400000)) + rnd.randbytes(150000) for i in range(14))  # dynamic + stored blocks
    return {"log": log, "period": period, "mixed": mixed}


def split_decode(raw, world, data_off=0, tail=b""):
    """All ranks of a world, one after another in this process.  Returns (bytes or None when not handled, end_pos, per-rank
    (offset, length))."""
    import torch
    from archive_amd.sharding import StreamSplit
    d_in = torch.frombuffer(bytearray(raw + tail), dtype=torch.uint8).cuda()
    sps = [StreamSplit(d_in, data_off, r, world) for r in range(world)]
    try:
        all_cand = np.concatenate([sp.candidates() for sp in sps])
        sized = [sp.size(all_cand) for sp in sps]
        assert len({h for h, _ in sized}) == 1, "the ranks disagree on whether the stream is handled"
        if not sized[0][0]:
            return None, None, []
        all_res = np.concatenate([r for _, r in sized])
        chains = [sp.chain(all_res) for sp in sps]
        assert len({c[0] for c in chains}) == 1
        if not chains[0][0]:
            return None, None, []
        total, end_pos = chains[0][3], chains[0][4]
        assert all(c[3] == total and c[4] == end_pos for c in chains)
        maps = torch.cat([sp.resolve() for sp in sps])
        out = bytearray(total)
        where, at = [], 0
        for sp, (_, off, n, _, _) in zip(sps, chains):
            assert off == at, "the slices are not back to back"
            d_out = torch.full((n + 64,), 0xA5, dtype=torch.uint8, device="cuda")
            handled, got = sp.finish(maps, d_out)
            if not handled:
                return None, None, []
            assert got == n and bool((d_out[n:] == 0xA5).all())
            out[off:off + n] = bytes(d_out[:n].cpu().numpy())
            where.append((off, n))
            at += n
        assert at == total
        return bytes(out), end_pos, where
    finally:
        for sp in sps:
            sp.close()


def test_every_world_size_gives_the_same_bytes(native_built, corpora):
    from archive_amd import _native as N
    assert N.lib().ahip_init(0) == 0
    for name, data in corpora.items():
        raw = _raw(data)
        for world in (1, 2, 3, 8):
            got, end_pos, where = split_decode(raw, world)
            assert got is not None, (name, world)
            assert got == data, (name, world)
            assert end_pos == len(raw), (name, world, end_pos, len(raw))
            if world > 1:
                assert sum(1 for _, n in where if n > 0) >= 2, (name, world, where)


def test_levels_flush_points_and_offsets(native_built, corpora):
    """Other block structures (level 1 and 9, full-flush points every 300 000 bytes), the DEFLATE data not at byte 0 of the
    buffer (a gzip header in front, a trailer behind: end_pos is the reference's stream position in front of the trailer)."""
    data = corpora["log"]
    for kw in (dict(level=1), dict(level=9), dict(flush_every=300000)):
        raw = _raw(data, **kw)
        got, end_pos, _ = split_decode(raw, 3)
        assert got == data and end_pos == len(raw), kw
    g = gzip.compress(data, 6, mtime=0)
    got, end_pos, _ = split_decode(g, 4, data_off=10)
    assert got == data and end_pos == len(g) - 8
    junk = b"\x00" * 7 + _raw(data) + b"trailing bytes"
    got, end_pos, _ = split_decode(junk, 2, data_off=7)
    assert got == data and end_pos == len(junk) - len(b"trailing bytes")


def test_tokenizing_again_gives_the_same_bytes(native_built, corpora, monkeypatch):
    """AHIP_SM_TWO_PASS=1: the sizing run keeps no tokens, every rank tokenizes the chunks of the chain again with exact
    offsets and windows (the way a chunk goes whose kept tokens cannot be used: less than a window in front of it, a token
    area that did not hold) and compares the outcome with the sizing run's."""
    monkeypatch.setenv("AHIP_SM_TWO_PASS", "1")
    for name in ("log", "mixed"):
        raw = _raw(corpora[name])
        got, end_pos, _ = split_decode(raw, 3)
        assert got == corpora[name] and end_pos == len(raw), name


def test_some_chunks_tokenized_again_next_to_kept_ones(native_built, corpora):
    """A chunk whose token area did not hold in the sizing run (it decoded across block starts that are not on the chain) is
    tokenized again with exact offsets while the others use what they kept -- two tokenizer outputs, two resolver launches, one
    symbol array.  AHIP_SPLIT_TEST_RETOK=3 makes every third chunk such a chunk (in a subprocess: the knob is read once)."""
    code = (
        "import sys, zlib; sys.path.insert(0, %r)\n"
        "from tests.test_stream_split_gpu import split_decode, _raw\n"
        "from tools import corpus\n"
        "from archive_amd import _native as N\n"
        "assert N.lib().ahip_init(0) == 0\n"
        "data = bytes(corpus.text(corpus.LOG, 4321, 0, 10 << 20))\n"
        "raw = _raw(data)\n"
        "for world in (1, 3):\n"
        "    got, end_pos, _ = split_decode(raw, world)\n"
        "    assert got == data and end_pos == len(raw), world\n"
        "print('mixed ok')\n" % ROOT)
    env = dict(os.environ, AHIP_SPLIT_TEST_RETOK="3", AHIP_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "mixed ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_what_the_path_does_not_take(native_built, corpora):
    """Too short, no block start to find (a Z_FIXED stream), a truncated or damaged stream: every rank says `not handled` at
    the same step and nothing is written; ShardedStreamDecoder then decodes on one rank with the exact path (verdicts of the
    reference: tests/test_single_stream_gpu.py)."""
    data = corpora["log"]
    assert split_decode(_raw(data[:100000]), 2)[0] is None
    assert split_decode(_raw(data, strategy=zlib.Z_FIXED), 2)[0] is None
    raw = _raw(data)
    assert split_decode(raw[:len(raw) // 2], 2)[0] is None
    b = bytearray(raw)
    b[len(b) * 2 // 3] ^= 0x10
    got = split_decode(bytes(b), 3)[0]
    assert got is None or got != data  # (a flipped bit that still decodes is the reference's verdict too; usually it does not)


def test_gzip_streams_through_the_host_class(native_built, corpora):
    """ShardedStreamDecoder.decode_gzip without a process group (a world of one): a member with every optional header field goes
    through the split; two members, bytes behind the trailer and a zlib stream go to the exact member loop (GZipDecoder) and
    come back with the reference's result for them."""
    import torch
    import archive_amd
    from archive_amd.sharding import ShardedStreamDecoder
    data = corpora["log"]
    raw = _raw(data)
    tail = zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")
    head = bytes([0x1f, 0x8b, 8, 4 | 8 | 16 | 2, 0, 0, 0, 0, 0, 3]) + (5).to_bytes(2, "little") + b"ab\0cd" + b"log.txt\0" + b"synthetic\0" + b"\x12\x34"
    dec = ShardedStreamDecoder(device_index=0)

    def run(stream):
        d_in = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
        d_out, n, off, total, end = dec.decode_gzip(d_in)
        return bytes(d_out[:n].cpu().numpy()), off, total, end, dec.last_handled
    one = head + raw + tail
    assert run(one) == (data, 0, len(data), len(one) - 8, True)
    two = one + gzip.compress(b"second member", mtime=0)
    want = archive_amd.GZipDecoder().decode_bytes(two)
    assert want == data + b"second member"
    assert run(two) == (want, 0, len(want), None, False)
    for stream in (one + b"junk behind the trailer", zlib.compress(data[:3 << 20])):
        want = archive_amd.GZipDecoder().decode_bytes(stream)
        got = run(stream)
        assert got[0] == want and got[4] is False


def test_api_misuse_is_an_error_not_a_crash(native_built, corpora):
    import ctypes
    import torch
    from archive_amd import _native as N
    from archive_amd.sharding import StreamSplit
    from archive_amd import errors
    L = N.lib()
    raw = _raw(corpora["log"])
    d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    h = ctypes.c_void_p()
    assert L.ahip_stream_split_create(d_in.data_ptr(), d_in.numel(), 0, 2, 2, None, ctypes.byref(h)) == N.AHIP_E_ARG  # rank == world
    sp = StreamSplit(d_in, 0, 1, 2)
    own = sp.candidates()
    with pytest.raises(errors.ArchiveHipError):
        sp.size(own[::-1].copy())            # not in stream order
    sp.close()
    sp = StreamSplit(d_in, 0, 1, 2)
    sp.candidates()
    with pytest.raises(errors.ArchiveHipError):
        sp.size(np.array([0, 5, 9, 11, 12], dtype=np.uint64))  # a list this rank's finds are not part of
    sp.close()


_RANK_SCRIPT = r"""
import os, sys, zlib
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from tools import corpus
from archive_amd.sharding import ShardedStreamDecoder

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
data = bytes(corpus.text(corpus.LOG, 99, 0, 9 << 20))
co = zlib.compressobj(6, zlib.DEFLATED, -15)
raw = co.compress(data) + co.flush()
dec = ShardedStreamDecoder(device_index=0, collective_device="cpu")
for case, stream in (("long", raw), ("short", raw[:0] + zlib.compress(data[:50000])[2:-4])):
    d_in = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
    d_out, n, offset, total, end_pos = dec.decode(d_in, 0)
    want = data if case == "long" else data[:50000]
    assert total == len(want), (case, total)
    assert bytes(d_out[:n].cpu().numpy()) == want[offset:offset + n], (case, rank)
    sizes = [None] * world
    dist.all_gather_object(sizes, (offset, n, dec.last_handled, end_pos))
    if rank == 0:
        at = 0
        for off, ln, _, _ in sizes:
            assert off == at or ln == 0, sizes
            at += ln
        assert at == len(want), sizes
        if case == "long":
            assert all(h for _, _, h, _ in sizes) and sum(1 for _, ln, _, _ in sizes if ln) == world and end_pos == len(raw), sizes
        else:
            assert not any(h for _, _, h, _ in sizes) and sizes[0][1] == len(want), sizes
        print("case %%s ok: %%s" %% (case, [(o, l) for o, l, _, _ in sizes]))
dist.destroy_process_group()
"""


def test_two_ranks_one_member_gloo(native_built, tmp_path):
    """Two processes, one device, the three all-gathers on gloo: ShardedStreamDecoder.decode as a launch on two GPUs runs it
    (`collective_device=None` there: RCCL).  A long stream splits into two slices; a short one falls back to rank 0."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29500 + os.getpid() % 2000), str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "case long ok" in r.stdout and "case short ok" in r.stdout, r.stdout[-2000:]


_RCCL_SCRIPT = r"""
import os, sys, zlib
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from tools import corpus
from archive_amd.sharding import ShardedStreamDecoder
world_gpus = int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
gz, crc = corpus.make_one_member(nbytes=48 << 20)      # pigz-style: pieces closed by sync-flush markers, one DEFLATE stream
d_in = torch.from_numpy(gz.copy()).to(dev)
dec = ShardedStreamDecoder(device_index=dev.index)     # collectives on the stream's device: RCCL
d_out, n, off, total, end = dec.decode(d_in, 10)
mine = torch.tensor([zlib.crc32(bytes(d_out[:n].cpu().numpy())), n, off], dtype=torch.int64, device=dev)
rows = torch.zeros(3 * world_gpus, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(rows, mine)
if dist.get_rank() == 0:
    from archive_amd.sharding import crc32_combine
    rows = rows.view(world_gpus, 3).tolist()
    c, at = 0, 0
    for rc, rn, ro in rows:
        assert ro == at
        c = crc32_combine(c, rc, rn) if rn else c
        at += rn
    assert c == crc and at == total == (48 << 20) and end == len(gz) - 8 and dec.last_handled, (rows, total, end)
    print("rccl ranks %%d ok" %% world_gpus)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("ranks", [1, 2])
def test_gathers_over_rccl(native_built, tmp_path, ranks):
    """The three exchanges on GPU tensors through RCCL (`collective_device=None`), the way a multi-GPU launch has them: with one
    rank on the one GPU of this box (the all-gathers of int64 / uint8 device tensors run, a world of one), with two where
    there are two GPUs.  The member is pigz-style: several pieces, each closed by an empty stored block."""
    import torch
    if torch.cuda.device_count() < ranks:
        pytest.skip("needs %d GPUs" % ranks)
    script = tmp_path / "rccl.py"
    script.write_text(_RCCL_SCRIPT % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1",
                        "--master-port", str(31500 + os.getpid() % 2000), str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rccl ranks %d ok" % ranks in r.stdout, r.stdout[-2000:]
