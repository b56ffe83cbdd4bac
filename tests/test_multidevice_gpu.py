"""One process, several device contexts (ahip_init_devices): the host-pointer gzip entry point partitions a BGZF stream
over the contexts and must return exactly what the single-device path (and the oracle) returns -- also for the inputs it
cannot partition and hands back to the exact path.  A one-GPU box runs the contexts on the same device
(AHIP_FAKE_DEVICES); a box with two GPUs also runs the two-rank RCCL form of the benchmark's strong-scaling leg."""
import os
import subprocess
import sys
import zlib

import pytest

from tests import streams

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_process_three_contexts(native_built, monkeypatch):
    import archive_amd
    from archive_amd import _native as N
    from oracle import pyoracle as orc
    from tools import corpus
    L = N.lib()
    assert L.ahip_init(0) == 0
    monkeypatch.setenv("AHIP_FAKE_DEVICES", "3")
    assert L.ahip_init_devices(1) == 0, N.last_error()
    assert L.ahip_device_count() == 3
    try:
        comp, plain = corpus.make_gzip(n_members=300, want_plain=True)  # 300 BGZF members, 19 MB out
        comp, plain = bytes(comp), bytes(plain)
        dec = archive_amd.GZipDecoder()
        assert dec.decode_bytes(comp) == plain and dec.last_status == 0           # sharded over the three contexts
        assert L.ahip_debug_last_shards() == 3
        # what the sharded path must hand back to the exact one
        far = streams.gz_wrap(streams.raw_far_reference())                        # q8: reaches into the member in front of it
        third = comp.find(b"\x1f\x8b\x08\x04", len(comp) // 3)
        cases = {"far_member_in_the_middle": comp[:third] + far + comp[third:],
                 "trailing_garbage": comp + b"junk",
                 "plain_gzip_members": comp + streams.gz_member(streams.text(50000, 3)),
                 "truncated": comp[:-5]}
        lying = bytearray(comp)
        lying[third - 4] ^= 0x40                                                   # ISIZE of one member lies
        cases["lying_isize"] = bytes(lying)
        for name, c in cases.items():
            st, out = orc.gzip_decode(c, cap=len(plain) + (1 << 20))
            try:
                got = (0, dec.decode_bytes(c))
                got = (dec.last_status, got[1])
            except archive_amd.errors.RangeError:
                got = (2, None)
            assert got == ((2, None) if st == 2 else (st, out)), name
            assert L.ahip_debug_last_shards() == 1, name
    finally:
        monkeypatch.delenv("AHIP_FAKE_DEVICES")
        assert L.ahip_init_devices(1) == 0 and L.ahip_device_count() == 1


def test_two_ranks_one_stream_rccl(native_built):
    """bench.py's strong-scaling leg on two GPUs: one stream, each rank decodes its slice, shard CRCs combine."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--members", "4096", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["check"]["ok"] and line["strong"]["check"]["ok"]
