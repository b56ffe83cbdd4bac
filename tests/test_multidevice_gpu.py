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
        # members WITHOUT size hints (an ordinary `cat a.gz b.gz`; the reference skips FEXTRA anyway): a sizing pass on one
        # device finds the member boundaries, then the contexts take their slices like above
        nobc, plain2 = corpus.make_gzip(n_members=200, bc=False, want_plain=True, seed=77)
        cases["no_size_hints"] = bytes(nobc)
        sharded = {"plain_gzip_members", "no_size_hints"}
        for name, c in cases.items():
            st, out = orc.gzip_decode(c, cap=len(plain) + (1 << 20))
            try:
                got = (0, dec.decode_bytes(c))
                got = (dec.last_status, got[1])
            except archive_amd.errors.RangeError:
                got = (2, None)
            assert got == ((2, None) if st == 2 else (st, out)), name
            assert L.ahip_debug_last_shards() == (3 if name in sharded else 1), name
        assert dec.decode_bytes(bytes(nobc)) == bytes(plain2)
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
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["check"]["ok"] and line["weak"]["check"]["ok"]
    assert sum(line["config"]["members_per_rank"]) == 4096


def test_two_ranks_on_one_device_gloo(native_built):
    """Every line of the N > 1 path the driver launches on 8 GPUs, on the hardware there is: two ranks of bench.py on
    ONE device, the collectives on CPU tensors (gloo) -- the headline is the strong-scaled ONE stream (partitioned on compressed
    bytes, size exchange every step, shard CRCs combined over GF(2)); the weak leg (every rank its own members) rides along."""
    import json
    # The BARE command, no launcher in front of it: bench.py starts its own ranks (bench.launch_ranks -> torch.distributed.run
    # on 127.0.0.1 and a free port) -- the form `python bench.py --gpus N` takes when a driver runs it like the N = 1 line.
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--steps", "2", "--warmup", "1", "--members", "2048", "--cpu-seconds", "0", "--one-member-mib", "24"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["check"]["ok"]
    om = line["one_member"]  # ONE gzip member decoded by both ranks together (ahip_stream_split_*): two slices, CRCs combine to the trailer's
    assert om["check"]["ok"] and om["split_path_on_every_rank"] and len(om["slice_MiB"]) == 2 and min(om["slice_MiB"]) > 0, om
    assert line["check"]["crc32_combined"] == line["check"]["crc32_expected"]
    assert line["config"]["collectives"].startswith("gloo")
    mpr = line["config"]["members_per_rank"]
    assert sum(mpr) == 2048 and min(mpr) > 0
    assert 0 < line["roofline"]["frac_step"] <= line["roofline"]["frac"]
    wk = line["weak"]
    assert wk["scaling"] == "weak" and wk["check"]["ok"] and wk["n_gpus"] == 2


def _decode_shards(L, N, devs, d_ins, out_caps):
    import ctypes
    import torch
    n = len(d_ins)
    d_outs = [torch.empty(c + 64, dtype=torch.uint8, device="cuda") for c in out_caps]
    arr_dev = (ctypes.c_int32 * n)(*devs)
    arr_in = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_ins])
    arr_len = (ctypes.c_size_t * n)(*[t.numel() for t in d_ins])
    arr_out = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_outs])
    arr_cap = (ctypes.c_size_t * n)(*[t.numel() for t in d_outs])
    out_len = (ctypes.c_size_t * n)()
    offsets = (ctypes.c_uint64 * (n + 1))()
    status = (ctypes.c_int32 * n)()
    rc = L.ahip_gzip_decode_shards(n, arr_dev, arr_in, arr_len, arr_out, arr_cap, out_len, offsets, status)
    return rc, d_outs, list(out_len), list(offsets), list(status)


def test_device_resident_shards(native_built, monkeypatch):
    """ahip_gzip_decode_shards: per-device resident shards of one BGZF stream, decoded by the device contexts, offsets
    from the size exchange.  Three contexts on the one device there is (host sums: RCCL refuses the same device
    twice), then one context, one shard -- the exchange goes through RCCL (a communicator of one rank): the dlopen'd
    ncclCommInitAll / ncclAllGather path runs on this box."""
    import numpy as np
    import torch
    from archive_amd import _native as N
    from archive_amd.sharding import partition_members
    from tools import corpus
    L = N.lib()
    assert L.ahip_init(0) == 0
    comp, plain = corpus.make_gzip(n_members=240, want_plain=True)
    comp_b = bytes(comp)
    # member boundaries along the BC chain
    offs, pos = [0], 0
    while pos < len(comp_b):
        bsize = comp_b[pos + 16] | (comp_b[pos + 17] << 8)
        pos += bsize + 1
        offs.append(pos)
    assert offs[-1] == len(comp_b) and len(offs) == 241
    monkeypatch.setenv("AHIP_FAKE_DEVICES", "3")
    assert L.ahip_init_devices(1) == 0, N.last_error()
    try:
        ranges = partition_members([offs[i + 1] - offs[i] for i in range(240)], 3)
        d_ins = [torch.from_numpy(comp[offs[lo]:offs[hi]].copy()).cuda() for lo, hi in ranges]
        caps = [(hi - lo) * 65536 for lo, hi in ranges]
        rc, d_outs, out_len, offsets, status = _decode_shards(L, N, [0, 0, 0], d_ins, caps)
        assert rc == 0 and status == [0, 0, 0], N.last_error()
        assert out_len == caps and offsets == [0, caps[0], caps[0] + caps[1], sum(caps)]
        assert L.ahip_debug_last_exchange() == 0
        got = np.concatenate([o[:n].cpu().numpy() for o, n in zip(d_outs, out_len)])
        assert np.array_equal(got, plain)
        # a damaged shard reports its own verdict and the others are unaffected
        bad = d_ins[1].clone()
        bad[40:48] = 0xff
        rc, d_outs, out_len, offsets, status = _decode_shards(L, N, [0, 0, 0], [d_ins[0], bad, d_ins[2]], caps)
        assert status[0] == 0 and status[2] == 0 and status[1] != 0 and rc == status[1]
        assert offsets[1] == caps[0] and offsets[3] == sum(out_len)
        # a device nobody selected
        rc = _decode_shards(L, N, [5], d_ins[:1], caps[:1])[0]
        assert rc == -4
    finally:
        monkeypatch.delenv("AHIP_FAKE_DEVICES")
        assert L.ahip_init_devices(1) == 0 and L.ahip_device_count() == 1
    whole = torch.from_numpy(comp).cuda()
    rc, d_outs, out_len, offsets, status = _decode_shards(L, N, [0], [whole], [len(plain)])
    assert rc == 0 and out_len == [len(plain)] and offsets == [0, len(plain)], N.last_error()
    assert np.array_equal(d_outs[0][:len(plain)].cpu().numpy(), plain)
    assert L.ahip_debug_last_exchange() == 1, "the size exchange did not go through RCCL"
    monkeypatch.setenv("AHIP_NO_RCCL", "1")
    rc, d_outs, out_len, offsets, status = _decode_shards(L, N, [0], [whole], [len(plain)])
    assert rc == 0 and offsets == [0, len(plain)] and L.ahip_debug_last_exchange() == 0


def test_device_resident_framed_encoders(native_built):
    """ahip_gzip_encode_device / ahip_zlib_encode_device: the framing bytes of the host-pointer encoders (the reference's,
    _gzip_encoder_web.dart:27-100 / _zlib_encoder_web.dart:27-73) around a DEFLATE stream that inflates to the input.
    (Host-pointer and device-resident forms of the same call give the same bytes: the encoder is deterministic.)"""
    import ctypes
    import gzip as _gz
    import numpy as np
    import struct
    import torch
    from archive_amd import _native as N
    L = N.lib()
    assert L.ahip_init(0) == 0
    data = (streams.text(300000, 5) + bytes(70000)) * 2
    src = np.frombuffer(data, dtype=np.uint8)
    d_in = torch.from_numpy(src.copy()).cuda()
    cap = L.ahip_deflate_bound(len(data)) + 32
    for level, wb in ((6, 15), (1, 12), (0, 15)):
        host = np.zeros(cap, dtype=np.uint8)
        n = ctypes.c_size_t()
        assert L.ahip_gzip_encode(src.ctypes.data, len(data), level, wb, 1234567, host.ctypes.data, cap, ctypes.byref(n)) == 0
        d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        m = ctypes.c_size_t()
        assert L.ahip_gzip_encode_device(d_in.data_ptr(), len(data), level, wb, 1234567, d_out.data_ptr(), cap, ctypes.byref(m), None) == 0, N.last_error()
        dev = bytes(d_out[:m.value].cpu().numpy())
        ref = bytes(host[:n.value])
        assert dev[:10] == ref[:10] == bytes([0x1f, 0x8b, 8, 0]) + struct.pack("<I", 1234567) + bytes([0, 0xff])
        assert dev[-8:] == ref[-8:] == struct.pack("<II", zlib.crc32(data), len(data))
        assert dev == ref  # byte for byte: same input, same parameters, same stream
        assert _gz.decompress(dev) == data and _gz.decompress(ref) == data
        assert L.ahip_zlib_encode(src.ctypes.data, len(data), level, wb, host.ctypes.data, cap, ctypes.byref(n)) == 0
        assert L.ahip_zlib_encode_device(d_in.data_ptr(), len(data), level, wb, d_out.data_ptr(), cap, ctypes.byref(m), None) == 0, N.last_error()
        dev = bytes(d_out[:m.value].cpu().numpy())
        ref = bytes(host[:n.value])
        assert dev[:2] == ref[:2] and dev[-4:] == ref[-4:] == struct.pack(">I", zlib.adler32(data))
        assert zlib.decompress(dev) == data and zlib.decompress(ref) == data
    # parameters the reference's Deflate refuses silently (deflate.dart:105-115): no DEFLATE bytes at all, the gzip trailer
    # carries the Deflate object's crc32 (0), but the zlib encoder took the Adler-32 of the input BEFORE it called
    # Deflate.stream (_zlib_encoder_web.dart:62-72): the real checksum, in the host and in the device-resident form
    for level, wb in ((12, 15), (6, 20), (6, 8)):
        host = np.zeros(cap, dtype=np.uint8)
        n, m = ctypes.c_size_t(), ctypes.c_size_t()
        d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        assert L.ahip_zlib_encode(src.ctypes.data, len(data), level, wb, host.ctypes.data, cap, ctypes.byref(n)) == 0
        assert L.ahip_zlib_encode_device(d_in.data_ptr(), len(data), level, wb, d_out.data_ptr(), cap, ctypes.byref(m), None) == 0, N.last_error()
        assert n.value == m.value == 6
        dev, ref = bytes(d_out[:6].cpu().numpy()), bytes(host[:6])
        assert dev == ref and ref[2:] == struct.pack(">I", zlib.adler32(data)) and (ref[0] * 256 + ref[1]) % 31 == 0
        assert L.ahip_gzip_encode(src.ctypes.data, len(data), level, wb, 7, host.ctypes.data, cap, ctypes.byref(n)) == 0
        assert L.ahip_gzip_encode_device(d_in.data_ptr(), len(data), level, wb, 7, d_out.data_ptr(), cap, ctypes.byref(m), None) == 0
        assert n.value == m.value == 18
        assert bytes(d_out[:18].cpu().numpy()) == bytes(host[:18]) and bytes(host[10:18]) == struct.pack("<II", 0, len(data))
    # too small a buffer reports the bound
    m = ctypes.c_size_t()
    tiny = torch.zeros(8, dtype=torch.uint8, device="cuda")
    assert L.ahip_gzip_encode_device(d_in.data_ptr(), len(data), 6, 15, 0, tiny.data_ptr(), 8, ctypes.byref(m), None) == -1 and m.value > len(data) // 100


def test_host_pointer_pipeline_on_one_device(native_built):
    """ahip_gzip_decode(host in, host out) on ONE device: a large stream of BGZF members is cut into slices that several
    contexts of the same device upload, decode and download side by side (PCIe overlapped with the decode); what cannot be
    cut that way takes the exact path.  Results equal the single-context decode."""
    import numpy as np
    import archive_amd
    from archive_amd import _native as N
    from tools import corpus
    L = N.lib()
    assert L.ahip_init(0) == 0
    assert L.ahip_init_devices(1) == 0 and L.ahip_device_count() == 1
    comp, plain = corpus.make_gzip(n_members=1400, want_plain=True)   # 37 MB -> 92 MB
    assert len(comp) >= (32 << 20)
    dec = archive_amd.GZipDecoder()
    out = dec.decode_bytes(bytes(comp))
    assert dec.last_status == 0 and L.ahip_debug_last_shards() == 4
    assert np.array_equal(np.frombuffer(out, dtype=np.uint8), plain)
    # a member without BC in the middle: not partitionable, exact path, same bytes
    mixed = bytes(comp) + streams.gz_member(streams.text(50000, 3))
    out2 = dec.decode_bytes(mixed)
    assert L.ahip_debug_last_shards() == 1 and out2 == bytes(plain) + streams.text(50000, 3)


def test_sharded_deflate_and_bzip2(native_built, monkeypatch):
    """ahip_deflate_shards / ahip_bzip2_decode_shards on three contexts of the one device there is (AHIP_FAKE_DEVICES):
    the shards' outputs laid end to end at the exchanged offsets are one DEFLATE stream of the whole input (zlib and the
    library's own Inflate agree), the per-shard CRCs combine to the input's; the bzip2 blocks of one stream decoded by
    three contexts give the bytes and the verdict of the unsharded call -- also when a block is damaged."""
    import bz2
    import ctypes
    import numpy as np
    import torch
    import archive_amd
    from archive_amd import _native as N
    from archive_amd.sharding import crc32_combine, partition_bytes
    L = N.lib()
    assert L.ahip_init(0) == 0
    monkeypatch.setenv("AHIP_FAKE_DEVICES", "3")
    assert L.ahip_init_devices(1) == 0, N.last_error()
    try:
        data = streams.text(900000, 3) + bytes(150000) + streams.text(333333, 4)
        src = np.frombuffer(data, dtype=np.uint8)
        for level, cuts in ((6, partition_bytes(len(data), 3)), (1, [(0, 100001), (100001, 100001), (100001, len(data))]), (0, partition_bytes(len(data), 3))):
            n = len(cuts)
            d_ins = [torch.from_numpy(src[lo:hi].copy()).cuda() if hi > lo else torch.zeros(1, dtype=torch.uint8, device="cuda") for lo, hi in cuts]
            lens = [hi - lo for lo, hi in cuts]
            caps = [L.ahip_deflate_bound(v) + 64 for v in lens]
            d_outs = [torch.zeros(c, dtype=torch.uint8, device="cuda") for c in caps]
            arr_dev = (ctypes.c_int32 * n)(*([0] * n))
            arr_in = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_ins])
            arr_len = (ctypes.c_size_t * n)(*lens)
            arr_out = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_outs])
            arr_cap = (ctypes.c_size_t * n)(*caps)
            out_len = (ctypes.c_size_t * n)()
            offsets = (ctypes.c_uint64 * (n + 1))()
            crcs = (ctypes.c_uint32 * n)()
            rc = L.ahip_deflate_shards(n, arr_dev, arr_in, arr_len, level, 15, arr_out, arr_cap, out_len, offsets, crcs)
            assert rc == 0, N.last_error()
            whole = bytearray(offsets[n])
            crc = 0
            for s in range(n):
                assert offsets[s + 1] - offsets[s] == out_len[s]
                whole[offsets[s]:offsets[s] + out_len[s]] = bytes(d_outs[s][:out_len[s]].cpu().numpy())
                crc = crc32_combine(crc, crcs[s], lens[s])
            assert zlib.decompress(bytes(whole), -15) == data, level
            assert archive_amd.Inflate(bytes(whole)).get_bytes() == data, level
            assert crc == zlib.crc32(data)
        # ---- bzip2 ----
        text = streams.text(1400000, 8) + bytes(range(256)) * 400
        stream = bz2.compress(text, 1)                       # 100 k blocks: about fifteen of them
        damaged = bytearray(stream)
        damaged[len(damaged) * 2 // 3] ^= 0x10               # inside some block of the last third
        for name, comp, verify in (("clean", stream, 1), ("damaged", bytes(damaged), 1), ("damaged, unverified", bytes(damaged), 0),
                                   ("truncated", stream[:len(stream) * 3 // 4], 1)):
            cap = len(text) + 65536
            one = torch.from_numpy(np.frombuffer(comp, dtype=np.uint8).copy()).cuda()
            ref_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            ref_n = ctypes.c_size_t()
            ref_rc = L.ahip_bzip2_decode_device(one.data_ptr(), len(comp), verify, ref_out.data_ptr(), cap, ctypes.byref(ref_n), None)
            n = 3
            copies = [one.clone() for _ in range(n)]
            d_outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(n)]
            arr_dev = (ctypes.c_int32 * n)(*([0] * n))
            arr_in = (ctypes.c_void_p * n)(*[t.data_ptr() for t in copies])
            arr_out = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_outs])
            arr_cap = (ctypes.c_size_t * n)(*([cap] * n))
            out_len = (ctypes.c_size_t * n)()
            offsets = (ctypes.c_uint64 * (n + 1))()
            status = (ctypes.c_int32 * n)()
            rc = L.ahip_bzip2_decode_shards(n, arr_dev, arr_in, len(comp), verify, arr_out, arr_cap, out_len, offsets, status)
            assert rc == ref_rc, (name, rc, ref_rc, N.last_error())
            got = b"".join(bytes(d_outs[s][:out_len[s]].cpu().numpy()) for s in range(n))
            assert offsets[n] == len(got)
            if ref_rc in (0, 1):
                assert got == bytes(ref_out[:ref_n.value].cpu().numpy()), name
            if name == "clean":
                assert got == text and min(out_len) > 0
        # ---- a false block magic on a shard boundary ----
        # The magic scan lists EVERY bit position that holds the 48 bits, also inside a block's data, and shard s starts
        # blindly at candidate K s / n.  A magic planted in the second block's data (at a spot where that block still decodes
        # and still ends where it did: found with the oracle's block function) becomes candidate 2 of 7 -- the first one of
        # shard 1 of 3.  decodeStream never looks there (the chain steps from block to block); neither must the shards: the
        # merge sees that shard 0's chain ends at candidate 3 and runs shard 1 again from there.
        from oracle import pyoracle as orc
        text2 = streams.text(450000, 8) + bytes(range(256)) * 100
        clean = bz2.compress(text2, 1)
        starts = orc.bzip2_block_bits(clean)
        assert len(starts) == 6                               # five blocks + the end-of-stream marker
        big, nb = int.from_bytes(clean, "big"), len(clean) * 8
        planted = None
        for bit in range(starts[1] + 2000, starts[2] - 100, 7):
            sh = nb - bit - 48
            cand = ((big & ~(((1 << 48) - 1) << sh)) | (0x314159265359 << sh)).to_bytes(len(clean), "big")
            r = orc.bzip2_block(cand, starts[1], 1)
            if r["status"] == 0 and r["end_bit"] == starts[2]:
                planted = cand
                break
        assert planted is not None
        for verify in (0, 1):
            cap = len(text2) + 200000
            one = torch.from_numpy(np.frombuffer(planted, dtype=np.uint8).copy()).cuda()
            ref_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            ref_n = ctypes.c_size_t()
            ref_rc = L.ahip_bzip2_decode_device(one.data_ptr(), len(planted), verify, ref_out.data_ptr(), cap, ctypes.byref(ref_n), None)
            st, want = orc.bzip2_decode(planted, verify=bool(verify), cap=cap)
            assert ref_rc == st and bytes(ref_out[:ref_n.value].cpu().numpy()) == want
            n = 3
            copies = [one.clone() for _ in range(n)]
            d_outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(n)]
            arr_dev = (ctypes.c_int32 * n)(*([0] * n))
            arr_in = (ctypes.c_void_p * n)(*[t.data_ptr() for t in copies])
            arr_out = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_outs])
            arr_cap = (ctypes.c_size_t * n)(*([cap] * n))
            out_len = (ctypes.c_size_t * n)()
            offsets = (ctypes.c_uint64 * (n + 1))()
            status = (ctypes.c_int32 * n)()
            before = L.ahip_debug_bz_reruns()
            rc = L.ahip_bzip2_decode_shards(n, arr_dev, arr_in, len(planted), verify, arr_out, arr_cap, out_len, offsets, status)
            assert rc == st, (verify, rc, st, N.last_error())
            # verify = 0: shard 1 began at the planted magic and was run again; verify = 1: the block the magic was planted
            # in fails its CRC, the stream ends in shard 0 and nothing behind it counts
            assert L.ahip_debug_bz_reruns() == before + (0 if verify else 1)
            got = b"".join(bytes(d_outs[s][:out_len[s]].cpu().numpy()) for s in range(n))
            assert got == want and offsets[n] == len(got), verify
    finally:
        monkeypatch.delenv("AHIP_FAKE_DEVICES")
        assert L.ahip_init_devices(1) == 0 and L.ahip_device_count() == 1


def test_shards_call_and_host_decode_from_two_threads(native_built, monkeypatch):
    """A *_shards call lets go of the library's lock while its shards run on the device workers (other threads may use the
    library meanwhile) -- and ahip_gzip_decode, which fans a BGZF stream out over the SAME workers, must not hand them a
    second job then: it takes the one-context path while they are busy.  Two threads, many rounds, every result checked."""
    import ctypes
    import threading
    import numpy as np
    import torch
    import archive_amd
    from archive_amd import _native as N
    from tools import corpus
    L = N.lib()
    assert L.ahip_init(0) == 0
    monkeypatch.setenv("AHIP_FAKE_DEVICES", "3")
    assert L.ahip_init_devices(1) == 0, N.last_error()
    try:
        comp, plain = corpus.make_gzip(n_members=192, want_plain=True)      # 12 MiB out: large enough to be fanned out
        comp, plain = bytes(comp), bytes(plain)
        # the shards call's input: the same members cut in three
        sizes = []
        pos = 0
        while pos < len(comp):
            bs = comp[pos + 16] | (comp[pos + 17] << 8)
            sizes.append(bs + 1)
            pos += bs + 1
        from archive_amd.sharding import partition_members
        parts = partition_members(sizes, 3)
        offs = [0]
        for v in sizes:
            offs.append(offs[-1] + v)
        n = 3
        src = np.frombuffer(comp, dtype=np.uint8)
        d_ins = [torch.from_numpy(src[offs[lo]:offs[hi]].copy()).cuda() for lo, hi in parts]
        lens = [offs[hi] - offs[lo] for lo, hi in parts]
        cap = len(plain)
        errors_seen = []

        def shards_loop():
            try:
                for _ in range(12):
                    d_outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(n)]
                    arr_dev = (ctypes.c_int32 * n)(*([0] * n))
                    arr_in = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_ins])
                    arr_len = (ctypes.c_size_t * n)(*lens)
                    arr_out = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_outs])
                    arr_cap = (ctypes.c_size_t * n)(*([cap] * n))
                    out_len = (ctypes.c_size_t * n)()
                    offsets = (ctypes.c_uint64 * (n + 1))()
                    status = (ctypes.c_int32 * n)()
                    rc = L.ahip_gzip_decode_shards(n, arr_dev, arr_in, arr_len, arr_out, arr_cap, out_len, offsets, status)
                    got = b"".join(bytes(d_outs[s][:out_len[s]].cpu().numpy()) for s in range(n))
                    if rc != 0 or got != plain:
                        errors_seen.append(("shards", rc, len(got)))
            except Exception as e:  # noqa: BLE001
                errors_seen.append(("shards", repr(e)))

        def host_loop():
            try:
                for _ in range(12):
                    out = archive_amd.GZipDecoder().decode_bytes(comp)
                    if out != plain:
                        errors_seen.append(("host", len(out)))
            except Exception as e:  # noqa: BLE001
                errors_seen.append(("host", repr(e)))

        ts = [threading.Thread(target=shards_loop), threading.Thread(target=host_loop)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in ts), "deadlock"
        assert not errors_seen, errors_seen[:3]
    finally:
        monkeypatch.delenv("AHIP_FAKE_DEVICES")
        assert L.ahip_init_devices(1) == 0 and L.ahip_device_count() == 1


def _stream_shards(L, N, n, d_in, in_len, data_off, caps):
    import ctypes
    import torch
    d_outs = [torch.full((c + 64,), 0xA5, dtype=torch.uint8, device="cuda") for c in caps]
    arr_dev = (ctypes.c_int32 * n)(*([0] * n))
    arr_in = (ctypes.c_void_p * n)(*([d_in.data_ptr()] * n))   # (one device here: every "device" sees the same copy)
    arr_out = (ctypes.c_void_p * n)(*[t.data_ptr() for t in d_outs])
    arr_cap = (ctypes.c_size_t * n)(*caps)
    out_len = (ctypes.c_size_t * n)()
    offsets = (ctypes.c_uint64 * (n + 1))()
    end_pos, handled = ctypes.c_uint64(), ctypes.c_int32()
    rc = L.ahip_inflate_stream_shards(n, arr_dev, arr_in, in_len, data_off, arr_out, arr_cap, out_len, offsets, ctypes.byref(end_pos), ctypes.byref(handled))
    return rc, d_outs, list(out_len), list(offsets), end_pos.value, handled.value


def test_one_member_over_the_contexts_of_one_process(native_built, monkeypatch):
    """ahip_inflate_stream_shards: ONE long DEFLATE stream decoded by the device contexts of one process -- three contexts of the
    one device there is (AHIP_FAKE_DEVICES, worker threads) and, without ahip_init_devices, several shards on the calling
    thread.  The slices at their offsets are the input; a slice that does not fit reports the sizes that are needed; a short
    stream is not taken."""
    import gzip
    import torch
    from archive_amd import _native as N
    from tools import corpus
    L = N.lib()
    assert L.ahip_init(0) == 0
    data = bytes(corpus.text(corpus.LOG, 55, 0, 11 << 20))
    g = gzip.compress(data, 6, mtime=0)
    d_in = torch.frombuffer(bytearray(g), dtype=torch.uint8).cuda()

    def check(n):
        rc, d_outs, out_len, offsets, end_pos, handled = _stream_shards(L, N, n, d_in, len(g), 10, [len(data)] * n)
        assert rc == 0 and handled == 1, (rc, handled, N.last_error())
        assert offsets[n] == len(data) and end_pos == len(g) - 8
        whole = bytearray(len(data))
        for s in range(n):
            assert offsets[s] + out_len[s] == offsets[s + 1]
            assert bool((d_outs[s][out_len[s]:] == 0xA5).all())
            whole[offsets[s]:offsets[s] + out_len[s]] = bytes(d_outs[s][:out_len[s]].cpu().numpy())
        assert bytes(whole) == data
        assert sum(1 for v in out_len if v) == n
        return out_len
    check(2)                      # no worker contexts: the shards run one after another on the calling thread
    monkeypatch.setenv("AHIP_FAKE_DEVICES", "3")
    assert L.ahip_init_devices(1) == 0, N.last_error()
    try:
        sizes = check(3)
        assert L.ahip_debug_last_shards() == 3
        check(5)                  # more shards than contexts: dealt out round robin
        rc, _, need, _, _, handled = _stream_shards(L, N, 3, d_in, len(g), 10, [sizes[0], sizes[1] - 1, sizes[2]])
        assert rc == N.AHIP_E_CAP and handled == 0 and list(need) == list(sizes), (rc, need, sizes)
        short = gzip.compress(data[:200000], 6, mtime=0)
        d_short = torch.frombuffer(bytearray(short), dtype=torch.uint8).cuda()
        rc, _, out_len, offsets, _, handled = _stream_shards(L, N, 3, d_short, len(short), 10, [200000] * 3)
        assert rc == 0 and handled == 0 and sum(out_len) == 0
    finally:
        monkeypatch.delenv("AHIP_FAKE_DEVICES")
        assert L.ahip_init_devices(1) == 0 and L.ahip_device_count() == 1
