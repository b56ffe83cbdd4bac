"""The WHOLE device path of a member on the CPU: the flow tokenizer (inflate_member<false, true>: block headers, table build
with second-level tables, huffman_block_tokenize, the serial token emitter) and the resolver, executed by 64 host threads as
the 64 lanes of one wave (tests/emu/wave_emu.hpp, tests/emu/inflate_emu.cc), glued like the two kernels of archive_hip.hip.
The decoded bytes must equal what zlib's compressor was given.  Runs without a GPU: it is the device code's own logic that
is checked here, not the hardware."""
import os
import random
import subprocess

import pytest

from tests import streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")


def _binary(variant):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "inflate_emu_" + variant)
    src = os.path.join(ROOT, "tests", "emu", "inflate_emu.cc")
    deps = [src, os.path.join(ROOT, "tests", "emu", "wave_emu.hpp")] + [os.path.join(ROOT, "archive_amd", "csrc", f)
                                                                        for f in ("common.hpp", "inflate_wave.hpp", "inflate_par.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        # ("all_lanes": the rounds of the resolver read for every lane, as before round 5 -- the masked reads must change nothing)
        cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-o", exe, src] + (["-DAHIP_ROUNDS_ALL_LANES"] if variant == "all_lanes" else [])
        subprocess.check_call(cmd)
    return exe


def _parts():
    from tools import corpus
    rnd = random.Random(7)
    noise = bytes(rnd.getrandbits(8) for _ in range(9000))
    recs = [bytes(rnd.getrandbits(8) for _ in range(25)) for _ in range(160)]
    return [
        (streams.text(30000, 1), 6), (bytes(70000), 6), (b"abc" * 5000 + b"0123456789" * 900, 6), (noise, 6), (noise[:5000], 0),
        (streams.text(3000, 2) + bytes(range(256)) * 20 + streams.text(3000, 2), 9), (b"", 6), (b"x", 6), (b"xy" * 2, 1),
        (streams.text(2000, 3) * 12, 6), ((streams.text(5000, 4) + noise[:3000]) * 6, 6), (streams.text(66000, 5), 1),
        (bytes(corpus.text(corpus.LOG, 1234, 0, 65536)), 6),   # one member of the benchmark stream (config 4)
        (bytes(corpus.text(corpus.WIKI, 8, 0, 65536)), 6),     # one member of config 2b
        (bytes(corpus.text(corpus.LOG, 1234, 5, 200000)), 9),  # several blocks, level 9
        (b"".join(r + bytes([rnd.getrandbits(8)]) for _ in range(6) for r in recs), 6),  # 25-byte matches from flushed output
    ]


@pytest.mark.parametrize("variant", ["production", "all_lanes"])
def test_device_path_on_the_cpu(tmp_path, variant):
    exe = _binary(variant)
    parts = _parts()
    (tmp_path / "m.gz").write_bytes(b"".join(streams.gz_member(p, level=lv) for p, lv in parts))
    (tmp_path / "m.bin").write_bytes(b"".join(p for p, _ in parts))
    (tmp_path / "m.sz").write_text(" ".join(str(len(p)) for p, _ in parts))
    r = subprocess.run([exe, str(tmp_path / "m.gz"), str(tmp_path / "m.bin"), str(tmp_path / "m.sz")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "inflate emu ok: %d members, %d bytes" % (len(parts), sum(len(p) for p, _ in parts)) in r.stdout, r.stdout


def test_back_references_into_earlier_members_on_the_cpu(tmp_path):
    """Quirk q8 (tests/test_inflate_gpu.py::test_back_reference_into_previous_gzip_member) through the device code: the
    tokenizer flags the member, it is resolved once its predecessors' bytes exist."""
    from oracle import pyoracle as orc
    first = streams.text(40000, 77)
    far = streams.gz_wrap(streams.raw_far_reference())
    far2 = streams.gz_wrap(streams.raw_far_reference(lit=b"xy", length=5, dist_extra=0))
    exe = _binary("production")
    for i, g in enumerate([streams.gz_member(first) + far,
                           streams.gz_member(first) + far + far2 + streams.gz_member(streams.text(3000, 78)),
                           streams.gz_member(b"12345") + far]):
        st, want = orc.gzip_decode(g)
        assert st == 0
        (tmp_path / "q.gz").write_bytes(g)
        (tmp_path / "q.bin").write_bytes(want)
        (tmp_path / "q.sz").write_text("auto")
        r = subprocess.run([exe, str(tmp_path / "q.gz"), str(tmp_path / "q.bin"), str(tmp_path / "q.sz")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "inflate emu ok" in r.stdout, (i, r.stdout + r.stderr)
        assert "members reaching into earlier ones 0" not in r.stdout, r.stdout


def test_sizing_run_that_runs_out_of_token_room_on_the_cpu(tmp_path):
    """A sizing run that keeps its tokens in an area cut short (on the device: by a false `1f 8b 08` right behind the
    member's start) gives the tokens up and goes on counting with the flow decoder; it used to hand the rest of the
    member to the symbol-by-symbol emitter, 26 x slower, and 36 such members set the kernel's tail on config 4 without BC."""
    exe = _binary("production")
    parts = [p for p in _parts() if len(p[0]) > 1000]
    (tmp_path / "m.gz").write_bytes(b"".join(streams.gz_member(p, level=lv) for p, lv in parts))
    (tmp_path / "m.bin").write_bytes(b"".join(p for p, _ in parts))
    (tmp_path / "m.sz").write_text(" ".join(str(len(p)) for p, _ in parts))
    env = dict(os.environ, EMU_SIZING_SMALL="1")
    r = subprocess.run([exe, str(tmp_path / "m.gz"), str(tmp_path / "m.bin"), str(tmp_path / "m.sz")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "inflate emu sizing ok: %d members" % len(parts) in r.stdout, r.stdout + r.stderr
    gave_up = int(r.stdout.split("bytes;")[1].split()[0])
    assert gave_up >= len(parts) - 2, r.stdout  # (nearly) every member is larger than the area
    assert "hand-overs to the serial emitter 0" in r.stdout, r.stdout  # ... and none of them fell back to the slow path
