"""Parity of the DEVICE code with the oracle, without a GPU: raw DEFLATE streams -- the valid and malformed shapes of
tests/streams.py, the reference's quirks (over-subscribed code lengths, a back-reference in front of the output), and a
sample of the damaged streams of tests/test_fuzz_gpu.py -- go through the tokenizer / late / resolver code of
archive_amd/csrc on a CPU emulation of one wave (tests/emu/raw_emu.cc) and must give the oracle's status, bytes and stream
position.  (The same comparisons run on the MI355X through the C-ABI in tests/test_inflate_gpu.py and test_fuzz_gpu.py; here it
is the device code's logic that is checked, on every round's CPU test run.)"""
import os
import struct
import subprocess

import pytest

from tests import streams
from tests.test_fuzz_gpu import _mutants

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def _binary():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "raw_emu")
    src = os.path.join(ROOT, "tests", "emu", "raw_emu.cc")
    deps = [src, os.path.join(ROOT, "tests", "emu", "wave_emu.hpp")] + [os.path.join(ROOT, "archive_amd", "csrc", f)
                                                                        for f in ("common.hpp", "inflate_wave.hpp", "inflate_par.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-o", exe, src])
    return exe


def _run(tmp_path, cases):
    with open(tmp_path / "cases.bin", "wb") as f:
        f.write(struct.pack("<I", len(cases)))
        for raw in cases:
            f.write(struct.pack("<I", len(raw)) + raw)
    r = subprocess.run([_binary(), str(tmp_path / "cases.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    outs, off, res = (tmp_path / "out.bin").read_bytes(), 0, []
    for line in r.stdout.strip().split("\n"):
        _, st, olen, pos, _, route = line.split()
        n = struct.unpack_from("<Q", outs, off)[0]
        off += 8
        res.append((int(st), outs[off:off + n], int(pos), route))
        off += n
    assert len(res) == len(cases)
    return res


def _compare(orc, cases, results, names=None):
    bad = []
    for i, (raw, (st, out, pos, route)) in enumerate(zip(cases, results)):
        ost, oout, opos = orc.inflate_raw(raw, cap=1 << 20)
        ok = st == ost and (ost not in (0, 1) or (out == oout and pos == opos))
        if not ok:
            bad.append((names[i] if names else i, route, (st, len(out), pos), (ost, len(oout) if oout is not None else None, opos)))
    assert not bad, bad[:10]


def test_stream_shapes_and_quirks(tmp_path, orc):
    named = list(streams.valid_raw_streams()) + list(streams.malformed_raw_streams())
    named += [("valid+4:" + n, r + bytes(4)) for n, r in streams.valid_raw_streams()]
    named += [("oversubscribed", streams.oversubscribed_dynamic_block()), ("oversubscribed+data", streams.oversubscribed_dynamic_block((1, 0, 1, 1, 0, 0))),
              ("far reference", streams.raw_far_reference()), ("far reference, longer", streams.raw_far_reference(b"q", 9, 6, 1))]
    cases = [r for _, r in named]
    res = _run(tmp_path, cases)
    _compare(orc, cases, res, [n for n, _ in named])
    routes = {route for _, _, _, route in res}
    assert "late:exact" in routes, routes  # the over-subscribed tables went through the reference's own table


def test_a_sample_of_the_damaged_streams(tmp_path, orc):
    cases = _mutants(1, 360)[::3] + _mutants(4, 360)[1::6]   # 180 of the 2 160 the GPU test decodes
    res = _run(tmp_path, cases)
    _compare(orc, cases, res)
