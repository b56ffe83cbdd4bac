"""The block finder of the single-long-member path (sm_find_wave: bit-parallel BTYPE / HLIT / HDIST filter over 2 048
positions a step, Kraft filter, header filter) on the CPU wave emulation, against a plain scan of the same ranges
(tests/emu/sm_find_emu.cc)."""
import os
import random
import subprocess
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import corpus  # noqa: E402

BUILD = os.path.join(ROOT, "tests", "emu", "_build")


@pytest.fixture(scope="module")
def emu():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "sm_find_emu")
    src = os.path.join(ROOT, "tests", "emu", "sm_find_emu.cc")
    deps = [src, os.path.join(ROOT, "tests", "emu", "wave_emu.hpp")] + [os.path.join(ROOT, "archive_amd", "csrc", f)
                                                                          for f in ("sm_inflate.hpp", "inflate_par.hpp", "inflate_wave.hpp", "common.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-o", exe, src])
    return exe


@pytest.mark.parametrize("kind,level", [(corpus.WIKI, 6), (corpus.LOG, 9), ("incompressible", 6)])
def test_finder_matches_a_plain_scan(emu, tmp_path, kind, level):
    if kind == "incompressible":  # stored blocks back to back, text in between: the stored-block starts of the finder
        rnd0 = random.Random(3)
        data = b"".join(rnd0.randbytes(300000) + bytes(corpus.text(corpus.LOG, 8, i, 100000)) for i in range(5))
    else:
        data = bytes(corpus.text(kind, 8, 0, 2 << 20))
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    p = tmp_path / "s.deflate"
    p.write_bytes(raw)
    nbits = len(raw) * 8
    rnd = random.Random(11)
    ranges = []
    for _ in range(16):
        q0 = rnd.randrange(0, nbits - 100)
        ranges += [q0, min(q0 + rnd.choice([40, 300, 5000, 70000, 200000]), nbits)]
    ranges += [nbits - 5000, nbits, 0, 64, 31, 33, 1000, 1000 + 2048, 4096, 4096 + 2048 * 9 + 5]
    r = subprocess.run([emu, str(p)] + [str(x) for x in ranges], capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0 and "sm find emu ok: %d ranges" % (len(ranges) // 2) in r.stdout, r.stdout + r.stderr
