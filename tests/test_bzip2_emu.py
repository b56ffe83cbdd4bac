"""Phase 1 of the bzip2 block decoder (header, tables, the windowed Huffman + MTF symbol loop of bz_decode_block_wave) run
on the CPU: 64 threads as the 64 lanes of a wave (tests/emu/wave_emu.hpp, tests/emu/bzip2_emu.cc).  The emulation
finishes each block on the host the plain way and the bytes are compared with what Python's bz2 compressed -- parity
for the device code without a GPU."""
import bz2
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")


@pytest.fixture(scope="module")
def emu():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "bzip2_emu")
    src = os.path.join(ROOT, "tests", "emu", "bzip2_emu.cc")
    deps = [src, os.path.join(ROOT, "tests", "emu", "wave_emu.hpp")] + [os.path.join(ROOT, "archive_amd", "csrc", f)
                                                                          for f in ("bzip2_kernels.hpp", "common.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-o", exe, src])
    return exe


def _cases():
    rnd = random.Random(5)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"the", b"quick", b"brown", b"fox", b"lorem", b"ipsum", b"0123", b"\n"]

    def text(n):
        out = bytearray()
        while len(out) < n:
            out += rnd.choice(words) + b" "
        return bytes(out[:n])

    return {
        "text": (text(20000), 9),
        "two_blocks_level1": (text(130000), 1),
        "zeros": (bytes(30000), 9),                                   # one long RUNA/RUNB number
        "random_bytes": (bytes(rnd.getrandbits(8) for _ in range(4000)), 9),   # 256 symbols in use: MTF indices >= 64
        "byte_runs": (b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 300) for _ in range(300)), 9),
        "every_byte_then_text": (bytes(range(256)) * 3 + text(60000), 9),  # rare symbols: codes longer than the 10-bit table
        "one_byte": (b"a", 9),
        "empty": (b"", 9),
    }


@pytest.mark.parametrize("name", list(_cases().keys()))
def test_symbol_loop_on_the_wave_emulation(emu, tmp_path, name):
    data, level = _cases()[name]
    comp = bz2.compress(data, level)
    a, b = tmp_path / "c.bz2", tmp_path / "plain.bin"
    a.write_bytes(comp)
    b.write_bytes(data)
    r = subprocess.run([emu, str(a), str(b)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bzip2 emu ok" in r.stdout and "%d bytes" % len(data) in r.stdout, r.stdout + r.stderr


def test_damaged_streams_get_the_same_verdict_from_both_forms_of_the_huffman_pass(emu, tmp_path):
    """Bit flips and truncations: the position-parallel pass (jump tiles, walk, groups) and the serial wave must stop at
    the same symbol with the same status -- or both decode the (then wrong) data alike."""
    rnd = random.Random(9)
    data, level = _cases()["text"]
    comp = bz2.compress(data, level)
    plain = tmp_path / "plain.bin"
    plain.write_bytes(data)
    seen = set()
    for t in range(36):
        b = bytearray(comp)
        if t % 3:
            pos = rnd.randrange(10 * 8, len(b) * 8 - 80)
            b[pos >> 3] ^= 1 << (pos & 7)
        else:
            b = b[:rnd.randrange(20, len(b) - 1)]
        p = tmp_path / "d.bz2"
        p.write_bytes(bytes(b))
        r = subprocess.run([emu, str(p), str(plain), "agree"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "bzip2 emu agree" in r.stdout, (t, r.stdout + r.stderr)
        seen.add(r.stdout.split(":")[1].strip().split(" ")[0] if ":" in r.stdout else "")
    assert len(seen) >= 2  # (errors and clean decodes both occurred)
