"""The resolver's DEVICE code (archive_amd/csrc/inflate_par.hpp: resolve_member and everything under it) run on the CPU:
64 host threads are the 64 lanes of one wave, the cross-lane primitives of common.hpp are exchanges between barriers
(tests/emu/wave_emu.hpp).  Token streams are made from real DEFLATE data by a plain tokenizer inside
tests/emu/resolver_emu.cc, cut into runs of random length like the tokenizer's directory (stored blocks as directory
entries of their own, some runs flagged DF_BIG), and the result is compared byte for byte with a sequential LZ77 replay.
Builds: the production window geometry, a small window (chunks split at the window's end all the time), a large one, and a
large one with a short list of deferred matches (chunks split where the list is full); and the 16-bit symbol form of the
chunked single-stream decode: the first part of every member lies "in front of the chunk", what refers to it must come out
as markers that point at the right bytes."""
import os
import random
import subprocess
import zlib

import pytest

from tests import streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")


def _binary(variant):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "resolver_emu_" + variant.replace("+", "_"))
    src = os.path.join(ROOT, "tests", "emu", "resolver_emu.cc")
    deps = [src, os.path.join(ROOT, "tests", "emu", "wave_emu.hpp")] + [os.path.join(ROOT, "archive_amd", "csrc", f)
                                                                        for f in ("common.hpp", "inflate_wave.hpp", "inflate_par.hpp", "inflate_res_wg.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        flags = {"production": [], "small_window": ["-DAHIP_WIN_CAP=1024", "-DAHIP_WIN_KEEP=400"],
                 "big_window": ["-DAHIP_WIN_CAP=8192", "-DAHIP_WIN_KEEP=2048"],
                 "small_pending": ["-DAHIP_WIN_CAP=8192", "-DAHIP_WIN_KEEP=512", "-DAHIP_PEND_CAP=128"],
                 "symbols": ["-DEMU_SYM"], "symbols_small_window": ["-DEMU_SYM", "-DAHIP_WIN_CAP=1024", "-DAHIP_WIN_KEEP=400"],
                 # the workgroup-per-member resolver (inflate_res_wg.hpp): production geometry (4 waves, 36 KiB ring); one wave
                 # (no chunk ever waits: logic without races); three waves on a ring with 1 KiB of room (chunks wait for room
                 # all the time, the ring wraps at another place); eight waves
                 "wg": ["-DEMU_WG"], "wg_one_wave": ["-DEMU_WG", "-DAHIP_WG_WAVES=1"],
                 "wg_tight_ring": ["-DEMU_WG", "-DAHIP_WG_WAVES=3", "-DAHIP_WG_RING=33856"],
                 "wg_eight_waves": ["-DEMU_WG", "-DAHIP_WG_WAVES=8", "-DAHIP_WG_RING=40960"]}[variant]
        cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-o", exe, src] + flags
        subprocess.check_call(cmd)
    return exe


def _records(rnd):
    recs = [bytes(rnd.getrandbits(8) for _ in range(25)) for _ in range(160)]
    return b"".join(r + bytes([rnd.getrandbits(8)]) for _ in range(6) for r in recs)


def _corpus():
    rnd = random.Random(7)
    noise = bytes(rnd.getrandbits(8) for _ in range(9000))
    parts = [
        (streams.text(30000, 1), 6),                        # text: short matches, mostly far
        (bytes(70000), 6),                                  # zeros: 258-byte matches at distance 1 (self-overlapping)
        (b"abc" * 5000 + b"0123456789" * 900, 6),           # periods 3 and 10
        (noise, 6),                                         # incompressible: literals (zlib may store it)
        (noise[:5000], 0),                                  # stored blocks: one 3-word record
        (streams.text(3000, 2) + bytes(range(256)) * 20 + streams.text(3000, 2), 9),
        (b"", 6), (b"x", 6), (b"xy" * 2, 1),                # empty and tiny members
        (streams.text(2000, 3) * 12, 6),                    # long matches (> 16 bytes) at 2 000-byte distance: inside the window
        ((streams.text(5000, 4) + noise[:3000]) * 6, 6),    # long matches at 8 000 bytes: flushed output
        (streams.text(66000, 5), 1),                        # level 1: shorter matches, more literals
        (_records(rnd), 6),                                 # 25-byte matches at ~4 KB distance: flushed output, 17..32 bytes
    ]
    blob = b"".join(streams.gz_member(p, level=lv) for p, lv in parts)
    # the plain tokenizer of the test rejects references across members; every member stands alone
    assert all(zlib.crc32(p) is not None for p, _ in parts)
    return blob, sum(len(p) for p, _ in parts), len(parts)


@pytest.mark.parametrize("variant", ["production", "small_window", "big_window", "small_pending", "symbols", "symbols_small_window"])
def test_resolver_device_code_on_the_cpu(tmp_path, variant):
    exe = _binary(variant)
    blob, total, members = _corpus()
    path = tmp_path / "members.gz"
    path.write_bytes(blob)
    for seed in (1, 2):  # different cuts of the token stream into runs
        r = subprocess.run([exe, str(path), str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "resolver emu ok" in r.stdout and "%d members, %d bytes" % (members, total) in r.stdout, r.stdout
        if variant.startswith("symbols"):
            assert int(r.stdout.split("markers checked:")[1].split()[0]) > 1000, r.stdout


@pytest.mark.parametrize("variant", ["wg", "wg_one_wave", "wg_tight_ring", "wg_eight_waves"])
def test_workgroup_resolver_on_the_cpu(tmp_path, variant):
    """The workgroup-per-member resolver (archive_amd/csrc/inflate_res_wg.hpp: several waves, one LDS ring that holds
    DEFLATE's whole reach, chunks completing in stream order through a frontier word in LDS) with every wave as 64 host
    threads.  The same corpus and run cuts as above, the output placed at every alignment in turn (the ring goes out in
    16-byte units aligned in global memory; nothing in front of the member or behind it may be touched), plus members of the
    benchmark's kind: 64 KiB of log text wrap the 36 KiB ring."""
    exe = _binary(variant)
    blob, total, members = _corpus()
    log = b"".join(streams.gz_member(streams.text(65536, 40 + i), level=6) for i in range(3))
    path, path2 = tmp_path / "members.gz", tmp_path / "log.gz"
    path.write_bytes(blob)
    path2.write_bytes(log)
    for seed in (1, 2):
        r = subprocess.run([exe, str(path), str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "resolver emu ok [workgroup-per-member resolver]" in r.stdout and "%d members, %d bytes" % (members, total) in r.stdout, r.stdout
    r = subprocess.run([exe, str(path2), "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "3 members, 196608 bytes" in r.stdout, r.stdout + r.stderr
