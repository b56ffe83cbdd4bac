"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer: `make -C oracle sanitize` builds
oracle/_build/oracle_san from the oracle's own sources; it decodes the reference's golden vectors, hand-made malformed
streams and a few hundred damaged streams with exactly-sized buffers.  Any out-of-bounds access or undefined
arithmetic aborts the run.  CPU only."""
import os
import random
import subprocess
import zlib

from tests import streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"])
    return os.path.join(ROOT, "oracle", "_build", "oracle_san")


def _run(exe, kind, paths):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, kind] + [str(p) for p in paths], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (kind, r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    return r.stdout


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = _build()
    g = os.path.join(ROOT, "tests", "golden")
    _run(exe, "gzip", [os.path.join(g, f) for f in ("a_txt_gz.in", "cat_jpg_gz.in", "test2_tar_gz.in", "gzip_multi_member.in")])
    _run(exe, "raw", [os.path.join(g, f) for f in ("inflate_data_bin.in", "zip_entry_test_txt.in")])
    _run(exe, "zlib", [os.path.join(g, "git_zlib_first_member.in")])
    _run(exe, "bzip2", [os.path.join(g, f) for f in ("bzip2_test_bz2.in", "test2_tar_bz2.in")])
    # hand-made streams: the quirks and every malformed shape of tests/streams.py, then random damage
    rnd = random.Random(3)
    raws = [streams.raw_deflate(streams.text(20000, 1)), streams.oversubscribed_dynamic_block(), streams.raw_far_reference(),
            streams.raw_deflate(bytes(rnd.getrandbits(8) for _ in range(5000)), level=0), b"", b"\x03", b"\x07"]
    base = streams.raw_deflate(streams.text(30000, 2))
    for k in range(150):
        b = bytearray(base)
        for _ in range(rnd.randrange(1, 4)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        raws.append(bytes(b[:rnd.randrange(1, len(b) + 1)] if k % 3 == 0 else b))
    paths = []
    for i, r in enumerate(raws):
        p = tmp_path / ("raw_%03d" % i)
        p.write_bytes(r)
        paths.append(p)
    _run(exe, "raw", paths)
    gz = streams.gz_member(streams.text(9000, 4)) + streams.gz_wrap(streams.raw_far_reference()) + streams.bgzf_member(streams.text(700, 5))
    gpaths = []
    for k in range(60):
        b = bytearray(gz)
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        p = tmp_path / ("gz_%03d" % k)
        p.write_bytes(bytes(b) if k % 4 else bytes(b[:rnd.randrange(len(b))]))
        gpaths.append(p)
    _run(exe, "gzip", gpaths)
    zs = zlib.compress(streams.text(7000, 6)) + zlib.compress(b"second")
    zpaths = []
    for k in range(40):
        b = bytearray(zs)
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        p = tmp_path / ("z_%03d" % k)
        p.write_bytes(bytes(b))
        zpaths.append(p)
    _run(exe, "zlib", zpaths)
    import bz2
    bz = bz2.compress(streams.text(60000, 7), 1)
    bpaths = []
    for k in range(30):
        b = bytearray(bz)
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        p = tmp_path / ("bz_%03d" % k)
        p.write_bytes(bytes(b))
        bpaths.append(p)
    _run(exe, "bzip2", bpaths)
    # the encoder: text, zeros, noise, empty
    dpaths = []
    for i, d in enumerate([streams.text(70000, 8), bytes(40000), bytes(rnd.getrandbits(8) for _ in range(20000)), b"", b"a"]):
        p = tmp_path / ("plain_%d" % i)
        p.write_bytes(d)
        dpaths.append(p)
    _run(exe, "deflate", dpaths)
