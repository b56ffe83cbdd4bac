"""Dev tool (GPU box): is the Deflate output of two builds of the library the same, byte for byte?

    python tools/df_same_bytes.py archive_amd/lib/var_prev.so archive_amd/lib/libarchive_hip.so

A change that only makes the match kernel cheaper must not move a single byte: every corpus of the size tests, random
and degenerate inputs, levels 1..9, three window sizes, through each library in a process of its own (AHIP_LIB), the
CRC-32 and length of every stream compared."""
import json, os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dump():
    sys.path.insert(0, ROOT)
    import random
    import archive_amd
    from archive_amd import _native as N
    from tests import streams
    from tools import corpus
    assert N.lib().ahip_init(0) == 0
    rnd = random.Random(77)
    C = {"text12": streams.text(200000, 2), "log8M": bytes(corpus.text(corpus.LOG, 1234, 0, 8 << 20)), "wiki4M": bytes(corpus.text(corpus.WIKI, 8, 0, 4 << 20)),
         "random": rnd.randbytes(300000), "zeros": bytes(500000), "ab": b"ab" * 150000, "short": b"hello hello hello hello", "one": b"x",
         "mixed": b"".join(rnd.choice([streams.text(rnd.randrange(1, 5000), i), rnd.randbytes(rnd.randrange(1, 3000)), bytes(rnd.randrange(1, 4000))]) for i in range(400))}
    out = {}
    for name, d in C.items():
        for level in range(1, 10):
            for wb in (15, 12, 9):
                if wb != 15 and (level not in (1, 6, 9) or len(d) > (1 << 20)):
                    continue
                z = archive_amd.Deflate(d, level=level, window_bits=wb).get_bytes()
                assert zlib.decompress(z, -15) == d
                out["%s/L%d/w%d" % (name, level, wb)] = (len(z), zlib.crc32(z))
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--dump":
        dump()
        sys.exit(0)
    res = []
    for lib in sys.argv[1:3]:
        env = dict(os.environ, AHIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dump"], env=env, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:])
            sys.exit(1)
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b = res
    diff = [k for k in a if a[k] != b.get(k)]
    print("%d streams, %d differ" % (len(a), len(diff)))
    for k in diff[:20]:
        print("  ", k, a[k], b.get(k))
    sys.exit(1 if diff or len(a) != len(b) else 0)
