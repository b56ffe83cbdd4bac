"""ctypes front-end of tools/corpus_gen.c -- synthetic inputs for tests and bench.py.

Not product code and not the oracle: it only manufactures workloads.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "corpus_gen.c")
_LIB = os.path.join(_HERE, "_build", "libcorpus.so")

LOG, WIKI = 0, 1


def build(force=False):
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC):
        return _LIB
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _LIB, _SRC, "-lz", "-lpthread"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        L.corpus_log_text.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
        L.corpus_log_text.restype = None
        L.corpus_wiki_text.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
        L.corpus_wiki_text.restype = None
        L.corpus_make_gzip.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_void_p]
        L.corpus_make_gzip.restype = ctypes.c_size_t
        _lib = L
    return _lib


def text(kind, seed, chunk, nbytes):
    out = np.empty(nbytes, dtype=np.uint8)
    fn = lib().corpus_log_text if kind == LOG else lib().corpus_wiki_text
    fn(seed, chunk, out.ctypes.data, nbytes)
    return out


def make_gzip(kind=LOG, seed=1234, n_members=16, member_bytes=65536, level=6, bc=True, threads=None,
              first_chunk=0, want_plain=False, batch=4096):
    """Concatenated multi-member gzip of n_members x member_bytes of synthetic text.

    Returns (compressed uint8 array, plain uint8 array or None)."""
    threads = threads or min(64, os.cpu_count() or 1)
    parts = []
    plain = np.empty(n_members * member_bytes, dtype=np.uint8) if want_plain else None
    done = 0
    while done < n_members:
        k = min(batch, n_members - done)
        cap = k * (member_bytes + member_bytes // 8 + 256)
        dst = np.empty(cap, dtype=np.uint8)
        pl = plain[done * member_bytes:].ctypes.data if want_plain else None
        n = lib().corpus_make_gzip(kind, seed, first_chunk + done, k, member_bytes, level, int(bc), threads,
                                   dst.ctypes.data, cap, pl)
        if n == 0:
            raise RuntimeError("corpus_make_gzip failed")
        parts.append(dst[:n].copy())
        done += k
    comp = parts[0] if len(parts) == 1 else np.concatenate(parts)
    return comp, plain


def make_one_member(kind=WIKI, seed=8, nbytes=256 << 20, level=6, threads=None, piece=8 << 20):
    """ONE gzip member of nbytes of synthetic text, compressed the way pigz does it: pieces of `piece` bytes on `threads`
    threads, every piece primed with the 32 KiB in front of it (its matches reach back across the cut like zlib's own would)
    and closed on a byte boundary by a sync flush (an empty stored block), the last one by the final block -- laid end to end
    the pieces ARE one raw DEFLATE stream.  Returns (gzip member as uint8 array, crc32 of the text)."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    data = text(kind, seed, 0, nbytes)
    view = memoryview(data)
    cuts = list(range(0, nbytes, piece)) or [0]

    def one(i):
        lo, hi = cuts[i], min(nbytes, cuts[i] + piece)
        kw = {"zdict": bytes(view[max(0, lo - 32768):lo])} if lo else {}
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, **kw)
        return co.compress(view[lo:hi]) + co.flush(zlib.Z_FINISH if hi == nbytes else zlib.Z_SYNC_FLUSH)

    def crc(i):
        return zlib.crc32(view[cuts[i]:min(nbytes, cuts[i] + piece)])
    with ThreadPoolExecutor(threads or min(32, os.cpu_count() or 1)) as ex:
        parts = list(ex.map(one, range(len(cuts))))
    c = 0
    for i in range(len(cuts)):  # (sequential: crc32 of 1 GiB is a second)
        c = zlib.crc32(view[cuts[i]:min(nbytes, cuts[i] + piece)], c)
    gz = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255]) + b"".join(parts) + c.to_bytes(4, "little") + (nbytes & 0xffffffff).to_bytes(4, "little")
    return np.frombuffer(gz, dtype=np.uint8), c
