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
