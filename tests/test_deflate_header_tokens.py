"""The block header's run-length tokens (deflate.dart:2820-2920 = zlib's scan_tree / send_tree) as a closed form per run.

`df_cl_tokens` (archive_amd/csrc/deflate_kernels.hpp) lets a thread per maximal run of equal code lengths write that
run's tokens, instead of one lane walking the reference's state machine (count, max_count, min_count, prevlen) over all
316 lengths twice.  That is only right if what the state machine does with a run depends on the run alone.  Here both are
written out in Python -- the reference's loop as it stands, the closed form exactly as the kernel computes it -- and
compared on every run length next to every kind of neighbour and on random length sequences; the GPU tests then hold the
kernel's bytes against zlib's inflate (tests/test_deflate_gpu.py), and its sizes did not move when it replaced the loop."""
import random


def reference_tokens(lens):
    """send_tree's loop: [(symbol, extra value)] for the code lengths lens[0 .. n)"""
    out = []
    n = len(lens)
    prevlen, nextlen, count, max_count, min_count = -1, lens[0], 0, 7, 4
    if nextlen == 0:
        max_count, min_count = 138, 3
    for i in range(n):
        curlen = nextlen
        nextlen = lens[i + 1] if i + 1 < n else 0xffff
        count += 1
        if count < max_count and curlen == nextlen:
            continue
        if count < min_count:
            out += [(curlen, 0)] * count
        elif curlen != 0:
            if curlen != prevlen:
                out.append((curlen, 0)); count -= 1
            out.append((16, count - 3))
        elif count <= 10:
            out.append((17, count - 3))
        else:
            out.append((18, count - 11))
        count, prevlen = 0, curlen
        if nextlen == 0:
            max_count, min_count = 138, 3
        elif curlen == nextlen:
            max_count, min_count = 6, 3
        else:
            max_count, min_count = 7, 4
    return out


def closed_form_tokens(lens):
    """df_cl_tokens: every maximal run (v, L) on its own"""
    out = []
    i, n = 0, len(lens)
    while i < n:
        v, L = lens[i], 1
        while i + L < n and lens[i + L] == v:
            L += 1
        if v != 0:
            first = min(L, 7)
            if first < 4:
                out += [(v, 0)] * first
            else:
                out += [(v, 0), (16, first - 1 - 3)]
            rest = L - first
            while rest >= 6:
                out.append((16, 3)); rest -= 6
            if rest >= 3:
                out.append((16, rest - 3))
            else:
                out += [(v, 0)] * rest
        else:
            rest = L
            while rest >= 138:
                out.append((18, 127)); rest -= 138
            if rest >= 11:
                out.append((18, rest - 11))
            elif rest >= 3:
                out.append((17, rest - 3))
            else:
                out += [(0, 0)] * rest
        i += L
    return out


def expand(tokens):
    """what an inflater makes of the tokens (the header must describe the lengths it was made from)"""
    lens = []
    for sym, x in tokens:
        if sym < 16:
            lens.append(sym)
        elif sym == 16:
            assert lens and 0 <= x <= 3
            lens += [lens[-1]] * (x + 3)
        elif sym == 17:
            assert 0 <= x <= 7
            lens += [0] * (x + 3)
        else:
            assert 0 <= x <= 127
            lens += [0] * (x + 11)
    return lens


def test_every_run_length_next_to_every_neighbour():
    for v in (0, 1, 7, 15):
        for L in range(1, 300):
            for before in ([], [3], [0], [3, 3, 3, 3, 3, 3, 3, 3, 3], [0] * 140, [v + 1 if v < 15 else 2]):
                for after in ([], [4], [0, 0, 0], [9] * 8):
                    if (before and before[-1] == v) or (after and after[0] == v):
                        continue  # the run would not be maximal
                    lens = before + [v] * L + after
                    if len(lens) > 320:
                        continue
                    a, b = reference_tokens(lens), closed_form_tokens(lens)
                    assert a == b, (v, L, before[:3], after[:3])
                    assert expand(a) == lens


def test_random_length_sequences():
    rnd = random.Random(8)
    for _ in range(4000):
        n = rnd.choice((1, 2, 19, 30, 257, 286))
        lens, i = [], 0
        while len(lens) < n:
            v = rnd.choice((0, 0, 0, rnd.randrange(1, 16), rnd.randrange(1, 16), 8))
            lens += [v] * rnd.choice((1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 20, 137, 138, 139, 150))
        lens = lens[:n]
        a, b = reference_tokens(lens), closed_form_tokens(lens)
        assert a == b, lens
        assert expand(a) == lens
