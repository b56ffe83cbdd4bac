"""The parse kernel's walk without a hop loop (archive_amd/csrc/deflate_kernels.hpp, deflate_parse_kernel).

The reference takes a match or a literal and moves on by its length (`_deflateSlow`, deflate.dart:997-1118): the positions it
visits are the orbit of position 0 under "next position".  The kernel marks the orbit inside a block of 63 positions wave-wide:
every lane knows the lane it hops to (J; a capped match and whatever lies behind the block's end hop to themselves), and the set S
of visited lanes grows by S |= J^(2^k)(S), k = 0 .. 5 -- lanes of S send a mark to their target with ds_permute (the other lanes
send theirs to the entry, which is marked anyway), then J is composed with itself with ds_bpermute.  Here the same steps are
written out in Python with the two instructions' semantics (ds_permute: unwritten lanes read 0, the highest writer wins;
ds_bpermute: a gather) and held against the plain hop loop on random blocks: steps of every size, absorbing lanes anywhere, every
entry.  The GPU side is held by tools/df_same_bytes.py (same bytes as the hop loop's build) and the round-trip tests."""
import random

W = 64


def ds_permute(dest, data):
    out = [0] * W
    for lane in range(W):  # ascending: the highest writer wins
        out[dest[lane] % W] = data[lane]
    return out


def ds_bpermute(src, data):
    return [data[src[lane] % W] for lane in range(W)]


def orbit_wave(entry, step, absorb):
    """what the kernel computes: (set of marked lanes as a list of 0/1, the J it ends with)"""
    J = [lane if absorb[lane] else min(lane + step[lane], 63) for lane in range(W)]
    S = [1 if lane == entry else 0 for lane in range(W)]
    for r in range(6):
        got = ds_permute([J[lane] if S[lane] else entry for lane in range(W)], [1] * W)
        S = [S[lane] | got[lane] for lane in range(W)]
        if r < 5:
            J = ds_bpermute(J, J)
    return S


def orbit_hops(entry, step, absorb):
    S = [0] * W
    j = entry
    while True:
        S[j] = 1
        if absorb[j]:
            return S
        j = min(j + step[j], 63)


def test_orbit_by_doubling_is_the_hop_loop():
    rnd = random.Random(41)
    for case in range(3000):
        kind = case % 5
        step = [1 if kind == 0 else rnd.choice([1, 1, 1, 2, 3, 4, 5, 8, 13, 32, 70, 258]) if kind < 4 else rnd.randrange(1, 259) for _ in range(W)]
        absorb = [lane == 63 or rnd.random() < (0.0, 0.02, 0.1, 0.3, 0.05)[kind] for lane in range(W)]
        if kind == 3:  # the chunk's last block: nothing behind some lane is a position
            cut = rnd.randrange(1, 63)
            absorb = [a or lane >= cut for lane, a in enumerate(absorb)]
        for entry in ([0, 1, 62] + [rnd.randrange(63) for _ in range(4)]):
            assert orbit_wave(entry, step, absorb) == orbit_hops(entry, step, absorb), (case, entry)


def test_all_literals_take_every_round():
    # 62 hops of one: the longest orbit a block can hold needs all six rounds (2^6 > 62)
    step, absorb = [1] * W, [lane == 63 for lane in range(W)]
    assert orbit_wave(0, step, absorb) == [1] * W
    J = [min(lane + 1, 63) for lane in range(W)]
    S = [1] + [0] * 63
    for r in range(5):  # one round short: not everything is marked yet
        got = ds_permute([J[lane] if S[lane] else 0 for lane in range(W)], [1] * W)
        S = [a | b for a, b in zip(S, got)]
        J = ds_bpermute(J, J)
    assert sum(S) == 32
