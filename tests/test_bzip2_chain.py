"""The bzip2 HOST chain of the product (archive_amd/csrc/bzip2_chain.hpp -- decodeStream's loop, bzip2_decoder.dart:46-87,
restated over per-block verdicts and driven batch by batch exactly as bzip2_device_impl drives it between its kernel
phases) on the CPU: tests/emu/bzip2_chain_emu.cc feeds it per-block results from the oracle's block function (mode 0) or
from the device code on the wave emulation (mode 1), and status + bytes must be the oracle's whole-stream decoder's.

This is the sweep that found round 4's open deviation on the GPU (a block that fails BEHIND bytes it has written --
`cNBlockUsed > sSaveNBlockPP`, bzip2_decoder.dart:612-631 -- lost those bytes in the chain), here without a GPU."""
import bz2
import ctypes
import os
import random
import subprocess

import pytest

from tests import streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "emu", "_build")


@pytest.fixture(scope="module")
def chain():
    os.makedirs(BUILD, exist_ok=True)
    lib = os.path.join(BUILD, "libbzchain.so")
    obj = os.path.join(BUILD, "bzip2_oracle_for_chain.o")
    src = os.path.join(ROOT, "tests", "emu", "bzip2_chain_emu.cc")
    orc = os.path.join(ROOT, "oracle", "bzip2_oracle.c")
    deps = [src, orc, os.path.join(ROOT, "tests", "emu", "wave_emu.hpp")] + [os.path.join(ROOT, "archive_amd", "csrc", f)
                                                                             for f in ("bzip2_kernels.hpp", "bzip2_chain.hpp", "common.hpp")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-c", "-o", obj, orc])
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-shared", "-fPIC", "-o", lib, src, obj])
    L = ctypes.CDLL(lib)
    L.bzchain_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32)]
    L.bzchain_decode.restype = ctypes.c_int
    L.bzchain_last_consumed.argtypes = []
    L.bzchain_last_consumed.restype = ctypes.c_size_t

    big = ctypes.create_string_buffer(6 << 20)       # (a damaged level-1 block makes up to 5.2 MB of runs)

    def run(buf, verify, batch=64, mode=0, cap=None):
        out = big if cap is None else ctypes.create_string_buffer(max(cap, 1))
        cap = len(big) if cap is None else cap
        olen, seen = ctypes.c_size_t(0), ctypes.c_uint32(0)
        st = L.bzchain_decode(bytes(buf), len(buf), int(verify), batch, mode, ctypes.addressof(out), cap, ctypes.byref(olen), ctypes.byref(seen))
        run.position = L.bzchain_last_consumed()   # what ahip_last_consumed() reports after the call
        return st, (ctypes.string_at(out, olen.value) if st in (0, 1) else olen.value), seen.value
    return run


def _expect(orc, buf, verify):
    st, out = orc.bzip2_decode(buf, verify=verify)
    _expect.position = orc.bzip2_last_position()   # where decodeStream left its InputStream
    return st, out


def _two_blocks():
    data = streams.text(110000, 9) + bytes(3000) + streams.text(2000, 10)
    return bz2.compress(data, 1), data      # two blocks


def test_valid_and_malformed_streams(chain):
    from oracle import pyoracle as orc
    from tests.test_bzip2 import _malformed
    c, data = _two_blocks()
    for batch in (1, 2, 3, 64):
        assert chain(c, True, batch)[:2] == (0, data)
    for name, buf in _malformed().items():
        for verify in (False, True):
            st, out = _expect(orc, buf, verify)
            got = chain(buf, verify, 2)
            assert got[0] == st and (st == 2 or got[1] == out), (name, verify, got[0], st)
            if st in (0, 1):   # (a RangeError leaves no position)
                assert chain.position == _expect.position, (name, verify, st, chain.position, _expect.position)


def test_two_block_stream_flipped_everywhere(chain):
    """Every 37th bit of a two-block stream flipped in turn (the sweep of tests/test_bzip2.py's GPU test, which found the
    deviation) and, denser, every 5th bit of the first block's last kilobit and of the trailer; batches of one, two and
    many candidates.  The partial-output case must occur."""
    from oracle import pyoracle as orc
    c, _ = _two_blocks()
    bits = list(range(0, len(c) * 8, 37))
    partial = 0
    n = 0
    whole = chain(c, False)[1]
    for k, bit in enumerate(bits):
        buf = bytearray(c); buf[bit >> 3] ^= 0x80 >> (bit & 7); buf = bytes(buf)
        for verify in (False, True):
            st, out = _expect(orc, buf, verify)
            batch = 1 if (k + verify) % 2 else 64      # (batches of one candidate and of all of them, in turn)
            got = chain(buf, verify, batch)
            if got[0] == -3:       # the obsolete randomised mode: not a verdict
                continue
            assert got[0] == st and (st == 2 or got[1] == out), (bit, verify, batch, got[0], st, len(out))
            if st in (0, 1):   # the InputStream's position after decodeStream, `true` or `false` (bz2_bit_reader.dart:12-44)
                assert chain.position == _expect.position, (bit, verify, batch, st, chain.position, _expect.position)
            n += 1
            if st == 1 and not verify and len(out) > 0:
                # did the last block the reference touched fail behind its own bytes?
                if out != whole[:len(out)] or len(out) not in (0, len(whole)):
                    partial += 1
    assert n > 6000
    assert partial > 0


def test_crc_mismatch_in_front_of_a_later_verdict(chain):
    """`verify`: the reference meets the first block's bad CRC before it ever reads the second block -- whatever that one's
    verdict (RangeError of a truncated stream, `false` of a damaged one) it must not be reported instead."""
    from oracle import pyoracle as orc
    c, _ = _two_blocks()
    rnd = random.Random(4)
    seen = set()
    for t in range(60):
        buf = bytearray(c)
        bit = 200 * 8 + rnd.randrange(3000 * 8)          # payload of the first block: it decodes, to other bytes
        buf[bit >> 3] ^= 0x80 >> (bit & 7)
        if t % 2:
            buf = buf[:len(buf) - rnd.randrange(12, 2000)]   # ... and the second block runs off the end
        else:
            b2 = (len(buf) - 600) * 8 + rnd.randrange(500 * 8)
            buf[b2 >> 3] ^= 0x80 >> (b2 & 7)
        buf = bytes(buf)
        for verify in (False, True):
            st, out = _expect(orc, buf, verify)
            for batch in (1, 2, 64):
                got = chain(buf, verify, batch)
                assert got[0] == st and (st == 2 or got[1] == out), (t, verify, batch, got[0], st)
            seen.add((verify, st))
    assert (True, 1) in seen and (False, 2) in seen


def test_what_reads_the_block_type_where_no_magic_is(chain):
    """_readBlockType compares byte by byte: garbage behind a block is `false` at its first byte even when fewer than six
    bytes are left; a cut inside a magic is a RangeError (bzip2_decoder.dart:90-111)."""
    from oracle import pyoracle as orc
    c = bz2.compress(b"hello world" * 1000)
    body = c[:-10]                                       # up to the end-of-stream magic (byte aligned here or not: both work)
    cases = [b"BZh9" + bytes([x]) for x in (0x31, 0x17, 0x00)] + [b"BZh9\x31\x41", b"BZh9\x31\x42", b"BZh9\x17\x72\x45\x38\x50", b"BZh9\x17\x72\x45\x38\x51"]
    for k in range(1, 12):
        cases += [c[:len(c) - k], c[:len(c) - k] + b"\xff", body + bytes([0x17, 0x72][:k % 3]) + b"\x99" * (k % 4)]
    for buf in cases:
        for verify in (False, True):
            st, out = _expect(orc, buf, verify)
            got = chain(buf, verify)
            assert got[0] == st and (st == 2 or got[1] == out), (buf[-12:], verify, got[0], st)


def test_nothing_behind_the_end_of_the_chain_is_decoded(chain):
    """Two streams back to back: the reference decodes ONE (returns at the first end-of-stream block); with batches of one
    candidate the second stream's blocks are never looked at."""
    c, data = _two_blocks()
    st, out, seen = chain(c + c, True, 1)
    assert (st, out) == (0, data) and seen == 3          # two blocks + the end-of-stream marker
    # a too-small buffer is told the full size
    st, need, _ = chain(c, True, 1, cap=1000)
    assert st == -1 and need == len(data)


def test_device_code_on_the_wave_emulation_feeds_the_chain(chain):
    """Mode 1: per-block verdicts from the device functions (Huffman side, move-to-front side, the serial inverse transform
    bz_unbwt_block the bz_unbwt kernel wraps) on the 64-thread wave emulation, a handful of streams."""
    from oracle import pyoracle as orc
    data = streams.text(9000, 3) + bytes(700) + b"abcd" * 40
    c = bz2.compress(data, 1)
    assert chain(c, True, 64, mode=1)[:2] == (0, data)
    rnd = random.Random(6)
    for t in range(6):
        buf = bytearray(c)
        bit = 45 * 8 + rnd.randrange((len(c) - 60) * 8)
        buf[bit >> 3] ^= 0x80 >> (bit & 7)
        buf = bytes(buf)
        st, out = _expect(orc, buf, True)
        got = chain(buf, True, 64, mode=1)
        if got[0] == -3:
            continue
        assert got[0] == st and (st == 2 or got[1] == out), (t, got[0], st)
        if st in (0, 1):
            assert chain.position == _expect.position, (t, bit, st, chain.position, _expect.position)


def test_a_failing_getmtfval_is_not_the_end_of_the_block(chain):
    """_getMtfVal returns -1 (selectors used up, a code longer than 20 bits, an index outside the alphabet) and the
    reference's loop does not look (bzip2_decoder.dart:304, :385): it goes on with -1 as a symbol.  Damage in the code
    lengths makes such codes.  bz_block_exact_lane -- what the device runs for those blocks -- against the oracle: every 3rd
    bit of the header and code-length area and every 29th bit of the rest through it (mode 2, every block), and a sample
    through the whole device pipeline on the wave emulation (mode 1), where only the blocks that need it take it."""
    from oracle import pyoracle as orc
    data = streams.text(9000, 3) + bytes(700) + b"abcd" * 40
    c = bz2.compress(data, 1)
    assert chain(c, True, 64, mode=2)[:2] == (0, data)
    seen = set()
    differs_from_stopping = 0
    bits = list(range(14 * 8 + 1, 1400, 3)) + list(range(1400, len(c) * 8 - 80, 29))
    for bit in bits:
        buf = bytearray(c); buf[bit >> 3] ^= 0x80 >> (bit & 7); buf = bytes(buf)
        st, out = _expect(orc, buf, False)
        got = chain(buf, False, 64, mode=2)
        if got[0] == -3:
            continue
        assert got[0] == st and (st == 2 or got[1] == out), (bit, got[0], st)
        if st in (0, 1):
            assert chain.position == _expect.position, (bit, st, chain.position, _expect.position)
        seen.add(st)
    assert seen == {0, 1, 2}
    rnd = random.Random(12)
    for bit in [479] + [rnd.randrange(400, 1300) for _ in range(14)]:
        buf = bytearray(c); buf[bit >> 3] ^= 0x80 >> (bit & 7); buf = bytes(buf)
        st, out = _expect(orc, buf, True)
        got = chain(buf, True, 64, mode=1)
        if got[0] == -3:
            continue
        assert got[0] == st and (st == 2 or got[1] == out), (bit, got[0], st)
        if st in (0, 1):
            assert chain.position == _expect.position, (bit, st, chain.position, _expect.position)
