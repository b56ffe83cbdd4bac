"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle, the
committed golden vectors and size-independent properties."""
import ctypes
import gzip
import hashlib
import os
import random
import zlib

import pytest

pytestmark = pytest.mark.gpu

from tests import streams  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def amd(native_built):
    import archive_amd
    from archive_amd import _native as N
    assert N.lib().ahip_init(0) == 0, N.last_error()
    return archive_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def _raw(amd, data):
    from archive_amd import errors
    try:
        z = amd.Inflate(data)
        return z.status, z.get_bytes(), z.input_position
    except errors.RangeError:
        return 2, None, None
    except errors.ReferenceWouldHang:
        return 3, None, None


_last_pos = [None]  # input position after the last _gz / _zl call that returned bytes (what decodeStream consumed)


def _gz(amd, data, **kw):
    from archive_amd import errors
    d = amd.GZipDecoder()
    try:
        out = d.decode_bytes(data, **kw)
        _last_pos[0] = d.input_position
        return d.last_status, out
    except errors.RangeError:
        return 2, None


def _zl(amd, data, **kw):
    from archive_amd import errors
    d = amd.ZLibDecoder()
    try:
        out = d.decode_bytes(data, **kw)
        _last_pos[0] = d.input_position
        return d.last_status, out
    except errors.RangeError:
        return 2, None


def test_golden_vectors(amd, golden):
    for v in golden["vectors"]:
        kind, data, exp = v["kind"], v["input"], v["expected"]
        if kind == "raw":
            st, out, _ = _raw(amd, data)
        elif kind == "gzip":
            st, out = _gz(amd, data)
        elif kind == "bzip2":
            d = amd.BZip2Decoder()
            out = d.decode_bytes(data, verify=True)
            st = d.last_status
        else:
            st, out = _zl(amd, data, verify=True)
        assert st == 0, v["name"]
        assert out == exp, v["name"]
        assert hashlib.sha256(out).hexdigest() == v["sha256"]


@pytest.mark.parametrize("name,raw", streams.valid_raw_streams())
def test_valid_raw_streams_match_oracle(amd, orc, name, raw):
    st, out, pos = _raw(amd, raw)
    ost, oout, opos = orc.inflate_raw(raw)
    assert (st, pos) == (ost, opos)
    assert out == oout
    st, out, pos = _raw(amd, raw + bytes(4))
    assert (st, out, pos) == orc.inflate_raw(raw + bytes(4))


def test_malformed_raw_streams_match_oracle(amd, orc):
    for name, raw in streams.malformed_raw_streams():
        st, out, pos = _raw(amd, raw)
        ost, oout, opos = orc.inflate_raw(raw)
        assert st == ost, name
        if ost in (0, 1):
            assert out == oout, name
            assert pos == opos, name


def test_gzip_framing_matches_oracle(amd, orc):
    a, b = streams.text(1000, 1), streams.text(70000, 2)
    g = streams.gz_member(a, name=b"a.txt", comment=b"hi", hcrc=True) + streams.gz_member(b, extra=b"XY\x02\x00zz") \
        + streams.bgzf_member(a)
    cases = [g, g + bytes(5), g[:-3], zlib.compress(a), b"", streams.bgzf_member(a) * 3,
             streams.bgzf_member(a) + g, g + zlib.compress(b), b"\x1f\x8b\x08", b"junk" + g,
             b"\x1f\x8b", g + b"\x1f\x8b", g + b"\x1f", b"\x1f", g + b"\x1f\x8b\x08\x00", g + b"\x1f\x8b\x07",
             g + b"\x1f\x8c"]
    for i, c in enumerate(cases):
        want = _noneify(orc.gzip_decode(c))
        try:
            got = _gz(amd, c)
        except Exception as e:  # keep the case index visible
            raise AssertionError("case %d: %r (oracle %r)" % (i, e, want[0]))
        assert got == want, i
        if want[0] in (0, 1):  # ... and the InputStream stands where the reference leaves it (decodeStream consumes it)
            assert _last_pos[0] == orc.last_position(), (i, _last_pos[0], orc.last_position(), len(c))
    assert _gz(amd, zlib.compress(a), verify=True) == _noneify(orc.gzip_decode(zlib.compress(a), verify=True))


def _noneify(t):
    st, out = t
    return (st, None) if st == 2 else (st, out)


def test_zlib_framing_matches_oracle(amd, orc):
    a, b = streams.text(3000, 3), streams.text(4000, 4)
    za, zb = zlib.compress(a), zlib.compress(b)
    bad = bytearray(za + zb)
    bad[-1] ^= 1
    for data, kw in [(za + zb, dict(verify=True)), (za + b"\x00\x00", dict(verify=True)), (za + zb + b"\x01\x02", {}),
                     (bytes(bad), dict(verify=True)), (bytes(bad), {}), (streams.raw_deflate(a), dict(raw=True)),
                     (za[:-2], {}), (b"", {}),
                     # where `false` leaves the stream: behind the two header bytes (method / FCHECK), behind the dictionary id,
                     # behind the Adler-32 that did not match
                     (za + b"\x77\x01" + zb, {}), (za + b"\x78\x02" + zb, {}), (za + b"\x78\x20" + zb, {}), (za + b"\x78\x20\x00", {}),
                     (b"\x78\xbb" + bytes(9), {})]:
        want = _noneify(orc.zlib_decode(data, **kw))
        assert _zl(amd, data, **kw) == want, (len(data), kw)
        if want[0] in (0, 1):
            assert _last_pos[0] == orc.last_position(), (len(data), kw, _last_pos[0], orc.last_position())


def test_bsize_or_isize_lies_are_caught(amd, orc):
    """A wrong BC/ISIZE must not change the result (the reference ignores both)."""
    a, b = streams.text(5000, 5), streams.text(6000, 6)
    m1, m2 = bytearray(streams.bgzf_member(a)), streams.bgzf_member(b)
    wrong_isize = bytes(m1[:-4]) + (len(a) + 7).to_bytes(4, "little") + m2
    assert _gz(amd, wrong_isize) == orc.gzip_decode(wrong_isize) == (0, a + b)
    m1[16] ^= 0x10  # BSIZE points into the middle of nowhere
    wrong_bsize = bytes(m1) + m2
    assert _gz(amd, wrong_bsize) == orc.gzip_decode(wrong_bsize) == (0, a + b)


def test_multimember_log_text_matches_oracle_and_zlib(amd, orc):
    from tools import corpus
    for bc in (True, False):
        comp, plain = corpus.make_gzip(n_members=257, bc=bc, want_plain=True)
        st, out = _gz(amd, bytes(comp))
        assert st == 0
        assert out == bytes(plain)
        assert orc.gzip_decode(bytes(comp), cap=len(plain) + 8) == (0, bytes(plain))


def test_wiki_text_dynamic_blocks(amd):
    from tools import corpus
    comp, plain = corpus.make_gzip(kind=corpus.WIKI, seed=8, n_members=64, want_plain=True)
    st, out = _gz(amd, bytes(comp))
    assert st == 0 and out == bytes(plain)


def test_ragged_members(amd):
    """empty, 1-byte, stored, incompressible and >64 KiB members in one stream"""
    import random
    rnd = random.Random(3)
    parts = [b"", b"x", streams.text(100, 1), bytes(rnd.getrandbits(8) for _ in range(70000)), streams.text(300000, 2),
             bytes(200000), b"", streams.text(65536, 3)]
    g = b"".join(streams.gz_member(p, level=(0 if i == 2 else 6)) for i, p in enumerate(parts))
    assert _gz(amd, g) == (0, b"".join(parts))
    assert gzip.decompress(g) == b"".join(parts)


@pytest.mark.parametrize("n_members", [40, 3500])
def test_false_member_magics_inside_payloads(amd, orc, n_members):
    """`1f 8b 08` inside stored payloads creates false member candidates; the chain must hop over them
    (few: exception list; thousands: pointer-doubling fallback of gz_chain)."""
    parts = [b"\x1f\x8b\x08\x00" + bytes([i & 255, i >> 8]) * 9 + streams.text(40 + i % 7, i) for i in range(n_members)]
    g = b"".join(streams.gz_member(p, level=0) if i % 3 else streams.bgzf_member(p) for i, p in enumerate(parts))
    want = b"".join(parts)
    assert _gz(amd, g) == (0, want)
    if n_members < 100:
        assert orc.gzip_decode(g, cap=len(want) + 8) == (0, want)


def test_output_groups(native_built):
    """Streams larger than one launch's output budget are decoded group by group (6 GiB in production; shrunk here
    through AHIP_GROUP_OUT_MAX, in a fresh process because the library reads it once)."""
    import subprocess
    import sys
    code = r'''
import sys, ctypes
sys.path.insert(0, %r)
import archive_amd
from tests import streams
from tools import corpus
comp, plain = corpus.make_gzip(n_members=300, want_plain=True)            # 300 x 64 KiB, groups of <= 1 MiB
st_out = archive_amd.GZipDecoder().decode_bytes(bytes(comp))
assert st_out == bytes(plain)
big = streams.text(3000000, 9)                                             # one member larger than a group
g = streams.gz_member(b"head") + streams.gz_member(big) + streams.gz_member(b"tail")
assert archive_amd.GZipDecoder().decode_bytes(g) == b"head" + big + b"tail"
print("groups ok")
''' % ROOT
    env = dict(os.environ, AHIP_GROUP_OUT_MAX=str(1 << 20))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "groups ok" in r.stdout, r.stdout + r.stderr


def test_workgroup_resolver_is_bit_exact(native_built):
    """The resolver as a workgroup per member (inflate_res_wg.hpp, AHIP_RES_WG=1; the library reads the switch once, so a
    fresh process): DEFLATE's whole reach in one LDS ring shared by four waves, chunks completing in stream order.  Members
    of the benchmark's kind against their plain text; streams without size hints (the kept tokens of the sizing run); every
    valid stream shape of tests/streams.py against the oracle -- zeros (a chunk of 64 matches is 16 KiB: token by token), periods
    shorter than a match, stored blocks, members of a few bytes; many members so that every output alignment occurs."""
    import subprocess
    import sys
    code = r'''
import sys, zlib, random
sys.path.insert(0, %r)
import archive_amd
from tests import streams
from tools import corpus
comp, plain = corpus.make_gzip(n_members=200, want_plain=True)
assert archive_amd.GZipDecoder().decode_bytes(bytes(comp)) == bytes(plain)
comp, plain = corpus.make_gzip(n_members=70, want_plain=True, bc=False)      # no size hints: resolved from the sizing run's tokens
assert archive_amd.GZipDecoder().decode_bytes(bytes(comp)) == bytes(plain)
rnd = random.Random(3)
noise = bytes(rnd.getrandbits(8) for _ in range(70000))
parts = [bytes(200000), b"abc" * 40000, b"0123456789" * 7000, noise, streams.text(150000, 4), b"", b"x", b"xy" * 3,
         (streams.text(5000, 4) + noise[:3000]) * 9, bytes(range(256)) * 300, streams.text(66000, 5)]
for lv in (0, 1, 6, 9):
    g = b"".join(streams.gz_member(p, level=lv) for p in parts)
    assert archive_amd.GZipDecoder().decode_bytes(g) == b"".join(parts), lv
from oracle import pyoracle
for name, raw in streams.valid_raw_streams():
    z = archive_amd.Inflate(raw)
    assert (z.status, z.get_bytes(), z.input_position) == pyoracle.inflate_raw(raw), name
odd = [streams.text(1 + 37 * i, i) for i in range(300)]                        # ragged members: every output alignment
assert archive_amd.GZipDecoder().decode_bytes(b"".join(streams.bgzf_member(p) for p in odd)) == b"".join(odd)
print("wg resolver ok")
''' % ROOT
    env = dict(os.environ, AHIP_RES_WG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "wg resolver ok" in r.stdout, r.stdout + r.stderr


def test_device_resident_plan_api(amd):
    import torch
    from archive_amd import _native as N
    from tools import corpus
    comp, plain = corpus.make_gzip(n_members=512, want_plain=True)
    d_in = torch.from_numpy(comp).cuda()
    plan = ctypes.c_void_p()
    assert N.lib().ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), None, ctypes.byref(plan)) == 0, N.last_error()
    members, out_bytes, payload = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    N.lib().ahip_gzip_plan_info(plan, ctypes.byref(members), ctypes.byref(out_bytes), ctypes.byref(payload))
    assert (members.value, out_bytes.value, payload.value) == (512, len(plain), len(comp))
    d_out = torch.zeros(len(plain), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert N.lib().ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), stream) == 0, N.last_error()
    torch.cuda.synchronize()
    olen = ctypes.c_size_t()
    assert N.lib().ahip_gzip_plan_status(plan, ctypes.byref(olen)) == 0
    assert olen.value == len(plain)
    assert bytes(d_out.cpu().numpy()) == bytes(plain)
    N.lib().ahip_gzip_plan_destroy(plan)
    # one-call device API, too-small buffer first
    small = torch.zeros(10, dtype=torch.uint8, device="cuda")
    assert N.lib().ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), small.data_ptr(), 10, ctypes.byref(olen), None) == -1
    assert olen.value == len(plain)
    d_out.zero_()
    assert N.lib().ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(),
                                           ctypes.byref(olen), None) == 0
    assert hashlib.sha256(d_out.cpu().numpy().tobytes()).digest() == hashlib.sha256(bytes(plain)).digest()


def test_back_reference_into_previous_gzip_member(amd, orc):
    """Quirk q8: the reference's gzip decoder appends every member to ONE OutputStream (_gzip_decoder_web.dart:27-41), so
    a distance may reach into what earlier members produced (util/output_memory_stream.dart:79-98)."""
    first = streams.text(40000, 77)
    far = streams.gz_wrap(streams.raw_far_reference())               # 'a' + copy 3 bytes from 6 back
    far2 = streams.gz_wrap(streams.raw_far_reference(lit=b"xy", length=5, dist_extra=0))
    cases = [streams.gz_member(first) + far,
             streams.gz_member(first) + far + far2 + streams.gz_member(streams.text(3000, 78)),  # a chain of them
             streams.bgzf_member(first[:5000]) + far + streams.bgzf_member(first[:700]) + far2,
             streams.gz_member(b"12345") + far,                          # reaches exactly the first byte of the stream
             streams.gz_member(b"1234") + far,                           # one byte too far: RangeError
             far]                                                        # first member: RangeError
    for i, c in enumerate(cases):
        want = _noneify(orc.gzip_decode(c))
        assert _gz(amd, c) == want, i
    assert orc.gzip_decode(cases[0])[1].endswith(b"a" + first[-5:-2])


def test_back_reference_into_a_long_member(amd, orc):
    """q8 next to a LONG member (>= 2 MiB compressed: decoded by many waves, outside the member launch): a member behind
    it that reaches into its output is resolved when those bytes exist -- with and without the BGZF size fields."""
    import random
    rnd = random.Random(5)
    words = [bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(3, 9))) for _ in range(6000)]
    big = b" ".join(rnd.choice(words) for _ in range(900000))          # ~ 5.8 MB, compresses to > 2 MiB
    comp_big = streams.gz_member(big, level=1)
    assert len(comp_big) > (2 << 20) + 4096
    far = streams.gz_wrap(streams.raw_far_reference())               # 'a' + copy 3 bytes from 6 back
    far2 = streams.gz_wrap(streams.raw_far_reference(lit=b"xy", length=5, dist_extra=0))
    small = streams.text(30000, 9)
    cases = [streams.gz_member(small) + comp_big + far + streams.gz_member(small[:777]) + far2,
             comp_big + far,
             streams.gz_member(small) + far + comp_big + far2 + comp_big + far]
    for i, c in enumerate(cases):
        want = _noneify(orc.gzip_decode(c, cap=3 * len(big) + (1 << 20)))
        assert want[0] == 0 and want[1].count(b"a" + big[-5:-2]) >= 1, i
        assert _gz(amd, c) == want, i


def test_long_member_reaching_into_earlier_members(amd, orc):
    """q8 ON a long member (>= 2 MiB compressed: the chunked path): its back-references reach into what the members in front
    of it wrote -- the reference appends all members to one OutputStream (_gzip_decoder_web.dart:27-41,
    output_memory_stream.dart:79-98), so that is legal there.  The stream is made with a preset dictionary = the tail of
    the output in front of the member: its first matches copy from there.  Also an ordinary member behind it that reaches
    back into IT, a second long member, and a reach that goes in front of the whole output (the reference's RangeError)."""
    import random
    rnd = random.Random(7)
    words = [bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(3, 9))) for _ in range(6000)]
    front = b" ".join(rnd.choice(words) for _ in range(9000))            # ~ 58 KB in front of the long member
    big = b" ".join(rnd.choice(words) for _ in range(900000))            # ~ 5.8 MB, compresses to > 2 MiB

    def member_with_history(data, history, level=1):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, history[-32768:])
        raw = co.compress(data) + co.flush()
        return streams.gz_wrap(raw, data)

    long_q8 = member_with_history(front[-20000:] + big, front)           # starts by copying 20 000 bytes out of `front`
    assert len(long_q8) > (2 << 20) + 4096
    far = streams.gz_wrap(streams.raw_far_reference())
    small = streams.text(30000, 9)
    cases = [streams.gz_member(front) + long_q8,
             streams.gz_member(small) + streams.gz_member(front) + long_q8 + far + streams.gz_member(small[:999]),
             streams.gz_member(front) + long_q8 + member_with_history(big[-5000:] + big[:3000000], front[-20000:] + big),
             streams.gz_member(front[-9000:]) + long_q8]                 # reaches in front of the whole output: RangeError
    for i, c in enumerate(cases):
        want = _noneify(orc.gzip_decode(c, cap=3 * len(big) + (1 << 20)))
        assert want[0] == (2 if i == 3 else 0), (i, want[0])
        if i < 3:
            assert front[-20000:] + big[:1000] in want[1]
        assert _gz(amd, c) == want, i


def test_over_subscribed_code_lengths(amd, orc):
    """HuffmanTable never checks the Kraft sum (_huffman_table.dart:24-45): later codes overwrite earlier ones."""
    raw = streams.oversubscribed_dynamic_block()
    assert orc.inflate_raw(raw)[:2] == (0, b"ccbb")
    assert _raw(amd, raw) == orc.inflate_raw(raw)
    for bits in [(1, 1, 0, 1, 0), (0, 1, 0), (1, 0), (0,) * 40 + (1, 0), (1,) * 9]:
        r = streams.oversubscribed_dynamic_block(bits)
        assert _raw(amd, r)[:2] == orc.inflate_raw(r)[:2], bits
    g = streams.gz_member(streams.text(20000, 5)) + streams.gz_wrap(raw[:-4], b"ccbb") + streams.gz_member(streams.text(9000, 6))
    assert _gz(amd, g) == _noneify(orc.gzip_decode(g))


def test_position_after_a_failed_block(amd, orc):
    """After `return -1` inside a block the reference does not un-read its accumulator (inflate.dart:159-211,300-343), so
    what follows a damaged gzip member is parsed from wherever its reader stopped."""
    import random
    rnd = random.Random(11)
    good = streams.raw_deflate(streams.text(30000, 31), strategy=zlib.Z_FIXED)  # the fixed code HAS invalid symbols
    tail = streams.gz_member(streams.text(2000, 32)) + streams.gz_member(streams.text(100, 33))
    raws = []
    for k in range(60):
        b = bytearray(good)
        p = rnd.randrange(20, len(b) - 8)
        b[p] ^= 1 << rnd.randrange(8)
        b[p + 1] ^= 0x55
        raws.append(bytes(b))
    raws += [streams._fixed_bits([(0b11000110, 8)]) + bytes(3), streams._fixed_bits([(0x30 + 0x61, 8), (1, 7), (30, 5)]) + bytes(3),
             bytes([0x07, 1, 2, 3]), bytes([1, 5, 0, 0, 0, 1, 2, 3, 4, 5]), bytes([0b101, 0xff, 0xff, 0xff, 0xff]) + bytes(8)]
    n_false = 0
    for i, raw in enumerate(raws):
        ost, oout, opos = orc.inflate_raw(raw + tail)
        st, out, pos = _raw(amd, raw + tail)
        assert (st, out, pos) == ((ost, oout, opos) if ost in (0, 1) else (ost, None, None)), i
        n_false += ost == 1
        g = streams.gz_wrap(raw) + tail
        want = orc.gzip_decode(g)
        if want[0] != 3:
            assert _gz(amd, g) == _noneify(want), i
    assert n_false >= 8


def test_kept_tokens_of_the_sizing_run(amd):
    """Members without BC: the sizing run keeps its tokens (laid out along the input) and the decode proper resolves them.
    Members whose token area a false candidate cut short (a stored block holding `1f 8b 08` in front of the Huffman
    blocks) are tokenized again in a launch of their own; a plan run a second time resolves the kept tokens again;
    after another call has used the token scratch the plan falls back to tokenizing everything."""
    import torch
    from archive_amd import _native as N
    L = N.lib()
    parts, blobs = [], []
    for i in range(200):
        body = streams.text(20000 + 211 * (i % 37), 100 + i)
        if i % 19 == 3:  # false candidate 5 bytes into the payload: the real candidate's area ends there
            head = b"\x1f\x8b\x08\x00" + bytes(range(i % 200, i % 200 + 20))
            raw = streams.stored_block(head, final=False) + streams.raw_deflate(body)
            parts.append(head + body)
            blobs.append(streams.gz_wrap(raw, head + body))
        else:
            parts.append(body)
            blobs.append(streams.gz_member(body))
    g, want = b"".join(blobs), b"".join(parts)
    assert gzip.decompress(g) == want
    d_in = torch.frombuffer(bytearray(g), dtype=torch.uint8).cuda()
    plan = ctypes.c_void_p()
    assert L.ahip_gzip_plan_create(d_in.data_ptr(), d_in.numel(), None, ctypes.byref(plan)) == 0, N.last_error()
    members, out_bytes, payload = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    L.ahip_gzip_plan_info(plan, ctypes.byref(members), ctypes.byref(out_bytes), ctypes.byref(payload))
    assert (members.value, out_bytes.value) == (200, len(want))
    d_out = torch.zeros(len(want), dtype=torch.uint8, device="cuda")
    olen = ctypes.c_size_t()

    def run_and_check():
        d_out.zero_()
        assert L.ahip_gzip_plan_run(plan, d_out.data_ptr(), d_out.numel(), None) == 0, N.last_error()
        torch.cuda.synchronize()
        assert L.ahip_gzip_plan_status(plan, ctypes.byref(olen)) == 0 and olen.value == len(want)
        assert bytes(d_out.cpu().numpy()) == want

    run_and_check()   # kept tokens + the listed members
    run_and_check()   # the kept tokens once more
    other = b"".join(streams.gz_member(streams.text(30000, 900 + i)) for i in range(8))
    assert _gz(amd, other)[0] == 0   # another decode writes the token scratch
    run_and_check()   # the plan notices and tokenizes again
    L.ahip_gzip_plan_destroy(plan)
    assert _gz(amd, g) == (0, want)  # and the one-call path


def test_damaged_gzip_framing_matches_oracle(amd, orc):
    """The member index parses headers on the device (gz_parse_headers: the fixed fields and short FEXTRA from four
    registers gathered while the scan streams by, FNAME / FCOMMENT through a zero-byte scan 64 bytes at a time -- both new
    in round 4) and everything after it trusts what it found.  Streams whose headers use every optional field -- names and
    comments shorter and longer than the gathered 32 bytes and than the 64-byte scan step, extras with and without a `BC`
    subfield, header CRCs -- cut at every byte of the first member's header and around every member boundary, and with every
    header byte damaged in turn: verdict and bytes are the oracle's (`_readHeader`, _gzip_decoder_web.dart:60-138)."""
    rnd = random.Random(3)
    a, b, c = streams.text(3000, 21), streams.text(40000, 22), streams.text(500, 23)
    long_name = bytes(rnd.randrange(1, 256) for _ in range(150))      # no zero byte: two scan steps and a bit
    name63 = bytes(rnd.randrange(1, 256) for _ in range(63))         # the terminator is the last byte of a scan step
    variants = [
        streams.gz_member(a, name=b"a.txt", comment=b"hi", hcrc=True) + streams.gz_member(b, extra=b"XY\x02\x00zz") + streams.bgzf_member(c),
        streams.gz_member(a, name=long_name, comment=long_name[:70]) + streams.bgzf_member(c) + streams.gz_member(b, name=name63, hcrc=True),
        streams.gz_member(c, extra=bytes(rnd.getrandbits(8) for _ in range(300)), name=b"n") + streams.gz_member(a, extra=b"BC\x02\x00\xff\xff", comment=b"lying BC"),
        streams.bgzf_member(a) + streams.bgzf_member(c) + streams.gz_member(b, comment=name63 + b"x"),
    ]
    cases = []
    for g in variants:
        first = g.index(b"\x1f\x8b\x08", 4)   # where the second member starts
        for cut in list(range(0, min(first, 420))) + list(range(max(0, first - 12), first + 40)) + list(range(len(g) - 30, len(g) + 1)):
            cases.append(g[:cut])
        hdr_end = min(first, 400)
        for p in range(hdr_end):                      # every header byte of the first member: a bit flipped, a zero byte
            for v in (g[p] ^ (1 << (p % 8)), 0):
                m = bytearray(g); m[p] = v; cases.append(bytes(m))
        for p in range(first, first + 24):            # and the second member's
            m = bytearray(g); m[p] ^= 1 << rnd.randrange(8); cases.append(bytes(m))
    from archive_amd import errors
    seen = set()
    for i, buf in enumerate(cases):
        want = _noneify(orc.gzip_decode(buf))
        if want[0] == 3:  # the reference would loop for ever on this input: both say so
            want = (3, None)
        try:
            got = _gz(amd, buf)
        except errors.ReferenceWouldHang:
            got = (3, None)
        except Exception as e:  # keep the case index visible
            raise AssertionError("case %d: %r (oracle %r)" % (i, e, want[0]))
        assert got == want, (i, len(buf), got[0], want[0])
        seen.add(want[0])
    assert {0, 1, 2} <= seen, seen


def test_every_cut_and_every_header_bit(amd, orc):
    """Short streams of every block kind cut at EVERY byte, and zlib streams with every bit of the header and the trailer
    flipped: verdict, bytes and (for raw streams) the position the reference's reader stops at are the oracle's.  The
    random mutants of test_fuzz_gpu.py sample these; here the small cases are complete."""
    texts = [streams.text(700, 31), bytes(range(256)) * 2, b"ab" * 300 + streams.text(200, 32)]
    raws = []
    for t in texts:
        raws += [streams.raw_deflate(t), streams.raw_deflate(t, level=1), streams.raw_deflate(t, strategy=zlib.Z_FIXED), streams.raw_deflate(t, level=0)]
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raws.append(c.compress(texts[0]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(texts[2]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(texts[1]) + c.flush())
    n = 0
    for r in raws:
        for cut in range(len(r) + 1):
            buf = r[:cut]
            st, out, pos = orc.inflate_raw(buf)
            want = (2, None, None) if st == 2 else ((3, None, None) if st == 3 else (st, out, pos))
            assert _raw(amd, buf) == want, (len(r), cut, st)
            n += 1
    z = zlib.compress(texts[0], 6)
    zcases = [z[:cut] for cut in range(len(z) + 1)]
    for bit in list(range(16)) + list(range((len(z) - 4) * 8, len(z) * 8)):
        m = bytearray(z); m[bit >> 3] ^= 1 << (bit & 7); zcases.append(bytes(m))
    zd = zlib.compressobj(6, zlib.DEFLATED, 15, zdict=b"preset dictionary")
    zcases.append(zd.compress(texts[0]) + zd.flush())   # FDICT set
    for buf in zcases:
        for verify in (False, True):
            assert _zl(amd, buf, verify=verify) == _noneify(orc.zlib_decode(buf, verify=verify)), (len(buf), verify)
            n += 1
    assert n > 3000


def test_magics_at_every_edge_of_the_index_scan(amd, orc):
    """gz_count_candidates reads a 64 KiB tile as 16 rows of 256 x 16 bytes, screens every 16 bytes for the byte 1f, and
    takes the tile's interior path only when the whole tile and 20 bytes behind it exist (round 4).  `1f 8b 08` -- true
    member starts and false ones inside stored payloads -- are planted around every edge there is: the 16-byte, 4 096-byte
    and 64 KiB boundaries (the magic's three bytes on either side and across), the last tiles of the stream, and the byte
    1f alone where nothing follows.  The oracle decodes the same streams."""
    rnd = random.Random(17)

    def build(total, plants):
        # one stored-payload member whose bytes the test owns, so that a plant lands on a chosen offset of the STREAM
        pay = bytearray(b"x" * total)
        hdr = 10 + 5                       # gzip header + the first stored block's header (level 0: blocks of 65 535 bytes)
        for off in plants:
            rel = off - hdr - 5 * ((off - hdr) // 65540)   # (about: every 65 535 payload bytes cost 5 more)
            if 0 <= rel < total - 24:
                pay[rel:rel + 3] = b"\x1f\x8b\x08"
        return bytes(pay)
    for total, tail in ((200000, b""), (65536 * 2 - 40, b""), (131072 + 9, b"\x1f"), (70000, b"\x1f\x8b")):
        edges = []
        for base in (16, 4096, 65536, 131072, 16 * 777, 4096 * 9):
            edges += [base + d for d in (-4, -3, -2, -1, 0, 1, 13, 14, 15)]
        pay = build(total, edges)
        small = streams.text(300, 5)
        g = streams.gz_member(pay, level=0) + streams.bgzf_member(small) + streams.gz_member(small, name=b"n") + tail
        want = _noneify(orc.gzip_decode(g, cap=len(pay) + 2 * len(small) + 64))
        assert _gz(amd, g) == want, (total, len(tail))
        assert want[0] == 2 if tail else (want[0] == 0 and want[1][:len(pay)] == pay)  # (a lone 1f / 1f 8b behind the last member: the reference's RangeError)
        # ... and member starts themselves on the edges: short members in front shift the long one byte by byte
        for shift in range(0, 20):
            pre = streams.gz_member(b"s" * 3, level=0, name=b"n" * shift)
            g2 = pre + g
            assert _gz(amd, g2) == _noneify(orc.gzip_decode(g2, cap=len(pay) + 2 * len(small) + 80)), (total, shift)


def test_every_trailer_bit_of_a_gzip_stream(amd, orc):
    """CRC-32 and ISIZE of every member -- each of their 64 bits flipped in turn, in a stream of three members (plain, BGZF
    with its `BC` size hint, plain), with and without verification: the reference checks neither unless asked
    (_gzip_decoder_web.dart:27-58), a BGZF hint that no longer fits is caught by the index' check and measured instead."""
    a, b = streams.text(5000, 61), streams.text(900, 62)
    members = [streams.gz_member(a, name=b"x"), streams.bgzf_member(b), streams.gz_member(b + a)]
    g = b"".join(members)
    ends, o = [], 0
    for m in members:
        o += len(m); ends.append(o)
    n = 0
    for e in ends:
        for bit in range((e - 8) * 8, e * 8):
            buf = bytearray(g); buf[bit >> 3] ^= 1 << (bit & 7); buf = bytes(buf)
            for verify in (False, True):
                assert _gz(amd, buf, verify=verify) == _noneify(orc.gzip_decode(buf, verify=verify)), (e, bit, verify)
                n += 1
    # ... and the BSIZE field of the BGZF member's BC subfield
    p0 = len(members[0]) + 16
    for bit in range(p0 * 8, (p0 + 2) * 8):
        buf = bytearray(g); buf[bit >> 3] ^= 1 << (bit & 7); buf = bytes(buf)
        assert _gz(amd, buf) == _noneify(orc.gzip_decode(buf)), bit
    assert n == 384


def test_small_gzip_stream_cut_everywhere_and_flipped_everywhere(amd, orc):
    """A three-member gzip stream of 0.6 KB (a name, a BGZF member, a header CRC; dynamic, fixed and stored blocks): cut at
    every byte, and every third bit flipped -- header, payload, trailer alike.  Verdict and bytes are the oracle's, with and
    without verification."""
    from archive_amd import errors
    a, b, c = streams.text(900, 71), streams.text(300, 72), bytes(range(200))
    g = streams.gz_member(a, name=b"first") + streams.bgzf_member(b, level=1) + streams.gz_member(c, level=0, hcrc=True)
    cases = [g[:cut] for cut in range(len(g) + 1)]
    for bit in range(0, len(g) * 8, 3):
        m = bytearray(g); m[bit >> 3] ^= 1 << (bit & 7); cases.append(bytes(m))
    seen = set()
    for i, buf in enumerate(cases):
        for verify in (False, True):
            want = _noneify(orc.gzip_decode(buf, verify=verify))
            if want[0] == 3:
                want = (3, None)
            try:
                got = _gz(amd, buf, verify=verify)
            except errors.ReferenceWouldHang:
                got = (3, None)
            assert got == want, (i, len(buf), verify, got[0], want[0])
            seen.add(want[0])
    assert {0, 1, 2} <= seen, seen
