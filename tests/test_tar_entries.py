"""The tar.gz / tar.bz2 helpers (archive_amd/tar_entries.py; ref io/extract_archive_to_disk.dart:180-211,
io/tar_command.dart:20-48, codecs/tar_decoder.dart): the record walk on the CPU against the header fields the reference's
own tar tests pin (tests/golden/tar/, from test/tar_test.dart) and against CPython's tarfile; the decompression in front
of it on the GPU."""
import hashlib
import io
import json
import os
import random
import tarfile

import pytest

from tests import streams

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "tar", "cases.json")))


def _rd(*p):
    with open(os.path.join(GOLD, *p), "rb") as f:
        return f.read()


@pytest.mark.parametrize("case", [c for c in CASES if "headers" in c], ids=lambda c: c["file"])
def test_reference_header_tables(case):
    """test/tar_test.dart:297-343: every field the reference's test compares, for every entry of its five archives."""
    from archive_amd import tar_entries
    got = tar_entries.read_tar(_rd("tar", case["file"]))
    assert len(got) == len(case["headers"])
    field = dict(Name="name", Mode="mode", Uid="owner_id", Gid="group_id", Size="size", Linkname="link_name", ModTime="last_mod_time",
                 Typeflag="type_flag", Uname="owner_user_name", Gname="owner_group_name")
    for e, hdr in zip(got, case["headers"]):
        for k, attr in field.items():
            if k in hdr:
                assert getattr(e, attr) == hdr[k], (case["file"], k)
        assert len(e.content) == e.size


def test_reference_gnu_checksums():
    """the `cksums` of gnu.tar's two files (test/tar_test.dart:37-40) are the MD5 of their contents"""
    from archive_amd import tar_entries
    got = tar_entries.read_tar(_rd("tar", "gnu.tar"))
    assert [hashlib.md5(e.content).hexdigest() for e in got] == ["e38b27eaccb4391bdec553a7f3ae6b2f", "c65bd2e50a56a2138bf1716f2fd56fe9"]


def test_reference_symlink_and_test2():
    from archive_amd import tar_entries
    c = [c for c in CASES if c["file"] == "symlink_tar.tar"][0]
    got = tar_entries.read_tar(_rd("tar", c["file"]))
    assert len(got) == c["length"]                                   # test/tar_test.dart:222-229
    assert got[c["symlink_at"]].is_symlink and got[c["symlink_at"]].link_name == c["symlink"]
    got = tar_entries.read_tar(_rd("test2_tar_gz.out"))               # test/tar_test.dart:231-253: four entries
    assert len(got) == 4
    want = tarfile.open(fileobj=io.BytesIO(_rd("test2_tar_gz.out")))
    assert [(e.name.rstrip("/"), e.size, e.is_file) for e in got] == [(m.name.rstrip("/"), m.size, not m.isdir()) for m in want.getmembers()]
    assert [e.content for e in got if e.is_file] == [want.extractfile(m).read() for m in want.getmembers() if m.isreg()]


def test_invalid_archive_is_not_an_exception_here():
    """`TarDecoder().decodeBytes([1, 2, 3])` (test/tar_test.dart:151-158: the test passes whether or not it throws): three
    bytes are one header cut short -- a nameless, empty entry"""
    from archive_amd import tar_entries
    got = tar_entries.read_tar(bytes([1, 2, 3]))
    assert len(got) == 1 and got[0].size == 0 and got[0].content == b""
    assert tar_entries.read_tar(b"") == [] and tar_entries.read_tar(b"\0" * 1024) == []


def _make_tar(fmt, rnd):
    files = [("a.txt", streams.text(700, 1)), ("dir/b.bin", rnd.randbytes(1536)), ("empty", b""),
             ("d/" + "x" * 120 + "/long-name.txt", streams.text(3000, 2)), ("ünïcode.txt", b"y" * 513)]
    b = io.BytesIO()
    with tarfile.open(fileobj=b, mode="w", format=fmt) as t:
        d = tarfile.TarInfo("dir")
        d.type = tarfile.DIRTYPE
        d.mode = 0o755
        t.addfile(d)
        for name, data in files:
            ti = tarfile.TarInfo(name)
            ti.size, ti.mode, ti.mtime, ti.uid, ti.gid, ti.uname, ti.gname = len(data), 0o640, 1244428340, 73025, 5000, "u", "g"
            t.addfile(ti, io.BytesIO(data))
        ln = tarfile.TarInfo("link")
        ln.type, ln.linkname = tarfile.SYMTYPE, "t/" + "y" * 130
        t.addfile(ln)
    return b.getvalue(), files


@pytest.mark.parametrize("fmt", [tarfile.USTAR_FORMAT, tarfile.GNU_FORMAT, tarfile.PAX_FORMAT], ids=["ustar", "gnu", "pax"])
def test_against_tarfile(fmt):
    """names (prefix field, GNU long names, pax path / linkpath records), sizes, modes and contents as CPython's tarfile
    reads them back"""
    from archive_amd import tar_entries
    if fmt == tarfile.USTAR_FORMAT:
        b = io.BytesIO()
        with tarfile.open(fileobj=b, mode="w", format=fmt) as t:
            for name, data in [("p" * 90 + "/" + "q" * 60, b"abc"), ("short", b"x" * 1000)]:
                ti = tarfile.TarInfo(name)
                ti.size = len(data)
                t.addfile(ti, io.BytesIO(data))
        raw = b.getvalue()
    else:
        raw, _ = _make_tar(fmt, random.Random(5))
    got = tar_entries.read_tar(raw)
    want = tarfile.open(fileobj=io.BytesIO(raw))
    members = want.getmembers()
    # (a GNU long LINK name -- type K -- is a `././@LongLink` entry too, and the reference takes every such entry for the next
    #  entry's NAME, tar_decoder.dart:42-45: the symbolic link comes out named like its target, its link name cut at 100 bytes)
    k_quirk = fmt == tarfile.GNU_FORMAT
    assert [e.name.rstrip("/") for e in got] == [(m.linkname if k_quirk and m.issym() else m.name).rstrip("/") for m in members]
    for e, m in zip(got, members):
        assert (e.size, e.mode, e.last_mod_time, e.is_file) == (m.size, m.mode, m.mtime, not m.isdir())
        if m.issym():
            assert e.is_symlink and e.link_name == (m.linkname[:100] if k_quirk else m.linkname)
        if m.isreg():
            assert e.content == want.extractfile(m).read()
    listed = tar_entries.read_tar(raw, store_data=False) if fmt != tarfile.PAX_FORMAT else None  # `listTarFiles`: storeData false
    if listed is not None:
        assert [e.name for e in listed] == [e.name for e in got] and all(e.content is None for e in listed)


def test_reference_quirks():
    """a numeric field that does not parse is 0 (tar_file.dart:218-228); only FILES are padded to 512 bytes (:115-122)"""
    from archive_amd import tar_entries
    raw, _ = _make_tar(tarfile.GNU_FORMAT, random.Random(6))
    h = bytearray(raw)
    h[512 + 108:512 + 116] = b"\x80\0\0\0\0\0\1\2"   # the first file's uid in base 256 (GNU extension): the reference reads 0
    h[512 + 124:512 + 136] = b"0000000128 \0"         # ... and a size with an 8 in it (not octal): 0 -- its data then reads as headers
    got = tar_entries.read_tar(bytes(h))
    assert got[1].owner_id == 0 and got[1].size == 0 and got[1].content == b""
    d = bytearray(512)                                 # a DIRECTORY with a size: its data is read, the padding is not skipped
    d[0:2] = b"d/"
    d[124:136] = b"00000000005\0"
    d[156] = ord("5")
    f = bytearray(512)
    f[0:1] = b"f"
    f[124:136] = b"00000000000\0"
    f[156] = ord("0")
    got = tar_entries.read_tar(bytes(d) + b"hello" + bytes(f) + bytes(1024))
    assert [(e.name, e.size, e.is_file) for e in got] == [("d/", 5, False), ("f", 0, True)] and got[0].content == b"hello"


def test_dispatch_on_the_name():
    from archive_amd import tar_entries
    raw = _rd("test2_tar_gz.out")
    assert len(tar_entries.read_archive("X.TAR", raw)) == 4
    with pytest.raises(ValueError):
        tar_entries.read_archive("x.tar.xz", raw)


@pytest.mark.gpu
def test_gpu_tar_gz_and_tar_bz2():
    """test/tar_test.dart:242-253 and test/io_test.dart:364,636: the reference's two compressed fixtures, and a larger
    archive made here, through the GPU decoders and the record walk"""
    import bz2
    import gzip
    from archive_amd import tar_entries
    plain = tar_entries.read_tar(_rd("test2_tar_gz.out"))
    for got in (tar_entries.gunzip_tar(_rd("test2_tar_gz.in")), tar_entries.bunzip2_tar(_rd("test2_tar_bz2.in")),
                tar_entries.read_archive("test2.tgz", _rd("test2_tar_gz.in")), tar_entries.read_archive("a/b/T.TAR.BZ2", _rd("test2_tar_bz2.in"))):
        assert len(got) == 4
        assert [(e.name, e.size, e.type_flag, e.content) for e in got] == [(e.name, e.size, e.type_flag, e.content) for e in plain]
    rnd = random.Random(11)
    b = io.BytesIO()
    files = []
    with tarfile.open(fileobj=b, mode="w", format=tarfile.GNU_FORMAT) as t:
        for i in range(40):
            data = streams.text(rnd.randrange(0, 200000), i) if i % 3 else rnd.randbytes(rnd.randrange(0, 50000))
            name = "dir%d/" % (i % 4) + ("n" * (20 + 7 * i)) + ".dat"
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            t.addfile(ti, io.BytesIO(data))
            files.append((name, data))
    raw = b.getvalue()
    for got in (tar_entries.gunzip_tar(gzip.compress(raw, 6)), tar_entries.bunzip2_tar(bz2.compress(raw, 9))):
        assert [(e.name, e.content) for e in got] == files
