#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference's own test fixtures.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py

Each vector is the (input, expected output) pair a reference test pins for the Inflate hot
path.  Expected outputs come from the reference's own expected files /
literals where they exist (cat.jpg, test2.tar, aTxt, gitExpectedOutput) and are cross-checked
here against C zlib; for inflate/data.bin the reference only pins the decoded length (5259),
so the expected bytes are zlib's (unique for a valid stream) and the length is asserted.
"""
import gzip
import hashlib
import json
import os
import re
import zlib

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def rd(rel):
    with open(os.path.join(REF, rel), "rb") as f:
        return f.read()


def wr(name, data):
    with open(os.path.join(HERE, name), "wb") as f:
        f.write(data)


def dart_int_list(src, name):
    m = re.search(r"final %s = Uint8List\.fromList\(<int>\[(.*?)\]\);" % name, src, re.S)
    return bytes(int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1)))


def main():
    manifest = []

    def add(name, kind, inp, exp, cite, **kw):
        wr(name + ".in", inp)
        wr(name + ".out", exp)
        manifest.append(dict(name=name, kind=kind, in_len=len(inp), out_len=len(exp),
                             sha256=hashlib.sha256(exp).hexdigest(), cite=cite, **kw))

    # test/inflate_test.dart:14-20  Inflate(data.bin) -> 5259 chars
    d = rd("test/_data/inflate/data.bin")
    exp = zlib.decompress(d, -15)
    assert len(exp.decode("utf-8")) == 5259
    add("inflate_data_bin", "raw", d, exp, "test/inflate_test.dart:14-20")

    # test/inflate_test.dart:57-179 (disabled test) gitInflateInput -> gitExpectedOutput (first zlib stream)
    src = rd("test/inflate_test.dart").decode()
    gin, gexp = dart_int_list(src, "gitInflateInput"), dart_int_list(src, "gitExpectedOutput")
    dz = zlib.decompressobj()
    assert dz.decompress(gin) == gexp and len(gin) - len(dz.unused_data) == 148
    add("git_zlib_first_member", "zlib_first", gin[:148], gexp, "test/inflate_test.dart:57-179", consumed=148)

    # test/gzip_test.dart:63-93
    aTxt = re.search(r"const aTxt = '''(.*?)''';", rd("test/_test_util.dart").decode(), re.S).group(1).replace('\r\n', '\n').encode()  # the .dart file is checked out with CRLF
    for fn, exp, cite in (("a.txt.gz", aTxt, "test/gzip_test.dart:85-93"),
                          ("cat.jpg.gz", rd("test/_data/cat.jpg"), "test/gzip_test.dart:63-72"),
                          ("test2.tar.gz", rd("test/_data/test2.tar"), "test/gzip_test.dart:74-83")):
        d = rd("test/_data/" + fn)
        assert gzip.decompress(d) == exp, fn
        add(fn.replace(".", "_"), "gzip", d, exp, cite)

    # multi-member: test/gzip_test.dart:44-52, test/zlib_test.dart:15-23 (inputs are built by the
    # tests with the encoder; any valid encoding pins the same decoded bytes)
    g = gzip.compress(bytes([1, 2, 3]), mtime=0) + gzip.compress(bytes([4, 5, 6]), mtime=0)
    add("gzip_multi_member", "gzip", g, bytes([1, 2, 3, 4, 5, 6]), "test/gzip_test.dart:44-52")
    z = zlib.compress(bytes([1, 2, 3])) + zlib.compress(bytes([4, 5, 6]))
    add("zlib_multi_member", "zlib_verify", z, bytes([1, 2, 3, 4, 5, 6]), "test/zlib_test.dart:15-23")

    # deflate-compressed zip entry with known content: test/zip_test.dart:11-28 (test.zip, a.txt)
    # -- raw deflate payload located through the local file header
    zp = rd("test/_data/zip/test.zip") if os.path.exists(os.path.join(REF, "test/_data/zip/test.zip")) else None
    if zp is not None:
        import io
        import zipfile
        zf = zipfile.ZipFile(io.BytesIO(zp))
        for info in zf.infolist():
            if info.compress_type == 8 and info.file_size > 0:
                off = info.header_offset
                nlen = int.from_bytes(zp[off + 26:off + 28], "little")
                xlen = int.from_bytes(zp[off + 28:off + 30], "little")
                payload = zp[off + 30 + nlen + xlen: off + 30 + nlen + xlen + info.compress_size]
                add("zip_entry_" + re.sub(r"\W", "_", info.filename), "raw", payload, zf.read(info),
                    "test/zip_test.dart:11-28 (test/_data/zip/test.zip)")
                break

    # bzip2: test/bzip2_test.dart:8-15 (test.bz2 only asserted to decode) and test2.tar.bz2 -> test2.tar
    # (test/io_test.dart:364,636)
    import bz2
    d = rd("test/_data/bzip2/test.bz2")
    add("bzip2_test_bz2", "bzip2", d, bz2.decompress(d), "test/bzip2_test.dart:8-15")
    d = rd("test/_data/test2.tar.bz2")
    assert bz2.decompress(d) == rd("test/_data/test2.tar")
    add("test2_tar_bz2", "bzip2", d, rd("test/_data/test2.tar"), "test/io_test.dart:364")

    # checksum known-answer tests: test/crc32_test.dart:5-25, test/adler32_test.dart:5-25
    kat = {
        "crc32": [["01", "A505DF1B"], ["01020304050607080900", "C5F5BE65"], ["01020304050607080900*10000", "3AC67C2B"]],
        "adler32": [["01", "00020002"], ["01020304050607080900", "00DC002E"], ["01020304050607080900*10000", "96C8DE2B"]],
    }
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(dict(vectors=manifest, checksum_kat=kat), f, indent=1)
    print("wrote", len(manifest), "vectors")


if __name__ == "__main__":
    main()
