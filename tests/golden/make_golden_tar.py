#!/usr/bin/env python3
"""Regenerates tests/golden/tar/ from the reference's own tar tests (build container only: needs /root/reference).

    python tests/golden/make_golden_tar.py

`test/tar_test.dart:10-146` pins the header fields `TarDecoder` reads from five small archives (gnu, star, v7, pax,
nil-uid: the `tarTests` table, compared field by field at :297-343), `:222-229` the symbolic link of `symlink_tar.tar`
and `:231-253` the four entries of `test2.tar` (which is what `test2.tar.gz` / `test2.tar.bz2` decompress to:
tests/golden/test2_tar_gz.out).  The table is taken from the Dart source as it stands -- a map literal that is also a
Python literal once `int.parse('0640', radix: 8)` and the `TarFile.*` constants are spelled out -- and the archives are
copied beside it.
"""
import ast
import json
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tar")


def main():
    os.makedirs(HERE, exist_ok=True)
    src = open(os.path.join(REF, "test/tar_test.dart")).read()
    m = re.search(r"var tarTests = (\[.*?\n\]);", src, re.S)
    lit = m.group(1)
    lit = re.sub(r"int\.parse\('0?([0-7]+)', radix: 8\)", r"0o\1", lit)
    lit = lit.replace("TarFile.normalFile", "'0'").replace("TarFile.symbolicLink", "'2'")
    cases = ast.literal_eval(lit)
    out = []
    for c in cases:
        name = os.path.basename(c["file"])
        shutil.copyfile(os.path.join(REF, "test", c["file"]), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)
        out.append(dict(file=name, headers=c["headers"], cite="test/tar_test.dart:10-146,297-343"))
    shutil.copyfile(os.path.join(REF, "test/_data/tar/symlink_tar.tar"), os.path.join(HERE, "symlink_tar.tar"))
    os.chmod(os.path.join(HERE, "symlink_tar.tar"), 0o644)
    out.append(dict(file="symlink_tar.tar", length=4, symlink_at=1, symlink="b/b.txt", cite="test/tar_test.dart:222-229"))
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("%d cases -> %s" % (len(out), HERE))


if __name__ == "__main__":
    main()
