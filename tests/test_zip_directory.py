"""Host logic of the ZIP entry path (no GPU): the central-directory reader of archive_amd/zip_entries.py against
CPython's zipfile."""
import io
import random
import zipfile

import pytest

from tests import streams


def _zip(files):
    b = io.BytesIO()
    with zipfile.ZipFile(b, "w") as z:
        for name, data, method in files:
            z.writestr(zipfile.ZipInfo(name), data, compress_type=method)
    return b.getvalue()


def test_directory_matches_zipfile():
    from archive_amd import zip_entries
    rnd = random.Random(3)
    files = [("a.txt", streams.text(5000, 1), zipfile.ZIP_DEFLATED), ("dir/b.bin", rnd.randbytes(3000), zipfile.ZIP_STORED),
             ("empty", b"", zipfile.ZIP_DEFLATED), ("c.bz2", streams.text(9000, 2), zipfile.ZIP_BZIP2),
             ("ünïcode.txt", b"x" * 100, zipfile.ZIP_DEFLATED)]
    z = _zip(files)
    got = zip_entries.read_directory(z)
    want = zipfile.ZipFile(io.BytesIO(z)).infolist()
    assert [e.name for e in got] == [i.filename for i in want]
    for e, i in zip(got, want):
        assert (e.method, e.crc32, e.compressed_size, e.uncompressed_size) == (i.compress_type, i.CRC, i.compress_size, i.file_size)
        # the data really starts there: stored entries can be compared directly, deflate ones via zlib
        raw = z[e.data_offset:e.data_offset + e.compressed_size]
        if e.method == 0:
            assert raw == dict((f[0], f[1]) for f in files)[e.name]
        elif e.method == 8:
            import zlib
            assert zlib.decompress(raw, -15) == dict((f[0], f[1]) for f in files)[e.name]


def test_rejects_what_it_does_not_handle():
    from archive_amd import zip_entries
    with pytest.raises(ValueError):
        zip_entries.read_directory(b"not a zip at all")
    z = bytearray(_zip([("a", b"abc", zipfile.ZIP_STORED)]))
    cd = z.rfind(b"PK\x01\x02")
    z[cd + 8] |= 1  # general purpose flag bit 0: encrypted
    with pytest.raises(ValueError):
        zip_entries.read_directory(bytes(z))
