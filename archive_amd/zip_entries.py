"""The ZIP entry path on the GPU: every deflate-compressed entry of an archive in ONE batch call.

ref (relative to /root/reference/lib/src/codecs): zip/zip_directory.dart (end-of-central-directory and central
directory records), zip/zip_file_header.dart, zip/zip_file.dart:182-248 -- the reference decompresses entry by
entry with `ZLibDecoder().decodeBytes(compressed, raw: true)`; entries are independent raw DEFLATE streams, i.e.
the same member-parallel shape as multi-member gzip, so they go through `ahip_inflate_batch` together.

Only what that path needs is parsed here (names, method, sizes, data offsets).  ZIP64, encryption and
multi-disk archives are the reference's own (CPU) business and raise `ValueError`.
"""
import ctypes
import struct

from . import _native as N
from .codecs import BZip2Decoder, _as_buffer
from .errors import ArchiveHipError

_EOCD = 0x06054B50
_CDH = 0x02014B50
_LFH = 0x04034B50
STORE, DEFLATE, BZIP2 = 0, 8, 12


class ZipEntry:
    __slots__ = ("name", "method", "flags", "crc32", "compressed_size", "uncompressed_size", "data_offset")

    def __repr__(self):
        return "ZipEntry(%r, method=%d, %d -> %d)" % (self.name, self.method, self.compressed_size, self.uncompressed_size)


def read_directory(data):
    """Central directory -> list of ZipEntry (archive order)."""
    buf, n = _as_buffer(data)
    lo = max(0, n - 65557)
    pos = buf.rfind(struct.pack("<I", _EOCD), lo)
    if pos < 0:
        raise ValueError("no end-of-central-directory record")
    (_, disk, cd_disk, n_here, n_total, cd_size, cd_off, _clen) = struct.unpack_from("<IHHHHIIH", buf, pos)
    if disk != 0 or cd_disk != 0 or n_here != n_total:
        raise ValueError("multi-disk archives are not handled here")
    if n_total == 0xFFFF or cd_off == 0xFFFFFFFF or cd_size == 0xFFFFFFFF:
        raise ValueError("ZIP64 archives are not handled here")
    entries = []
    p = cd_off
    for _ in range(n_total):
        (sig, _vm, _vn, flags, method, _t, _d, crc, csize, usize, nlen, xlen, clen, _dn, _ia, _ea, lho) = \
            struct.unpack_from("<IHHHHHHIIIHHHHHII", buf, p)
        if sig != _CDH:
            raise ValueError("bad central directory record")
        e = ZipEntry()
        e.name = buf[p + 46:p + 46 + nlen].decode("utf-8" if flags & 0x800 else "cp437")
        e.method, e.flags, e.crc32, e.compressed_size, e.uncompressed_size = method, flags, crc, csize, usize
        if csize == 0xFFFFFFFF or usize == 0xFFFFFFFF or lho == 0xFFFFFFFF:
            raise ValueError("ZIP64 entries are not handled here")
        if flags & 1:
            raise ValueError("encrypted entries are not handled here")
        (lsig, _v, _f, _m, _lt, _ld, _c, _cs, _us, lnlen, lxlen) = struct.unpack_from("<IHHHHHIIIHH", buf, lho)
        if lsig != _LFH:
            raise ValueError("bad local file header")
        e.data_offset = lho + 30 + lnlen + lxlen
        entries.append(e)
        p += 46 + nlen + xlen + clen
    return entries


def inflate_entries(data, entries, trust_sizes=True):
    """Decompresses `entries` (DEFLATE ones in one GPU batch) -> list of bytes, archive order.

    trust_sizes: use the directory's uncompressed sizes as output windows (one pass); an entry whose stream
    produces more than the directory says is redone alone.  False measures every stream first."""
    buf, n = _as_buffer(data)
    out = [None] * len(entries)
    idx = [i for i, e in enumerate(entries) if e.method == DEFLATE]
    for i, e in enumerate(entries):
        if e.method == STORE:
            out[i] = buf[e.data_offset:e.data_offset + e.compressed_size]
        elif e.method == BZIP2:
            out[i] = BZip2Decoder().decode_bytes(buf[e.data_offset:e.data_offset + e.compressed_size])
        elif e.method != DEFLATE:
            raise ValueError("compression method %d is not on this path" % e.method)
    if idx:
        k = len(idx)
        u64s = ctypes.c_uint64 * k
        in_off = u64s(*[entries[i].data_offset for i in idx])
        in_size = u64s(*[entries[i].compressed_size for i in idx])
        hint = u64s(*[entries[i].uncompressed_size for i in idx]) if trust_sizes else None
        out_off, out_len, status = u64s(), u64s(), (ctypes.c_int32 * k)()
        cap = sum(entries[i].uncompressed_size for i in idx) if trust_sizes else 0
        total = ctypes.c_size_t(0)
        for _ in range(2):
            obuf = ctypes.create_string_buffer(max(1, cap))
            rc = N.lib().ahip_inflate_batch(buf, n, k, in_off, in_size, hint, obuf, cap, out_off, out_len, status,
                                            ctypes.byref(total))
            if rc != N.AHIP_E_CAP:
                break
            cap = total.value
        if rc != N.AHIP_OK:
            raise ArchiveHipError(rc, N.last_error())
        raw = obuf.raw
        from .codecs import Inflate
        for j, i in enumerate(idx):
            if status[j] == N.AHIP_E_CAP:  # the directory understated this entry: alone, with its own sizing
                e = entries[i]
                out[i] = Inflate(buf[e.data_offset:e.data_offset + e.compressed_size]).get_bytes()
            elif status[j] in (N.AHIP_OK, N.AHIP_FALSE):  # FALSE: the reference keeps the bytes it got, silently
                out[i] = raw[out_off[j]:out_off[j] + out_len[j]]
            else:
                raise ArchiveHipError(status[j], "entry %r" % entries[i].name)
    return out


def read_zip(data, trust_sizes=True):
    """[(name, bytes)] of every entry, like iterating `ZipDecoder().decodeBytes(data)` and reading `.content`."""
    entries = read_directory(data)
    return list(zip([e.name for e in entries], inflate_entries(data, entries, trust_sizes)))
