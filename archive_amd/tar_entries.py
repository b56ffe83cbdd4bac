"""The tar.gz / tar.bz2 helpers: the compressed stream through the GPU decoders, then the tar records.

ref (relative to /root/reference/lib/src): io/extract_archive_to_disk.dart:180-197 and io/tar_command.dart:20-48 --
`GZipDecoder().decodeStream(input, output)` / `BZip2Decoder().decodeStream(input, output)` into a temporary tar, then
`TarDecoder().decodeStream` (codecs/tar_decoder.dart:24-127, codecs/tar/tar_file.dart:76-124,218-244).  The
decompression is the hot path and goes through `GZipDecoder` / `BZip2Decoder` of this package (the HIP library); the
record walk below is the host side of `TarDecoder`, restated field by field so that entries come out as the
reference lists them -- GNU `././@LongLink` names, the pax `path` / `linkpath` records, global pax headers skipped,
the reference's quirks included (a numeric field that does not parse is 0; only files are padded to 512 bytes).

No files are written here: what to do with the entries is the caller's business (`extractArchiveToDisk` is I/O).
"""
import re

from .codecs import BZip2Decoder, GZipDecoder, _as_buffer

# what Dart's String.trim() removes (the reference trims every header string, tar_file.dart:233-238)
_DART_WS = "".join(map(chr, list(range(0x09, 0x0E)) + [0x20, 0x85, 0xA0, 0x1680] + list(range(0x2000, 0x200B)) +
                       [0x2028, 0x2029, 0x202F, 0x205F, 0x3000, 0xFEFF]))
_PAX_RECORD = re.compile(r"(\d+) (\w+)=([^\n\r\u2028\u2029]*)", re.ASCII)  # tar_decoder.dart:9 (Dart's `.` stops at \n, \r, U+2028, U+2029)

NORMAL_FILE, HARD_LINK, SYMBOLIC_LINK, DIRECTORY = "0", "1", "2", "5"


class TarEntry:
    """One `TarFile` (tar_file.dart:36-75) -- the fields `TarDecoder` copies into its `ArchiveFile`."""
    __slots__ = ("name", "mode", "owner_id", "group_id", "size", "last_mod_time", "checksum", "type_flag", "link_name",
                 "ustar", "owner_user_name", "owner_group_name", "content")

    @property
    def is_file(self):  # tar_file.dart:126
        return self.type_flag != DIRECTORY

    @property
    def is_symlink(self):
        return self.type_flag == SYMBOLIC_LINK

    def __repr__(self):  # TarFile.toString
        return "[%s, %d, %d]" % (self.name, self.mode, self.size)


def _string(field, decode_utf8=True):
    """`_parseString` (tar_file.dart:230-244): up to the first NUL, UTF-8 (Latin-1 code units when that fails), trimmed."""
    r = field.find(b"\0")
    s = field if r < 0 else field[:r]
    try:
        text = s.decode("utf-8") if decode_utf8 else s.decode("latin-1")
    except UnicodeDecodeError:
        text = s.decode("latin-1")
    return text.strip(_DART_WS)


_OCTAL = re.compile(r"[+-]?[0-7]+\Z")


def _int(field):
    """`_parseInt` (tar_file.dart:218-228): octal; anything `int.parse(s, radix: 8)` rejects is 0."""
    s = _string(field)
    return int(s, 8) if _OCTAL.match(s) else 0


def _read_entry(buf, pos, store_data):
    """`TarFile.read` (tar_file.dart:76-124) at `pos` -> (entry, position behind it).  A header cut short by the end of
    the input reads as what is there (the reference's readBytes returns the rest), fields behind it empty."""
    n = len(buf)
    h = bytes(buf[pos:pos + 512])
    pos = min(n, pos + 512)
    e = TarEntry()
    e.name = _string(h[0:100])
    e.mode = _int(h[100:108])
    e.owner_id = _int(h[108:116])
    e.group_id = _int(h[116:124])
    e.size = _int(h[124:136])
    e.last_mod_time = _int(h[136:148])
    e.checksum = _int(h[148:156])
    e.type_flag = _string(h[156:157])
    e.link_name = _string(h[157:257])
    e.ustar = _string(h[257:263]) == "ustar"
    e.owner_user_name = e.owner_group_name = ""
    if e.ustar:
        e.owner_user_name = _string(h[265:297])
        e.owner_group_name = _string(h[297:329])
        prefix = _string(h[345:500])
        if prefix:
            e.name = prefix + "/" + e.name
    size = max(0, e.size)
    end = min(n, pos + size)
    e.content = bytes(buf[pos:end]) if (store_data or e.name == "././@LongLink") else None
    pos = end
    if e.is_file and e.size > 0 and e.size % 512:
        pos = min(n, pos + 512 - e.size % 512)
    return e, pos


def read_tar(data, store_data=True):
    """`TarDecoder().decodeBytes(data, storeData:)` (tar_decoder.dart:24-127) -> the entries it adds to its archive."""
    buf, n = _as_buffer(data)
    entries = []
    next_name = next_link = None
    pos = 0
    while pos < n:
        if n - pos < 2 or (buf[pos] == 0 and buf[pos + 1] == 0):  # two zero bytes: the end of the archive
            break
        e, pos = _read_entry(buf, pos, store_data)
        if e.name == "././@LongLink":  # GNU tar: the next entry's name, as a file
            c = e.content
            z = c.find(b"\0")
            c = c if z < 0 else c[:z]
            try:
                next_name = c.decode("utf-8")
            except UnicodeDecodeError:
                next_name = c.decode("latin-1")
            continue
        if e.type_flag in ("g", "G"):  # global pax header: skipped
            continue
        if e.type_flag in ("x", "X"):  # pax records for the next entry: path and linkpath are honoured
            if e.content is None:  # (storeData: false leaves the reference without the records too: it throws there)
                raise ValueError("pax header without its data (store_data=False)")
            for record in e.content.decode("utf-8").split("\n"):
                m = _PAX_RECORD.search(record)
                if not m:
                    continue
                if m.group(2) == "path":
                    next_name = m.group(3)
                elif m.group(2) == "linkpath":
                    next_link = m.group(3)
            continue
        if next_name is not None:
            e.name, next_name = next_name, None
        if next_link is not None:
            e.link_name, next_link = next_link, None
        entries.append(e)
    return entries


def gunzip_tar(data, store_data=True):
    """.tar.gz / .tgz: `GZipDecoder().decodeStream` then `TarDecoder` (extract_archive_to_disk.dart:180-188,208-211)."""
    return read_tar(GZipDecoder().decode_bytes(data), store_data)


def bunzip2_tar(data, store_data=True):
    """.tar.bz2 / .tbz: `BZip2Decoder().decodeStream` then `TarDecoder` (extract_archive_to_disk.dart:189-197)."""
    return read_tar(BZip2Decoder().decode_bytes(data), store_data)


def read_archive(name, data, store_data=True):
    """The reference's dispatch on the file name (`getInputExtension`, extract_archive_to_disk.dart:146-158,180-211)."""
    low = name.lower()
    if low.endswith((".tar.gz", ".tgz")):
        return gunzip_tar(data, store_data)
    if low.endswith((".tar.bz2", ".tbz")):
        return bunzip2_tar(data, store_data)
    if low.endswith(".tar"):
        return read_tar(data, store_data)
    raise ValueError("%r: must end with .tar.gz, .tgz, .tar.bz2, .tbz or .tar here" % name)
