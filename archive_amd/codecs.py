"""Host-side mirror of the reference's codec classes for the Inflate hot path.

Same names, argument meaning and (silent) error behaviour as the Dart classes; the work is done
by libarchive_hip.so on the GPU through the C-ABI (include/archive_hip.h).  camelCase aliases
(`decodeBytes`, `getBytes`) keep reference call sites readable side by side.

ref (relative to /root/reference/lib/src):
  codecs/zlib/inflate.dart:12-116            Inflate
  codecs/zlib_decoder.dart:14-35             ZLibDecoder   (+ zlib/zlib_decoder_web.dart:8 ZLibDecoderWeb)
  codecs/gzip_decoder.dart:14-30             GZipDecoder   (+ zlib/gzip_decoder_web.dart:8 GZipDecoderWeb)
  util/crc32.dart:6, util/adler32.dart:29    getCrc32 / getAdler32
"""
import ctypes

from . import _native as N
from .errors import ArchiveHipError, RangeError, ReferenceWouldHang


def _as_buffer(data):
    if isinstance(data, (bytes, bytearray)):
        b = bytes(data)
    elif isinstance(data, memoryview):
        b = data.tobytes()
    else:
        b = bytes(bytearray(data))  # List<int> semantics: values are truncated to bytes by Uint8List
    return b, len(b)


def _check(rc):
    """Maps C-ABI status to the reference's behaviour: 0/1 -> bytes (silent), 2 -> RangeError."""
    if rc in (N.AHIP_OK, N.AHIP_FALSE):
        return rc
    if rc == N.AHIP_RANGE:
        raise RangeError("reference would throw RangeError on this input")
    if rc == N.AHIP_HANG:
        raise ReferenceWouldHang("reference decoder does not terminate on this input")
    raise ArchiveHipError(rc, N.last_error())


def _call_growing(fn, n_in, hint=None):
    """Runs fn(out_ptr, cap, out_len_ref) growing the buffer on AHIP_E_CAP."""
    cap = max(64, hint if hint is not None else 4 * n_in + 64)
    for _ in range(3):
        out = ctypes.create_string_buffer(cap)
        olen = ctypes.c_size_t(0)
        rc = fn(out, cap, ctypes.byref(olen))
        if rc == N.AHIP_E_CAP:
            cap = olen.value + 64
            continue
        return _check(rc), out.raw[:olen.value]
    raise ArchiveHipError(N.AHIP_E_CAP, "output size did not settle")


class Inflate:
    """`Inflate(bytes).getBytes()` -- raw DEFLATE decode.

    Whole-buffer only: the streaming constructor `Inflate.stream(null)` + `addBytes` of the
    reference (inflate.dart:36-99) stays on the reference's own Dart path (scope)."""

    def __init__(self, data, output=None, uncompressed_size=None):
        buf, n = _as_buffer(data)
        consumed = ctypes.c_size_t(0)

        def run(out, cap, olen):
            return N.lib().ahip_inflate_raw(buf, n, out, cap, olen, ctypes.byref(consumed))

        self.status, self._bytes = _call_growing(run, n, uncompressed_size)
        self.input_position = consumed.value
        if output is not None:
            output.extend(self._bytes)  # any bytearray-like stands in for OutputStream

    def get_bytes(self):
        return self._bytes

    getBytes = get_bytes


class ZLibDecoder:
    """`ZLibDecoder().decodeBytes(data, verify: false, raw: false)`"""

    def decode_bytes(self, data, verify=False, raw=False):
        buf, n = _as_buffer(data)
        st, out = _call_growing(
            lambda o, cap, olen: N.lib().ahip_zlib_decode(buf, n, int(verify), int(raw), o, cap, olen), n)
        self.last_status = st
        self.input_position = N.lib().ahip_last_consumed()  # where decodeStream leaves its InputStream
        return out

    def decode_stream(self, input_bytes, output, verify=False, raw=False):
        """decodeStream(input, output) -> bool, with bytes in / bytearray out."""
        out = self.decode_bytes(input_bytes, verify=verify, raw=raw)
        output.extend(out)
        return self.last_status == N.AHIP_OK

    decodeBytes = decode_bytes
    decodeStream = decode_stream


class GZipDecoder:
    """`GZipDecoder().decodeBytes(data, verify: false, raw: false)` -- multi-member gzip;
    members are inflated in parallel, one wavefront each."""

    def decode_bytes(self, data, verify=False, raw=False):
        buf, n = _as_buffer(data)
        st, out = _call_growing(
            lambda o, cap, olen: N.lib().ahip_gzip_decode(buf, n, int(verify), int(raw), o, cap, olen), n)
        self.last_status = st
        self.input_position = N.lib().ahip_last_consumed()  # where decodeStream leaves its InputStream
        return out

    def decode_stream(self, input_bytes, output, verify=False, raw=False):
        out = self.decode_bytes(input_bytes, verify=verify, raw=raw)
        output.extend(out)
        # _GZipDecoder.decodeStream returns true unless it fell through to the zlib decoder
        return self.last_status == N.AHIP_OK

    decodeBytes = decode_bytes
    decodeStream = decode_stream


class BZip2Decoder:
    """`BZip2Decoder().decodeBytes(data, verify: false)` (bzip2_decoder.dart:12-88); blocks are
    decoded in parallel, one wavefront each."""

    def decode_bytes(self, data, verify=False):
        buf, n = _as_buffer(data)
        st, out = _call_growing(lambda o, cap, olen: N.lib().ahip_bzip2_decode(buf, n, int(verify), o, cap, olen), n,
                                hint=8 * n + 1024)
        self.last_status = st
        # where decodeStream leaves its InputStream -- true: behind the last block or marker it read; false: where the failing
        # check stood (the bytes its Bz2BitReader had pulled, bzip2/bz2_bit_reader.dart:12-44)
        self.input_position = N.lib().ahip_last_consumed()
        return out

    def decode_stream(self, input_bytes, output, verify=False):
        out = self.decode_bytes(input_bytes, verify=verify)
        output.extend(out)
        return self.last_status == N.AHIP_OK

    decodeBytes = decode_bytes
    decodeStream = decode_stream


class DeflateLevel:
    """deflate.dart:10-18"""
    none = 0
    bestSpeed = 1
    defaultCompression = 6
    bestCompression = 9


class Deflate:
    """`Deflate(bytes, level: 6).getBytes()` -- raw DEFLATE encode (deflate.dart:25-99).

    Output is valid DEFLATE whose size is within the tolerance stated in DESIGN.md of the
    reference's; an invalid level yields empty output, like the reference's silent `_init`."""

    def __init__(self, data, level=DeflateLevel.defaultCompression, window_bits=15, output=None):
        buf, n = _as_buffer(data)
        self.level = level
        crc = ctypes.c_uint32(0)
        bound = N.lib().ahip_deflate_bound(n)
        out = ctypes.create_string_buffer(max(1, bound))
        olen = ctypes.c_size_t(0)
        rc = N.lib().ahip_deflate_raw(buf, n, level, window_bits, out, bound, ctypes.byref(olen), ctypes.byref(crc))
        if rc != N.AHIP_OK:
            _check(rc)
        self._bytes = out.raw[:olen.value]
        self.crc32 = crc.value
        self.total = n
        if output is not None:
            output.extend(self._bytes)

    def get_bytes(self):
        return self._bytes

    def take_bytes(self):
        b, self._bytes = self._bytes, b""
        return b

    def finish(self):
        """Deflate.finish() flushes the pending buffer (deflate.dart:69): everything is already in get_bytes() here."""
        return None

    getBytes = get_bytes
    takeBytes = take_bytes


def _encode(fn, data, extra):
    buf, n = _as_buffer(data)
    bound = N.lib().ahip_deflate_bound(n) + 32
    out = ctypes.create_string_buffer(bound)
    olen = ctypes.c_size_t(0)
    rc = fn(buf, n, *extra, out, bound, ctypes.byref(olen))
    if rc != N.AHIP_OK:
        _check(rc)
    return out.raw[:olen.value]


class ZLibEncoder:
    """`ZLibEncoder().encodeBytes(data, level: 6, raw: false)` (zlib_encoder.dart:14-38)"""

    def encode_bytes(self, data, level=None, window_bits=None, raw=False):
        level = 6 if level is None else level
        if raw:
            return Deflate(data, level=level, window_bits=15 if window_bits is None else window_bits).get_bytes()
        return _encode(N.lib().ahip_zlib_encode, data, (level, 15 if window_bits is None else window_bits))

    encodeBytes = encode_bytes


class GZipEncoder:
    """`GZipEncoder().encodeBytes(data, level: 6)` (gzip_encoder.dart:14-30).  The reference
    stamps the current time into the header; pass `mtime` to make the bytes reproducible."""

    def encode_bytes(self, data, level=None, window_bits=None, raw=False, mtime=None):
        import time
        level = 6 if level is None else level
        if raw:
            return Deflate(data, level=level, window_bits=15 if window_bits is None else window_bits).get_bytes()
        return _encode(N.lib().ahip_gzip_encode, data, (level, 15 if window_bits is None else window_bits, int(time.time()) if mtime is None else mtime))

    encodeBytes = encode_bytes


ZLibEncoderWeb = ZLibEncoder
GZipEncoderWeb = GZipEncoder

# The reference's *Web classes force the pure-Dart path; here both names are the HIP path.
ZLibDecoderWeb = ZLibDecoder
GZipDecoderWeb = GZipDecoder


def get_crc32(data, crc=0):
    buf, n = _as_buffer(data)
    return N.lib().ahip_crc32(buf, n, crc)


def get_adler32(data, adler=1):
    buf, n = _as_buffer(data)
    return N.lib().ahip_adler32(buf, n, adler)


getCrc32 = get_crc32
getAdler32 = get_adler32
