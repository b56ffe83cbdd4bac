"""archive_amd -- MI355X-native Inflate hot path behind the Dart `archive` codec API.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C-ABI), the ctypes binding
and the host mirror of the reference's codec classes.
"""
from .codecs import (BZip2Decoder, Deflate, DeflateLevel, GZipDecoder, GZipDecoderWeb, GZipEncoder, GZipEncoderWeb, Inflate,
                     ZLibDecoder, ZLibDecoderWeb, ZLibEncoder, ZLibEncoderWeb, get_adler32, get_crc32, getAdler32,
                     getCrc32)
from .errors import ArchiveHipError, RangeError, ReferenceWouldHang

__all__ = ["BZip2Decoder", "Deflate", "DeflateLevel", "ZLibEncoder", "ZLibEncoderWeb", "GZipEncoder", "GZipEncoderWeb", "Inflate", "ZLibDecoder", "ZLibDecoderWeb", "GZipDecoder", "GZipDecoderWeb", "get_crc32", "get_adler32",
           "getCrc32", "getAdler32", "ArchiveHipError", "RangeError", "ReferenceWouldHang"]
