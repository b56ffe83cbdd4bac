"""CPU-only: the C-ABI library builds for gfx950, loads, exports every symbol
include/archive_hip.h declares, and fails loudly (no CPU decode path) without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(native_built):
    from archive_amd import _native as N
    header = open(os.path.join(ROOT, "include", "archive_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ahip_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(N.EXPORTS)
    L = ctypes.CDLL(native_built)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert N.lib().ahip_abi_version() >> 16 == 2


def test_host_checksums_match_reference_kats(native_built, golden):
    import archive_amd

    def expand(s):
        if "*" in s:
            h, n = s.split("*")
            return bytes.fromhex(h) * int(n)
        return bytes.fromhex(s)
    for h, want in golden["checksum_kat"]["crc32"]:
        assert archive_amd.get_crc32(expand(h)) == int(want, 16)
    for h, want in golden["checksum_kat"]["adler32"]:
        assert archive_amd.get_adler32(expand(h)) == int(want, 16)
    assert archive_amd.get_crc32(b"world", archive_amd.get_crc32(b"hello ")) == archive_amd.get_crc32(b"hello world")


def test_no_cpu_fallback_without_gpu(native_built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import archive_amd
    with pytest.raises(archive_amd.ArchiveHipError) as e:
        archive_amd.GZipDecoder().decode_bytes(b"\x1f\x8b\x08\x00" + bytes(20))
    assert e.value.code == -2  # AHIP_E_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "archive_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "oracle" not in src.lower(), os.path.join(root, f)


def _split_args(arglist):
    """top-level comma split (no nested parentheses / angle brackets inside one argument are cut)"""
    out, depth, cur = [], 0, ""
    for ch in arglist:
        if ch in "(<":
            depth += 1
        elif ch in ")>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_dart_bindings_match_the_header():
    """No Dart SDK here, so the binding cannot be run; what can be checked is that every symbol dart/*.dart looks up is
    declared in include/archive_hip.h with the same number of parameters, and that the seam files implement the
    reference's base classes (lib/src/codecs/zlib/_zlib_{en,de}coder_base.dart)."""
    header = open(os.path.join(ROOT, "include", "archive_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = {m.group(1): _split_args(m.group(2)) for m in re.finditer(r"\b(ahip_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", header, flags=re.S)}
    dart_dir = os.path.join(ROOT, "dart")
    ffi = open(os.path.join(dart_dir, "archive_hip_ffi.dart")).read()
    typedefs = {m.group(1): _split_args(m.group(2)) for m in re.finditer(r"typedef\s+(\w+)\s*=\s*\w+(?:<\w+>)?\s+Function\((.*?)\);", ffi, flags=re.S)}
    looked_up = re.findall(r"lookupFunction<\s*([^,]+?),.*?>\(\s*'(ahip_[a-z0-9_]+)'\)", ffi, flags=re.S)
    assert len(looked_up) >= 12
    for native, sym in looked_up:
        assert sym in protos, sym
        want = [a for a in protos[sym] if a != "void"]
        native = native.strip()
        if native in typedefs:
            got = typedefs[native]
        else:  # inline `Ret Function(args)`
            got = _split_args(re.search(r"Function\((.*)\)", native, flags=re.S).group(1))
        assert len(got) == len(want), (sym, got, want)
    for needed in ["ahip_gzip_decode", "ahip_zlib_decode", "ahip_inflate_raw", "ahip_deflate_raw", "ahip_gzip_encode",
                   "ahip_zlib_encode", "ahip_decode_bound", "ahip_inflate_batch", "ahip_bzip2_decode"]:
        assert needed in [s_ for _, s_ in looked_up], needed
    seam = {"_gzip_decoder_hip.dart": ("platformGZipDecoder", "ZLibDecoderBase"), "_zlib_decoder_hip.dart": ("platformZLibDecoder", "ZLibDecoderBase"),
            "_gzip_encoder_hip.dart": ("platformGZipEncoder", "ZLibEncoderBase"), "_zlib_encoder_hip.dart": ("platformZLibEncoder", "ZLibEncoderBase")}
    for f, (const, base) in seam.items():
        src = open(os.path.join(dart_dir, f)).read()
        assert re.search(r"const\s+%s\s*=" % const, src) and ("extends %s" % base) in src, f
        for method in (("decodeBytes", "decodeStream") if "Decoder" in base else ("encodeBytes", "encodeStream")):
            assert re.search(r"\b%s\(" % method, src), (f, method)
    inflate = open(os.path.join(dart_dir, "inflate_hip.dart")).read()
    for piece in ["Inflate(List<int> bytes, {OutputStream? output, int? uncompressedSize})", "Inflate.stream(InputStream? input",
                  "void addStream(InputStream stream)", "void addBytes(List<int> bytes)", "Uint8List getBytes()"]:
        assert piece in inflate, piece
    deflate = open(os.path.join(dart_dir, "deflate_hip.dart")).read()
    for piece in ["Deflate(List<int> bytes,", "Deflate.stream(InputStream input,", "void finish()", "Uint8List getBytes()", "Uint8List takeBytes()",
                  "void addBytes(List<int> bytes)", "int addStream(InputStream buffer)", "int get level", "int crc32", "int total",
                  "static const defaultCompression = 6", "static const bestCompression = 9", "static const bestSpeed = 1", "static const none = 0"]:
        assert piece in deflate, piece
    # BZip2Decoder with the reference's two methods and their signatures (bzip2_decoder.dart:12-21)
    bz = open(os.path.join(dart_dir, "bzip2_decoder_hip.dart")).read()
    for piece in ["class BZip2Decoder", "Uint8List decodeBytes(List<int> data, {bool verify = false})",
                  "bool decodeStream(InputStream input, OutputStream output, {bool verify = false})", "bzip2Decode(data, verify: verify)"]:
        assert piece in bz, piece
    # the decoders' decodeStream leaves the InputStream where the reference does (ahip_last_consumed), not at its end
    assert "ahip_last_consumed" in [s_ for _, s_ in looked_up]
    for f in ("_gzip_decoder_hip.dart", "_zlib_decoder_hip.dart", "bzip2_decoder_hip.dart"):
        src = open(os.path.join(dart_dir, f)).read()
        assert "input.skip(hip.lastStreamPosition)" in src and "input.skip(data.length)" not in src, f
