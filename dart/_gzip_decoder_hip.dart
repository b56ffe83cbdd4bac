// lib/src/codecs/zlib/_gzip_decoder_hip.dart -- third platform implementation of the reference's
// decoder seam (next to _gzip_decoder_io.dart / _gzip_decoder_web.dart).  UNTESTED here (no Dart
// SDK in the build image); mirrors ZLibDecoderBase (lib/src/codecs/zlib/_zlib_decoder_base.dart:5-13).
import 'dart:typed_data';

import '../../util/input_stream.dart';
import '../../util/output_stream.dart';
import '_zlib_decoder_base.dart';
import 'archive_hip_ffi.dart';

final _hip = ArchiveHip();

const platformGZipDecoder = _GZipDecoderHip();
const platformZLibDecoder = _ZLibDecoderHip();

class _GZipDecoderHip extends ZLibDecoderBase {
  const _GZipDecoderHip();

  @override
  Uint8List decodeBytes(List<int> data, {bool verify = false, bool raw = false}) =>
      _hip.gzipDecode(data, verify: verify, raw: raw);

  @override
  bool decodeStream(InputStream input, OutputStream output, {bool verify = false, bool raw = false}) {
    output.writeBytes(_hip.gzipDecode(input.toUint8List(), verify: verify, raw: raw));
    return _hip.lastStatus == ArchiveHip.ok;
  }
}

class _ZLibDecoderHip extends ZLibDecoderBase {
  const _ZLibDecoderHip();

  @override
  Uint8List decodeBytes(List<int> data, {bool verify = false, bool raw = false}) =>
      _hip.zlibDecode(data, verify: verify, raw: raw);

  @override
  bool decodeStream(InputStream input, OutputStream output, {bool verify = false, bool raw = false}) {
    output.writeBytes(_hip.zlibDecode(input.toUint8List(), verify: verify, raw: raw));
    return _hip.lastStatus == ArchiveHip.ok;
  }
}
