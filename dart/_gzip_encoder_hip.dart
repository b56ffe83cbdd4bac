// lib/src/codecs/zlib/_gzip_encoder_hip.dart -- platform implementation of the gzip encoder seam
// (_gzip_encoder.dart:1; reference: _gzip_encoder_web.dart:12-100), implementing ZLibEncoderBase
// (_zlib_encoder_base.dart:5-13).  The header carries the current time like the reference's
// (`DateTime.now()`, _gzip_encoder_web.dart:81).  UNTESTED here (no Dart SDK in the build image).
import 'dart:typed_data';

import '../../util/input_stream.dart';
import '../../util/output_stream.dart';
import '_zlib_encoder_base.dart';
import 'archive_hip_ffi.dart';

const platformGZipEncoder = _GZipEncoderHip();

class _GZipEncoderHip extends ZLibEncoderBase {
  const _GZipEncoderHip();

  @override
  Uint8List encodeBytes(List<int> bytes, {int? level, int? windowBits, bool raw = false}) {
    final hip = ArchiveHip.instance;
    if (raw) return hip.deflateRaw(bytes, level: level ?? 6, windowBits: windowBits ?? 15).bytes;
    return hip.gzipEncode(bytes, level: level ?? 6, windowBits: windowBits ?? 15);
  }

  @override
  void encodeStream(InputStream input, OutputStream output, {int? level, int? windowBits, bool raw = false}) {
    output.writeBytes(encodeBytes(input.toUint8List(), level: level, windowBits: windowBits, raw: raw));
    output.flush();
  }
}
