// lib/src/codecs/zlib/_zlib_encoder_hip.dart -- platform implementation of the zlib encoder seam
// (_zlib_encoder.dart:1; reference: _zlib_encoder_web.dart:12-73), implementing ZLibEncoderBase
// (_zlib_encoder_base.dart:5-13).  UNTESTED here (no Dart SDK in the build image).
import 'dart:typed_data';

import '../../util/input_stream.dart';
import '../../util/output_stream.dart';
import '_zlib_encoder_base.dart';
import 'archive_hip_ffi.dart';

const platformZLibEncoder = _ZLibEncoderHip();

class _ZLibEncoderHip extends ZLibEncoderBase {
  const _ZLibEncoderHip();

  @override
  Uint8List encodeBytes(List<int> bytes, {int? level, int? windowBits, bool raw = false}) {
    final hip = ArchiveHip.instance;
    if (raw) return hip.deflateRaw(bytes, level: level ?? 6, windowBits: windowBits ?? 15).bytes;
    return hip.zlibEncode(bytes, level: level ?? 6, windowBits: windowBits ?? 15);
  }

  @override
  void encodeStream(InputStream input, OutputStream output, {int? level, int? windowBits, bool raw = false}) {
    output.writeBytes(encodeBytes(input.toUint8List(), level: level, windowBits: windowBits, raw: raw));
    output.flush();
  }
}
