// lib/src/codecs/zlib/_zlib_decoder_hip.dart -- platform implementation of the zlib decoder seam
// (_zlib_decoder.dart:1; reference implementations _zlib_decoder_io.dart / _zlib_decoder_web.dart:12-107).
// UNTESTED here (no Dart SDK in the build image).
import 'dart:typed_data';

import '../../util/input_stream.dart';
import '../../util/output_stream.dart';
import '_zlib_decoder_base.dart';
import 'archive_hip_ffi.dart';

const platformZLibDecoder = _ZLibDecoderHip();

class _ZLibDecoderHip extends ZLibDecoderBase {
  const _ZLibDecoderHip();

  @override
  Uint8List decodeBytes(List<int> data, {bool verify = false, bool raw = false}) =>
      ArchiveHip.instance.zlibDecode(data, verify: verify, raw: raw);

  @override
  bool decodeStream(InputStream input, OutputStream output, {bool verify = false, bool raw = false}) {
    final hip = ArchiveHip.instance;
    final data = input.toUint8List();  // what is left of the stream, from its current position
    output.writeBytes(hip.zlibDecode(data, verify: verify, raw: raw));
    // The reference's decodeStream CONSUMES the stream: all of it when it returns true; on `false` it stands where the
    // failing check left it (two header bytes in, behind a dictionary id, behind a wrong Adler-32:
    // _zlib_decoder_web.dart:53-99).  Callers that go on reading `input` must find it exactly there.
    input.skip(hip.lastStreamPosition);
    return hip.lastStatus == ArchiveHip.ok;
  }
}
