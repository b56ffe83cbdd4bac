// lib/src/codecs/zlib/_gzip_decoder_hip.dart -- third platform implementation of the reference's decoder seam
// (next to _gzip_decoder_io.dart / _gzip_decoder_web.dart; selected in _gzip_decoder.dart:1).  UNTESTED here (no
// Dart SDK in the build image); mirrors ZLibDecoderBase (lib/src/codecs/zlib/_zlib_decoder_base.dart:5-13).
import 'dart:typed_data';

import '../../util/input_stream.dart';
import '../../util/output_stream.dart';
import '_zlib_decoder_base.dart';
import 'archive_hip_ffi.dart';

const platformGZipDecoder = _GZipDecoderHip();

class _GZipDecoderHip extends ZLibDecoderBase {
  const _GZipDecoderHip();

  @override
  Uint8List decodeBytes(List<int> data, {bool verify = false, bool raw = false}) =>
      ArchiveHip.instance.gzipDecode(data, verify: verify, raw: raw);

  /// The reference's gzip decodeStream always returns true unless it fell through to the zlib decoder
  /// (_gzip_decoder_web.dart:27-58); status 1 is exactly that fall-through returning false.
  @override
  bool decodeStream(InputStream input, OutputStream output, {bool verify = false, bool raw = false}) {
    final hip = ArchiveHip.instance;
    final data = input.toUint8List();  // what is left of the stream, from its current position
    output.writeBytes(hip.gzipDecode(data, verify: verify, raw: raw));
    // The reference's decodeStream CONSUMES the stream: all of it when it returns true; on `false` it stands where the
    // failing check left it (two header bytes in, behind a dictionary id, behind a wrong Adler-32:
    // _zlib_decoder_web.dart:53-99).  Callers that go on reading `input` must find it exactly there.
    input.skip(hip.lastStreamPosition);
    return hip.lastStatus == ArchiveHip.ok;
  }
}
