// lib/src/codecs/zlib/inflate_hip.dart -- `Inflate` with the reference's constructors and getters
// (lib/src/codecs/zlib/inflate.dart:23-58,102), decoding on the GPU.  Export it in place of inflate.dart
// (lib/archive.dart) to route `Inflate(bytes).getBytes()` callers (zip/zip_file.dart:184,229,232) here.
// UNTESTED here (no Dart SDK in the build image).
import 'dart:typed_data';

import '../../util/input_memory_stream.dart';
import '../../util/input_stream.dart';
import '../../util/output_memory_stream.dart';
import '../../util/output_stream.dart';
import 'archive_hip_ffi.dart';
import 'inflate.dart' as ref;

class Inflate {
  final OutputStream _output;
  final int? _uncompressedSize;
  ref.Inflate? _streaming; // the reference's own block-at-a-time path, only for addBytes / addStream callers

  /// Inflate(bytes, output: ..., uncompressedSize: ...)  (inflate.dart:23-27): whole buffer, on the GPU.
  Inflate(List<int> bytes, {OutputStream? output, int? uncompressedSize})
      : _output = output ?? OutputMemoryStream(size: uncompressedSize),
        _uncompressedSize = uncompressedSize {
    _inflateAll(bytes, null);
  }

  /// Inflate.stream(input, ...)  (inflate.dart:29-32).  A null [input] starts an incremental decode that is fed by
  /// [addBytes] / [addStream] (inflate.dart:36-58); that mode stays on the reference's pure-Dart decoder -- the GPU
  /// path takes whole streams.
  Inflate.stream(InputStream? input, {OutputStream? output, int? uncompressedSize})
      : _output = output ?? OutputMemoryStream(size: uncompressedSize),
        _uncompressedSize = uncompressedSize {
    if (input == null) {
      _streaming = ref.Inflate.stream(null, output: _output, uncompressedSize: uncompressedSize);
    } else {
      _inflateAll(input.toUint8List(), input);
    }
  }

  void _inflateAll(List<int> bytes, InputStream? source) {
    final hip = ArchiveHip.instance;
    _output.writeBytes(hip.inflateRaw(bytes, uncompressedSize: _uncompressedSize));
    // leave the caller's stream where the reference would: right behind the deflate data (inflate.dart:104-116)
    source?.skip(hip.lastConsumed);
  }

  void addStream(InputStream stream) =>
      (_streaming ??= ref.Inflate.stream(null, output: _output, uncompressedSize: _uncompressedSize))
          .addStream(stream);

  void addBytes(List<int> bytes) => addStream(InputMemoryStream(bytes));

  /// Inflate.getBytes()  (inflate.dart:102)
  Uint8List getBytes() => _output.getBytes();
}
