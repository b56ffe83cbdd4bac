// lib/src/codecs/zlib/deflate_hip.dart -- `Deflate` with the reference's constructors and getters
// (lib/src/codecs/zlib/deflate.dart:10-18,31,39-99,1213), compressing on the GPU.  Valid DEFLATE, size within the
// tolerance DESIGN.md states of the reference's; an invalid level / windowBits writes nothing (the reference's
// silent _init, deflate.dart:102-121).  UNTESTED here (no Dart SDK in the build image).
import 'dart:typed_data';

import '../../util/input_memory_stream.dart';
import '../../util/input_stream.dart';
import '../../util/output_memory_stream.dart';
import '../../util/output_stream.dart';
import 'archive_hip_ffi.dart';

class DeflateLevel {
  static const none = 0;
  static const defaultCompression = 6;
  static const bestCompression = 9;
  static const bestSpeed = 1;
  final int value;
  const DeflateLevel(this.value);
}

class Deflate {
  static const maxWindowBits = 15;
  static const zOk = 0;
  static const zStreamEnd = 1;

  final OutputStream _output;
  final int _level;
  final int _windowBits;

  /// CRC-32 of everything compressed so far (Deflate.crc32, deflate.dart:37)
  int crc32 = 0;

  /// total input bytes (deflate.dart:1213)
  int total = 0;

  Deflate(List<int> bytes,
      {int level = DeflateLevel.defaultCompression, int windowBits = maxWindowBits, OutputStream? output})
      : _output = output ?? OutputMemoryStream(),
        _level = level,
        _windowBits = windowBits {
    _deflate(bytes);
  }

  Deflate.stream(InputStream input,
      {int level = DeflateLevel.defaultCompression, int windowBits = maxWindowBits, OutputStream? output})
      : _output = output ?? OutputMemoryStream(),
        _level = level,
        _windowBits = windowBits {
    _deflate(input.toUint8List());
  }

  int _deflate(List<int> bytes) {
    final r = ArchiveHip.instance.deflateRaw(bytes, level: _level, windowBits: _windowBits);
    _output.writeBytes(r.bytes);
    crc32 = r.crc32;
    total += r.total;
    return zStreamEnd; // what the reference's _deflate(finish) returns once everything is out (deflate.dart:1305)
  }

  /// The reference flushes its pending buffer here (deflate.dart:69); nothing is ever pending in this class.
  void finish() {}

  Uint8List getBytes() => _output.getBytes();

  Uint8List takeBytes() {
    final bytes = _output.getBytes();
    _output.clear();
    return bytes;
  }

  /// Each call compresses [bytes] as a stream of its own with flush mode `finish`, exactly what the reference
  /// does (deflate.dart:87-99).
  void addBytes(List<int> bytes) => _deflate(bytes);

  int addStream(InputStream buffer) => _deflate(buffer.toUint8List());

  int get level => _level;
}
