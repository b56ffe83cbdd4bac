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
    // the reference chains the checksum over every call (deflate.dart:1231 `crc32 = getCrc32(bytes, crc32)`): the
    // library returns the CRC-32 of this call's bytes alone, so the two are combined over GF(2)
    crc32 = total == 0 ? r.crc32 : _crc32Combine(crc32, r.crc32, r.total);
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

  /// CRC-32 of A || B from CRC-32(A), CRC-32(B) and len(B): multiplication by x^(8 len) in GF(2)[x] / P
  /// (square-and-multiply on the 32 x 32 bit matrix of "append one zero bit").
  static int _crc32Combine(int crcA, int crcB, int lenB) {
    if (lenB <= 0) return crcA;
    List<int> times(List<int> mat, List<int> m2) => [for (final v in m2) _gf2Times(mat, v)];
    var odd = List<int>.filled(32, 0); // one zero bit
    odd[0] = 0xedb88320;
    for (var n = 1, row = 1; n < 32; ++n, row <<= 1) {
      odd[n] = row;
    }
    var even = times(odd, odd); // two zero bits
    odd = times(even, even); // four
    var len = lenB;
    var crc = crcA;
    do {
      even = times(odd, odd); // first pass: one zero BYTE
      if (len & 1 != 0) crc = _gf2Times(even, crc);
      len >>= 1;
      if (len == 0) break;
      odd = times(even, even);
      if (len & 1 != 0) crc = _gf2Times(odd, crc);
      len >>= 1;
    } while (len != 0);
    return (crc ^ crcB) & 0xffffffff;
  }

  static int _gf2Times(List<int> mat, int vec) {
    var sum = 0;
    for (var i = 0; vec != 0; vec >>= 1, ++i) {
      if (vec & 1 != 0) sum ^= mat[i];
    }
    return sum;
  }
}
