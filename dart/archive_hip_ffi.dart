// dart/archive_hip_ffi.dart -- dart:ffi binding of libarchive_hip.so (include/archive_hip.h).
//
// UNTESTED IN THIS REPOSITORY'S CI: the build image has no Dart SDK.  It is the binding a
// maintainer of brendan-duncan/archive drops next to lib/src/codecs/zlib/ (see INTEGRATION.md);
// the same entry points are exercised from Python (archive_amd/_native.py) by the test suite.
import 'dart:ffi';
import 'dart:typed_data';

import 'package:ffi/ffi.dart';

typedef _DecodeNative = Int32 Function(Pointer<Uint8> input, IntPtr inLen, Int32 verify, Int32 raw,
    Pointer<Uint8> out, IntPtr outCap, Pointer<IntPtr> outLen);
typedef _DecodeDart = int Function(
    Pointer<Uint8> input, int inLen, int verify, int raw, Pointer<Uint8> out, int outCap, Pointer<IntPtr> outLen);
typedef _InflateNative = Int32 Function(Pointer<Uint8> input, IntPtr inLen, Pointer<Uint8> out, IntPtr outCap,
    Pointer<IntPtr> outLen, Pointer<IntPtr> consumed);
typedef _InflateDart = int Function(
    Pointer<Uint8> input, int inLen, Pointer<Uint8> out, int outCap, Pointer<IntPtr> outLen, Pointer<IntPtr> consumed);

class ArchiveHip {
  static const ok = 0, stoppedEarly = 1, rangeError = 2, wouldHang = 3, eCap = -1;

  final DynamicLibrary _lib;
  late final _DecodeDart _gzip = _lib.lookupFunction<_DecodeNative, _DecodeDart>('ahip_gzip_decode');
  late final _DecodeDart _zlib = _lib.lookupFunction<_DecodeNative, _DecodeDart>('ahip_zlib_decode');
  late final _InflateDart _inflate = _lib.lookupFunction<_InflateNative, _InflateDart>('ahip_inflate_raw');
  late final int Function(int) _init =
      _lib.lookupFunction<Int32 Function(Int32), int Function(int)>('ahip_init');
  late final Pointer<Utf8> Function() _lastError =
      _lib.lookupFunction<Pointer<Utf8> Function(), Pointer<Utf8> Function()>('ahip_last_error');

  ArchiveHip([String path = 'libarchive_hip.so']) : _lib = DynamicLibrary.open(path) {
    final rc = _init(-1);
    if (rc != ok) throw StateError('ahip_init: ${_lastError().toDartString()}');
  }

  /// Runs [call] with a growing output buffer; maps status codes to the reference's behaviour:
  /// 0/1 -> bytes (the reference is silent about early stops), 2 -> RangeError.
  Uint8List _run(List<int> data, int Function(Pointer<Uint8>, int, Pointer<Uint8>, int, Pointer<IntPtr>) call,
      {int? sizeHint}) {
    final n = data.length;
    final inp = malloc<Uint8>(n == 0 ? 1 : n);
    inp.asTypedList(n).setAll(0, data);
    final outLen = malloc<IntPtr>();
    var cap = sizeHint ?? (4 * n + 64);
    try {
      for (var attempt = 0; attempt < 3; ++attempt) {
        final out = malloc<Uint8>(cap);
        try {
          final rc = call(inp, n, out, cap, outLen);
          if (rc == eCap) {
            cap = outLen.value + 64;
            continue;
          }
          if (rc == rangeError) throw RangeError('archive_hip: read past the end of the input');
          if (rc < 0 || rc == wouldHang) throw StateError('archive_hip $rc: ${_lastError().toDartString()}');
          lastStatus = rc;
          return Uint8List.fromList(out.asTypedList(outLen.value));
        } finally {
          malloc.free(out);
        }
      }
      throw StateError('archive_hip: output size did not settle');
    } finally {
      malloc.free(inp);
      malloc.free(outLen);
    }
  }

  int lastStatus = 0;

  Uint8List gzipDecode(List<int> data, {bool verify = false, bool raw = false}) =>
      _run(data, (i, n, o, c, l) => _gzip(i, n, verify ? 1 : 0, raw ? 1 : 0, o, c, l));

  Uint8List zlibDecode(List<int> data, {bool verify = false, bool raw = false}) =>
      _run(data, (i, n, o, c, l) => _zlib(i, n, verify ? 1 : 0, raw ? 1 : 0, o, c, l));

  Uint8List inflateRaw(List<int> data, {int? uncompressedSize}) {
    final consumed = malloc<IntPtr>();
    try {
      return _run(data, (i, n, o, c, l) => _inflate(i, n, o, c, l, consumed), sizeHint: uncompressedSize);
    } finally {
      malloc.free(consumed);
    }
  }
}
