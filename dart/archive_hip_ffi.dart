// dart/archive_hip_ffi.dart -- dart:ffi binding of libarchive_hip.so (include/archive_hip.h, ABI 2.x).
//
// UNTESTED IN THIS REPOSITORY'S CI: the build image has no Dart SDK.  It is the binding a maintainer of
// brendan-duncan/archive drops next to lib/src/codecs/zlib/ (see INTEGRATION.md); the same entry points are
// exercised from Python (archive_amd/_native.py) by the test suite, and tests/test_abi.py checks that every symbol
// looked up here exists in the header with the same number of arguments.
import 'dart:ffi';
import 'dart:typed_data';

import 'package:ffi/ffi.dart';

typedef _DecodeNative = Int32 Function(Pointer<Uint8> input, Size inLen, Int32 verify, Int32 raw,
    Pointer<Uint8> out, Size outCap, Pointer<Size> outLen);
typedef _DecodeDart = int Function(
    Pointer<Uint8> input, int inLen, int verify, int raw, Pointer<Uint8> out, int outCap, Pointer<Size> outLen);
typedef _InflateNative = Int32 Function(Pointer<Uint8> input, Size inLen, Pointer<Uint8> out, Size outCap,
    Pointer<Size> outLen, Pointer<Size> consumed);
typedef _InflateDart = int Function(
    Pointer<Uint8> input, int inLen, Pointer<Uint8> out, int outCap, Pointer<Size> outLen, Pointer<Size> consumed);
typedef _BoundNative = Size Function(Pointer<Uint8> input, Size inLen);
typedef _BoundDart = int Function(Pointer<Uint8> input, int inLen);

typedef _BatchNative = Int32 Function(
    Pointer<Uint8> input,
    Size inLen,
    Uint32 nEntries,
    Pointer<Uint64> inOff,
    Pointer<Uint64> inSize,
    Pointer<Uint64> sizeHint,
    Pointer<Uint8> out,
    Size outCap,
    Pointer<Uint64> outOff,
    Pointer<Uint64> outLen,
    Pointer<Int32> status,
    Pointer<Size> outTotal);
typedef _BatchDart = int Function(
    Pointer<Uint8> input,
    int inLen,
    int nEntries,
    Pointer<Uint64> inOff,
    Pointer<Uint64> inSize,
    Pointer<Uint64> sizeHint,
    Pointer<Uint8> out,
    int outCap,
    Pointer<Uint64> outOff,
    Pointer<Uint64> outLen,
    Pointer<Int32> status,
    Pointer<Size> outTotal);
typedef _DeflateNative = Int32 Function(Pointer<Uint8> input, Size inLen, Int32 level, Int32 windowBits,
    Pointer<Uint8> out, Size outCap, Pointer<Size> outLen, Pointer<Uint32> crc32);
typedef _DeflateDart = int Function(Pointer<Uint8> input, int inLen, int level, int windowBits, Pointer<Uint8> out,
    int outCap, Pointer<Size> outLen, Pointer<Uint32> crc32);
typedef _GzEncodeNative = Int32 Function(Pointer<Uint8> input, Size inLen, Int32 level, Int32 windowBits,
    Uint32 mtime, Pointer<Uint8> out, Size outCap, Pointer<Size> outLen);
typedef _GzEncodeDart = int Function(Pointer<Uint8> input, int inLen, int level, int windowBits, int mtime,
    Pointer<Uint8> out, int outCap, Pointer<Size> outLen);
typedef _ZlEncodeNative = Int32 Function(Pointer<Uint8> input, Size inLen, Int32 level, Int32 windowBits,
    Pointer<Uint8> out, Size outCap, Pointer<Size> outLen);
typedef _ZlEncodeDart = int Function(
    Pointer<Uint8> input, int inLen, int level, int windowBits, Pointer<Uint8> out, int outCap, Pointer<Size> outLen);
typedef _BoundSizeNative = Size Function(Size inLen);
typedef _BoundSizeDart = int Function(int inLen);
typedef _BzNative = Int32 Function(
    Pointer<Uint8> input, Size inLen, Int32 verify, Pointer<Uint8> out, Size outCap, Pointer<Size> outLen);
typedef _BzDart = int Function(
    Pointer<Uint8> input, int inLen, int verify, Pointer<Uint8> out, int outCap, Pointer<Size> outLen);

/// What `Deflate(bytes, ...)` produced: the stream, the CRC-32 of the input, the input length.
class DeflateResult {
  final Uint8List bytes;
  final int crc32;
  final int total;
  DeflateResult(this.bytes, this.crc32, this.total);
}

class ArchiveHip {
  static const ok = 0, stoppedEarly = 1, rangeError = 2, wouldHang = 3, eCap = -1;

  /// One library per isolate group; every class of the seam shares it.
  static final ArchiveHip instance = ArchiveHip();

  final DynamicLibrary _lib;
  late final _DecodeDart _gzip = _lib.lookupFunction<_DecodeNative, _DecodeDart>('ahip_gzip_decode');
  late final _DecodeDart _zlib = _lib.lookupFunction<_DecodeNative, _DecodeDart>('ahip_zlib_decode');
  late final _InflateDart _inflate = _lib.lookupFunction<_InflateNative, _InflateDart>('ahip_inflate_raw');
  late final _BoundDart _decodeBound = _lib.lookupFunction<_BoundNative, _BoundDart>('ahip_decode_bound');
  late final _BatchDart _batch = _lib.lookupFunction<_BatchNative, _BatchDart>('ahip_inflate_batch');
  late final _DeflateDart _deflate = _lib.lookupFunction<_DeflateNative, _DeflateDart>('ahip_deflate_raw');
  late final _GzEncodeDart _gzipEncode = _lib.lookupFunction<_GzEncodeNative, _GzEncodeDart>('ahip_gzip_encode');
  late final _ZlEncodeDart _zlibEncode = _lib.lookupFunction<_ZlEncodeNative, _ZlEncodeDart>('ahip_zlib_encode');
  late final _BoundSizeDart _deflateBound =
      _lib.lookupFunction<_BoundSizeNative, _BoundSizeDart>('ahip_deflate_bound');
  late final _BzDart _bzip2 = _lib.lookupFunction<_BzNative, _BzDart>('ahip_bzip2_decode');
  late final int Function() _lastConsumed = _lib.lookupFunction<Size Function(), int Function()>('ahip_last_consumed');
  late final int Function(int) _init =
      _lib.lookupFunction<Int32 Function(Int32), int Function(int)>('ahip_init');
  late final Pointer<Utf8> Function() _lastError =
      _lib.lookupFunction<Pointer<Utf8> Function(), Pointer<Utf8> Function()>('ahip_last_error');

  late final int Function(int) _initDevices =
      _lib.lookupFunction<Int32 Function(Uint64), int Function(int)>('ahip_init_devices');

  /// [deviceMask]: bit d selects GPU d; with more than one bit set, gzipDecode() spreads the members of a BGZF
  /// stream over those GPUs (one process drives them all).  Default: the current device only.
  ArchiveHip([String path = 'libarchive_hip.so', int? deviceMask]) : _lib = DynamicLibrary.open(path) {
    final rc = deviceMask == null ? _init(-1) : _initDevices(deviceMask);
    if (rc != ok) throw StateError('ahip_init: ${_lastError().toDartString()}');
  }

  /// Runs [call] on [data]; the output buffer is sized by [sizeHint], by `ahip_decode_bound` (the ISIZE trailers
  /// of a gzip stream) or by a guess, and grown once if the library reports the real size (AHIP_E_CAP).  Status
  /// codes are mapped to the reference's behaviour: 0/1 -> bytes (the reference is silent about early stops),
  /// 2 -> RangeError.
  Uint8List _run(List<int> data, int Function(Pointer<Uint8>, int, Pointer<Uint8>, int, Pointer<Size>) call,
      {int? sizeHint, bool askBound = false}) {
    final n = data.length;
    final inp = malloc<Uint8>(n == 0 ? 1 : n);
    inp.asTypedList(n).setAll(0, data);
    final outLen = malloc<Size>();
    var cap = sizeHint ?? 0;
    if (cap == 0 && askBound) cap = _decodeBound(inp, n);
    if (cap == 0) cap = 4 * n + 64;
    try {
      for (var attempt = 0; attempt < 3; ++attempt) {
        final out = malloc<Uint8>(cap == 0 ? 1 : cap);
        try {
          final rc = call(inp, n, out, cap, outLen);
          if (rc == eCap) {
            cap = outLen.value + 64;
            continue;
          }
          if (rc == rangeError) throw RangeError('archive_hip: read past the end of the input');
          if (rc < 0 || rc == wouldHang) throw StateError('archive_hip $rc: ${_lastError().toDartString()}');
          lastStatus = rc;
          lastStreamPosition = _lastConsumed();
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

  /// Bytes of the input the reference's decodeStream would have consumed in the last gzipDecode / zlibDecode /
  /// bzip2Decode: all of them when it returns true; on `false` the position its failing check left the InputStream at
  /// (ahip_last_consumed; _zlib_decoder_web.dart:53-99).
  int lastStreamPosition = 0;

  /// Bytes of the input the last [inflateRaw] consumed (the reference InputStream's position afterwards).
  int lastConsumed = 0;

  Uint8List gzipDecode(List<int> data, {bool verify = false, bool raw = false}) =>
      _run(data, (i, n, o, c, l) => _gzip(i, n, verify ? 1 : 0, raw ? 1 : 0, o, c, l), askBound: true);

  Uint8List zlibDecode(List<int> data, {bool verify = false, bool raw = false}) =>
      _run(data, (i, n, o, c, l) => _zlib(i, n, verify ? 1 : 0, raw ? 1 : 0, o, c, l));

  Uint8List inflateRaw(List<int> data, {int? uncompressedSize}) {
    final consumed = malloc<Size>();
    try {
      final out =
          _run(data, (i, n, o, c, l) => _inflate(i, n, o, c, l, consumed), sizeHint: uncompressedSize);
      lastConsumed = consumed.value;
      return out;
    } finally {
      malloc.free(consumed);
    }
  }

  /// Deflate(bytes, level: level, windowBits: windowBits)  (zlib/deflate.dart:39-48): an invalid level or window
  /// yields no output, like the reference's silent _init.
  DeflateResult deflateRaw(List<int> data, {int level = 6, int windowBits = 15}) {
    final crc = malloc<Uint32>();
    try {
      final out = _run(data, (i, n, o, c, l) => _deflate(i, n, level, windowBits, o, c, l, crc),
          sizeHint: _deflateBound(data.length));
      return DeflateResult(out, crc.value, data.length);
    } finally {
      malloc.free(crc);
    }
  }

  /// ZLibEncoder().encodeBytes(data, level: level, windowBits: windowBits)  (zlib/_zlib_encoder_web.dart:27-73)
  Uint8List zlibEncode(List<int> data, {int level = 6, int windowBits = 15}) =>
      _run(data, (i, n, o, c, l) => _zlibEncode(i, n, level, windowBits, o, c, l),
          sizeHint: _deflateBound(data.length) + 6);

  /// GZipEncoder().encodeBytes(data, level: level)  (zlib/_gzip_encoder_web.dart:27-100); [mtime] defaults to now,
  /// like the reference's DateTime.now().
  Uint8List gzipEncode(List<int> data, {int level = 6, int windowBits = 15, int? mtime}) {
    final t = mtime ?? DateTime.now().millisecondsSinceEpoch ~/ 1000;
    return _run(data, (i, n, o, c, l) => _gzipEncode(i, n, level, windowBits, t, o, c, l),
        sizeHint: _deflateBound(data.length) + 18);
  }

  /// BZip2Decoder().decodeBytes(data, verify: verify)  (bzip2_decoder.dart:13-88)
  Uint8List bzip2Decode(List<int> data, {bool verify = false}) =>
      _run(data, (i, n, o, c, l) => _bzip2(i, n, verify ? 1 : 0, o, c, l), sizeHint: 8 * data.length + 1024);

  /// All DEFLATE entries of a ZIP archive in one call (zip/zip_file.dart:182-248 does them one by one):
  /// entry k is archive[offsets[k], offsets[k] + sizes[k]); uncompressedSizes come from the directory.
  /// Returns the entries' bytes in order; an entry the reference would stop early on keeps what it produced.
  List<Uint8List> inflateEntries(Uint8List archive, List<int> offsets, List<int> sizes, List<int> uncompressedSizes) {
    final k = offsets.length;
    final inp = malloc<Uint8>(archive.isEmpty ? 1 : archive.length);
    inp.asTypedList(archive.length).setAll(0, archive);
    final off = malloc<Uint64>(k), sz = malloc<Uint64>(k), hint = malloc<Uint64>(k);
    final oOff = malloc<Uint64>(k), oLen = malloc<Uint64>(k);
    final st = malloc<Int32>(k);
    final total = malloc<Size>();
    var cap = 0;
    for (var i = 0; i < k; ++i) {
      off[i] = offsets[i];
      sz[i] = sizes[i];
      hint[i] = uncompressedSizes[i];
      cap += uncompressedSizes[i];
    }
    final out = malloc<Uint8>(cap == 0 ? 1 : cap);
    try {
      final rc = _batch(inp, archive.length, k, off, sz, hint, out, cap, oOff, oLen, st, total);
      if (rc != ok) throw StateError('archive_hip $rc: ${_lastError().toDartString()}');
      final view = out.asTypedList(cap);
      return [
        for (var i = 0; i < k; ++i)
          if (st[i] == eCap)
            inflateRaw(archive.sublist(offsets[i], offsets[i] + sizes[i])) // the directory understated it
          else if (st[i] == rangeError)
            throw RangeError('archive_hip: entry $i reads before its first byte')
          else
            Uint8List.fromList(view.sublist(oOff[i], oOff[i] + oLen[i]))
      ];
    } finally {
      for (final p in [off, sz, hint, oOff, oLen]) {
        malloc.free(p);
      }
      malloc.free(inp);
      malloc.free(out);
      malloc.free(st);
      malloc.free(total);
    }
  }
}
