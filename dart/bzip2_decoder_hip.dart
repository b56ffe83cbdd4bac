// lib/src/codecs/bzip2_decoder_hip.dart -- BZip2Decoder with the reference's surface (lib/src/codecs/bzip2_decoder.dart:12-21:
// `decodeBytes(List<int> data, {bool verify = false})`, `decodeStream(InputStream, OutputStream, {bool verify = false})`),
// the blocks of the stream decoded in parallel on the GPU (ahip_bzip2_decode).  A maintainer swaps the export of
// `bzip2_decoder.dart` for this file (INTEGRATION.md); every caller -- `TarBZ2` helpers, `ZipFile` entries with method 12,
// `io/extract_archive_to_disk.dart` -- keeps its code.  UNTESTED here (no Dart SDK in the build image).
import 'dart:typed_data';

import '../util/input_stream.dart';
import '../util/output_stream.dart';
import 'zlib/archive_hip_ffi.dart';

/// Decompress bzip2 compressed data (one stream: like the reference, decoding stops at the first end-of-stream block).
class BZip2Decoder {
  /// `output.getBytes()` of the reference: whatever was written before a silent `return false` is kept
  /// (bzip2_decoder.dart:13-18 ignores decodeStream's bool); a read past the end of the data throws RangeError.
  Uint8List decodeBytes(List<int> data, {bool verify = false}) => ArchiveHip.instance.bzip2Decode(data, verify: verify);

  /// The reference's bool: true = the stream ended with its end-of-stream block (or the input ended behind a block),
  /// false = bad signature / block size / block magic / block data, or with [verify] a CRC that does not match
  /// (bzip2_decoder.dart:29-87).
  bool decodeStream(InputStream input, OutputStream output, {bool verify = false}) {
    final hip = ArchiveHip.instance;
    final data = input.toUint8List();  // what is left of the stream, from its current position
    output.writeBytes(hip.bzip2Decode(data, verify: verify));
    // the reference's bit reader has pulled exactly these bytes (bzip2/bz2_bit_reader.dart:12-44): up to the end of the
    // end-of-stream block's CRC on `true`; on `false` up to the check that failed -- the signature, a block magic, a block's
    // CRC under [verify], or wherever _readCompressed gave up inside a damaged block (ahip_last_consumed)
    input.skip(hip.lastStreamPosition);
    if (hip.lastStatus == ArchiveHip.ok) {
      output.flush();
      return true;
    }
    return false;
  }
}
