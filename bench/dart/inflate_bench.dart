// inflate_bench.dart -- the reference's own CPU path on the benchmark workload (SURVEY.md section 8d, baseline 2).
//
// For boxes that have a Dart SDK (this repository's build image has none: `which dart` finds nothing, so this file has
// never been run here; bench.py's `cpu_baseline` times the C restatement of the same algorithm instead).  It times
// `GZipDecoderWeb().decodeBytes` -- lib/src/codecs/zlib/gzip_decoder_web.dart:12 -> _gzip_decoder_web.dart:19-58, the
// pure-Dart member loop + lib/src/codecs/zlib/inflate.dart -- on a multi-member .gz file, single isolate and N isolates
// (members are independent: each isolate gets a contiguous range of members, like the GPU shards).
//
//   dart pub get                       (in a checkout of brendan-duncan/archive 4.x)
//   dart run bench/dart/inflate_bench.dart <file.gz> [isolates] [seconds]
//
// Make the file with the generator this repository benchmarks on:
//   python -c "from tools import corpus; c, _ = corpus.make_gzip(n_members=4096); open('bench.gz', 'wb').write(c.tobytes())"
// Output: one JSON line {"cpu_baseline": {"value": GB/s of decoded output, "unit": "GB/s", "cores": N, "kind": "reference", ...}}.
import 'dart:convert';
import 'dart:io';
import 'dart:isolate';
import 'dart:typed_data';

import 'package:archive/archive.dart';

/// offsets of the gzip members of a BGZF-style stream (every member carries the `BC` subfield: total size - 1)
List<int> memberOffsets(Uint8List data) {
  final offs = <int>[0];
  var pos = 0;
  while (pos + 18 <= data.length && data[pos] == 0x1f && data[pos + 1] == 0x8b) {
    final bsize = data[pos + 16] | (data[pos + 17] << 8);
    pos += bsize + 1;
    offs.add(pos);
  }
  return offs;
}

int decodeRange(Uint8List data) => GZipDecoderWeb().decodeBytes(data).length;

Future<void> main(List<String> args) async {
  if (args.isEmpty) {
    stderr.writeln('usage: dart run bench/dart/inflate_bench.dart <file.gz> [isolates] [seconds]');
    exit(2);
  }
  final data = File(args[0]).readAsBytesSync();
  final isolates = args.length > 1 ? int.parse(args[1]) : 1;
  final seconds = args.length > 2 ? double.parse(args[2]) : 10.0;
  final offs = memberOffsets(data);
  final members = offs.length - 1;
  // contiguous member ranges, balanced on compressed bytes (what archive_amd/sharding.py::partition_members does)
  final slices = <Uint8List>[];
  var lo = 0;
  for (var r = 1; r <= isolates; ++r) {
    var hi = lo;
    final target = data.length * r ~/ isolates;
    while (hi < members && offs[hi + 1] <= target) {
      ++hi;
    }
    if (r == isolates) hi = members;
    slices.add(Uint8List.sublistView(data, offs[lo], offs[hi]));
    lo = hi;
  }
  decodeRange(slices[0]); // warm-up (JIT)
  var outBytes = 0;
  var rounds = 0;
  final sw = Stopwatch()..start();
  while (sw.elapsedMicroseconds < seconds * 1e6) {
    final parts = await Future.wait([for (final s in slices) Isolate.run(() => decodeRange(s))]);
    outBytes += parts.fold<int>(0, (a, b) => a + b);
    ++rounds;
  }
  final el = sw.elapsedMicroseconds / 1e6;
  print(jsonEncode({
    'cpu_baseline': {
      'value': outBytes / el / 1e9,
      'unit': 'GB/s',
      'cores': isolates,
      'kind': 'reference',
      'sample': '$rounds x $members gzip members (${data.length} compressed bytes) through GZipDecoderWeb().decodeBytes, $isolates isolate(s), ${el.toStringAsFixed(1)} s',
    }
  }));
}
