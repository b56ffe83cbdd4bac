/*
 * archive_hip.h -- C-ABI of libarchive_hip.so: the MI355X (gfx950) implementation of the
 * Inflate/Deflate hot path of the Dart `archive` package (brendan-duncan/archive 4.2.0).
 *
 * This is the drop-in boundary: these are the entry points a dart:ffi binding (see
 * INTEGRATION.md, dart/) or the Python host mirror (archive_amd/) binds.  Plain pointers and
 * sizes only; no C++ exceptions cross it; the library never frees caller memory.
 * All `ref:` citations are relative to /root/reference/lib/src/.
 *
 * Status codes (shared by every entry point that returns int32_t):
 *    0  AHIP_OK      the reference's decodeStream returned true / Inflate reached a final block
 *    1  AHIP_FALSE   the reference stopped early (its silent `return false` / `-1`); the bytes
 *                    produced so far are in `out` exactly as the reference would have kept them
 *    2  AHIP_RANGE   the reference would throw RangeError (read past the end of the buffer,
 *                    back-reference before the start of the stream); no output is defined
 *    3  AHIP_HANG    the reference would not terminate (zero-length litlen table entry)
 *   -1  AHIP_E_CAP   `out` is too small; *out_len holds the required size
 *   -2  AHIP_E_DEVICE no usable GPU / HIP runtime error (ahip_last_error() has the text)
 *   -3  AHIP_E_UNSUPPORTED input uses a construct this build does not reproduce (only bzip2's obsolete
 *                    randomised blocks)
 *   -4  AHIP_E_ARG   bad argument
 */
#ifndef ARCHIVE_HIP_H
#define ARCHIVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AHIP_OK 0
#define AHIP_FALSE 1
#define AHIP_RANGE 2
#define AHIP_HANG 3
#define AHIP_E_CAP (-1)
#define AHIP_E_DEVICE (-2)
#define AHIP_E_UNSUPPORTED (-3)
#define AHIP_E_ARG (-4)

/* ---- library lifetime ---- */
/* Selects the HIP device for this process (one process per GPU).  device < 0 keeps the
 * current device.  Returns AHIP_OK or AHIP_E_DEVICE.  Idempotent. */
int32_t ahip_init(int32_t device);
/* One process driving several GPUs: bit d of device_mask selects HIP device d.  Starts one worker thread (= one device
 * context: scratch pools, streams) per selected device; afterwards ahip_gzip_decode() partitions a multi-member stream
 * whose members carry the BGZF `BC` subfield into contiguous member ranges balanced on compressed bytes, and every
 * device uploads, decodes and downloads only its slice (output offsets = prefix sums of the ISIZE trailers, checked).
 * Streams that cannot be partitioned that way, and every other entry point, run on the first selected device with
 * unchanged semantics.  Calling it again replaces the set; ahip_shutdown() stops the workers. */
int32_t ahip_init_devices(uint64_t device_mask);
/* number of device contexts ahip_gzip_decode() fans out over (1 unless ahip_init_devices selected more) */
int32_t ahip_device_count(void);
/* Diagnostics: how many device contexts the last ahip_gzip_decode() on this process used (1 = the exact single-device path). */
int32_t ahip_debug_last_shards(void);
void ahip_shutdown(void);
/* Text of the last AHIP_E_* error on this thread ("" if none).  Never NULL. */
const char *ahip_last_error(void);
/* ABI version of this header (major << 16 | minor). */
uint32_t ahip_abi_version(void);  /* 2.1: + ahip_gzip_decode_shards, ahip_gzip_encode_device, ahip_zlib_encode_device;
                                    * 2.2: + ahip_deflate_shards, ahip_bzip2_decode_shards, ahip_debug_last_chunks;
                                    * 2.3: + ahip_debug_bz_reruns, ahip_last_consumed;
                                    * 2.4: + ahip_stream_split_*, ahip_inflate_stream_shards, ahip_deflate_piece_device,
                                    *      ahip_bzip2_decode_range_device */

/* ---- Inflate: host-pointer entry points (what dart:ffi binds) ---- */

/* ref: codecs/zlib/inflate.dart:23-39,104-116  `Inflate(bytes).getBytes()`.
 * Raw DEFLATE stream -> bytes.  *consumed = the reference InputStream position afterwards. */
int32_t ahip_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                         size_t *out_len, size_t *consumed);

/* ref: codecs/zlib/_gzip_decoder_web.dart:19-58  `GZipDecoderWeb().decodeBytes(data, verify, raw)`.
 * Multi-member gzip; CRC32/ISIZE are not verified (the reference ignores them); a non-gzip
 * header falls through to the zlib decoder exactly as the reference does. */
int32_t ahip_gzip_decode(const uint8_t *in, size_t in_len, int32_t verify, int32_t raw,
                         uint8_t *out, size_t out_cap, size_t *out_len);

/* ref: codecs/zlib/_zlib_decoder_web.dart:21-107  `ZLibDecoderWeb().decodeBytes(data, verify, raw)`.
 * Multi-member zlib with the reference's deferred-flush behaviour; Adler-32 big-endian. */
int32_t ahip_zlib_decode(const uint8_t *in, size_t in_len, int32_t verify, int32_t raw,
                         uint8_t *out, size_t out_cap, size_t *out_len);

/* Where the reference's InputStream stands after the calling thread's last ahip_gzip_decode / ahip_zlib_decode /
 * ahip_inflate_raw -- what `decodeStream(input, output)` has consumed of `input`: everything when it returned true; on `false`
 * the position the failing check left the reader at (ref: _zlib_decoder_web.dart:53-99: two header bytes, + 4 for a
 * preset-dictionary id, behind the Adler-32 that did not match; _gzip_decoder_web.dart:31-37 rewinds to the start of what
 * is not a gzip header before it hands over).  After ahip_bzip2_decode: the bytes the reference's bit reader has pulled
 * (ref: bzip2/bz2_bit_reader.dart:12-44 -- whole bytes, rounded up) -- AHIP_OK: behind the end-of-stream marker's CRC (or the
 * last block when the input ends without one); AHIP_FALSE: where the check that failed stood -- the signature byte that
 * differs, the first byte that fits neither block magic (bzip2_decoder.dart:90-111), behind a block whose CRC `verify`
 * rejects, and inside a damaged block wherever _readCompressed gave up (bzip2_decoder.dart:113-730: the header's own bit; in
 * the symbol loop the reference's loop is run once more for that one block to find it).  Undefined after AHIP_RANGE / an
 * AHIP_E_* code. */
size_t ahip_last_consumed(void);

/* Size the decoded output is expected to have, so that a caller can allocate once: for a gzip stream whose members all
 * carry a BGZF `BC` subfield (or that is one member) the sum of the ISIZE trailers; 0 = unknown (zlib / raw input, members
 * without `BC`, ISIZE that cannot be trusted because the stream is > 4 GiB).  Host-only, reads a few bytes per member;
 * the decode never relies on it (the reference ignores ISIZE, _gzip_decoder_web.dart:43-44): a too-small guess just
 * comes back as AHIP_E_CAP with the real size. */
size_t ahip_decode_bound(const uint8_t *in, size_t in_len);

/* ---- Inflate: device-resident entry points (bench, multi-GPU sharding, zero-copy callers) ----
 * d_in / d_out are device pointers on the current HIP device; `stream` is a hipStream_t
 * (NULL = the default stream).  The call enqueues all work on `stream` and synchronises it
 * before returning (sizes come back to the host).  d_in must be readable for in_len bytes;
 * d_out writable for out_cap bytes. */
int32_t ahip_gzip_decode_device(const void *d_in, size_t in_len, void *d_out, size_t out_cap,
                                size_t *out_len, void *stream);

/* One process, several GPUs, everything device-resident (SURVEY.md section 8b "one process drives all 8 GPUs", 8e): shard s
 * -- a run of WHOLE gzip members, cut by the caller (e.g. along the BGZF `BC` chain, balanced on compressed bytes) --
 * lives on HIP device devices[s] at d_in[s] (in_len[s] bytes) and is decoded into d_out[s] (out_cap[s] bytes, same
 * device) by that device's context; ahip_init_devices() must have selected the device (without it every shard has to
 * sit on the current device).  The loop being sharded is the reference's member loop, _gzip_decoder_web.dart:29-55.
 * out_len[s] = bytes shard s produced, status[s] (may be NULL) = its verdict.  The only exchange between the devices
 * is the size prefix-scan: an all-gather of one uint64 per device over RCCL (ncclAllGather; librccl is loaded at run
 * time) when the shards sit on distinct devices, host sums otherwise (or with AHIP_NO_RCCL=1);
 * offsets[s] = where shard s goes in the logical output (exclusive prefix sum), offsets[n_shards] = the total.
 * The concatenation of the shards at those offsets is what ahip_gzip_decode() returns for the whole stream whenever
 * no member refers back across a shard boundary (quirk q8: such a shard reports AHIP_RANGE and the caller falls back
 * to the one-device call).  Returns the worst shard verdict. */
int32_t ahip_gzip_decode_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, const size_t *in_len,
                                void *const *d_out, const size_t *out_cap, size_t *out_len, uint64_t *offsets,
                                int32_t *status);
/* Diagnostics: how the last sharded call exchanged the sizes (1 = RCCL all-gather, 0 = host sums). */
int32_t ahip_debug_last_exchange(void);

/* Deflate of ONE input cut into per-device shards (same contexts, same size exchange as ahip_gzip_decode_shards).
 * ref: codecs/zlib/deflate.dart:219 -- `_trStoredBlock(0, 0, false)`, the byte-aligning empty stored block the reference
 * itself emits as a flush marker, is what lets independently compressed pieces be spliced.  Shard s = d_in[s][0, in_len[s])
 * on devices[s] (the caller cuts the input anywhere; multiples of 32 KiB lose nothing) is compressed into d_out[s]; every
 * shard but the last ends with that marker instead of a final block, so the shards' outputs laid end to end at
 * offsets[] (exclusive prefix sum of out_len[], offsets[n_shards] = the total) are ONE raw DEFLATE stream of the
 * concatenated input, for this library's Inflate and for any other.  A match never reaches into another shard.
 * crc32s (may be NULL): CRC-32 of every shard's input, taken on its device -- a gzip trailer's CRC is combined from them.
 * level / window_bits as in ahip_deflate_raw (invalid values: the reference's silent no-op, every out_len[s] = 0). */
int32_t ahip_deflate_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, const size_t *in_len,
                            int32_t level, int32_t window_bits, void *const *d_out, const size_t *out_cap, size_t *out_len,
                            uint64_t *offsets, uint32_t *crc32s);

/* BZip2 blocks over per-device shards.  ref: codecs/bzip2_decoder.dart:20-88 decodes the blocks of ONE stream one after
 * the other; they are independent once their bit positions are known.  d_in[s] = a copy of the WHOLE compressed stream
 * (in_len bytes) on devices[s] -- blocks start at arbitrary bit positions and the compressed stream is the small side;
 * shard s decodes the blocks among candidates [K s / n, K (s + 1) / n) of the K block magics every device finds, into
 * d_out[s].  The shards are merged in stream order exactly like decodeStream: the first verdict that is not AHIP_OK, or
 * the end-of-stream block, ends the stream (shards behind it report out_len 0); the stream CRC is folded from the
 * shards' block CRCs (verify != 0: a mismatch is AHIP_FALSE).  offsets[] as above; status[s] (may be NULL) = shard s's
 * own verdict.  Returns the stream's verdict: what ahip_bzip2_decode_device returns for the same stream. */
int32_t ahip_bzip2_decode_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, size_t in_len,
                                 int32_t verify, void *const *d_out, const size_t *out_cap, size_t *out_len,
                                 uint64_t *offsets, int32_t *status);
/* The same two, one RANK at a time (one process per GPU; the caller exchanges and merges: archive_amd/sharding.py).
 * ahip_deflate_piece_device: this rank's piece of the input compressed on its own; every piece but the last (last = 0) ends
 * with the reference's flush marker (ref: deflate.dart:219) instead of a final block, so the pieces laid end to end in rank
 * order are ONE raw DEFLATE stream; *crc32 (may be NULL) = CRC-32 of the piece's input.
 * ahip_bzip2_decode_range_device: the blocks among the block-magic candidates [K rank / world, K (rank + 1) / world) of the
 * stream (d_in = the WHOLE compressed stream), or from candidate `from` on when from != ~0 (the merge's second try).
 * info[8] = {blocks folded, CRC fold of those blocks, met the end-of-stream block, its stored CRC, the stream ended inside
 * this range, the candidate the chain started at, the candidate it expects next, 0}; merged in rank order like decodeStream
 * (ref: bzip2_decoder.dart:20-88): a rank whose chain did not start where the ranks in front expect the next block runs again
 * with `from`; the first verdict that is not AHIP_OK or the end-of-stream block ends the stream; the stream CRC is
 * rotl-xor over the ranks' folds. */
int32_t ahip_deflate_piece_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, int32_t last, void *d_out,
                                  size_t out_cap, size_t *out_len, uint32_t *crc32, void *stream);
int32_t ahip_bzip2_decode_range_device(const void *d_in, size_t in_len, int32_t verify, uint32_t rank, uint32_t world, uint64_t from,
                                       void *d_out, size_t out_cap, size_t *out_len, uint64_t *info, void *stream);
/* Diagnostics: shards ahip_bzip2_decode_shards has decoded a second time, process-wide -- the chain of the shards in front
 * ended somewhere else than at the shard's first candidate (a false block magic inside a block's data on the boundary). */
int32_t ahip_debug_bz_reruns(void);
/* Diagnostics: chunks the calling thread's last long stream (one DEFLATE stream of >= 2 MiB: ahip_inflate_raw, a zlib
 * stream, a long gzip member) was decoded in by the many-waves path; 0 = it went to the one-wave path. */
int32_t ahip_debug_last_chunks(void);

/* Plan/run split of the same path: the plan holds the member index (payload offsets, output
 * offsets) in device memory, so repeated runs time only the decode (bench.py times run). */
typedef struct ahip_gzip_plan ahip_gzip_plan;
int32_t ahip_gzip_plan_create(const void *d_in, size_t in_len, void *stream, ahip_gzip_plan **plan);
/* number of gzip members, total decoded size, total compressed payload bytes */
int32_t ahip_gzip_plan_info(const ahip_gzip_plan *plan, uint64_t *members, uint64_t *out_bytes,
                            uint64_t *payload_bytes);
/* Decode every member of the plan into d_out (asynchronous on `stream`; no host sync). */
int32_t ahip_gzip_plan_run(ahip_gzip_plan *plan, void *d_out, size_t out_cap, void *stream);
/* After the stream is synchronised: per-run verdict (AHIP_OK / AHIP_FALSE / ...). */
int32_t ahip_gzip_plan_status(ahip_gzip_plan *plan, size_t *out_len);
void ahip_gzip_plan_destroy(ahip_gzip_plan *plan);
/* Diagnostics only: per-member results of the last run, 20 u32 words per member
 * {end_pos lo,hi, out_len lo,hi, status, blocks, windows, rounds, fallbacks, partial, cyc[8], tok_words lo,hi}
 * (cyc: per-phase shader-clock cycles / 16, filled by -DAHIP_PROFILE builds only). */
int32_t ahip_debug_plan_results(ahip_gzip_plan *plan, uint32_t *host_words, size_t max_members, size_t *n_members);

/* ---- ONE long DEFLATE stream decoded by several ranks (one process per GPU) ----
 * ref: zlib/inflate.dart:104-156 is one loop over the blocks of one stream and _gzip_decoder_web.dart:29-55 one loop over
 * members: a stream of ONE member offers the member loop nothing to shard ("replicas only until the
 * speculative path exists").  Here every rank holds the whole COMPRESSED stream in its device memory and decodes an equal
 * range of it: it finds the DEFLATE block starts behind its own cuts, sizes them, resolves the chunks that turn out to lie
 * on the true chain of blocks to 16-bit symbols (a byte, or "byte j of the 32 KiB in front of my range"), learns those
 * 32 KiB from the ranks in front of it and writes ITS slice of the output.  Three all-gathers carry everything that
 * crosses between ranks; the CALLER performs them (torch.distributed over RCCL, MPI, ...: archive_amd/sharding.py
 * ::ShardedStreamDecoder is the Python host), the library never talks to another rank:
 *
 *   ahip_stream_split_create      (host only) rank r of `world` takes the cuts [n r / world, n (r + 1) / world)
 *   ahip_stream_split_candidates  -> this rank's block starts (bit positions, ascending; rank 0's begin with data_off * 8)
 *        all-gather #1: the lists, concatenated in rank order (= stream order)
 *   ahip_stream_split_size        -> 4 words per OWN candidate {status, bytes produced, end position, blocks}
 *        all-gather #2: the results, concatenated in rank order (entry i belongs to candidate i of the gathered list)
 *   ahip_stream_split_chain       (host only) follows the chain of "ends where the next one starts" from the stream's first
 *                                 bit: where this rank's slice lies in the stream's output, how long it is, the total
 *   ahip_stream_split_resolve     -> this rank's window map in d_map (ahip_stream_split_map_bytes() bytes of device memory)
 *        all-gather #3: the maps, concatenated in rank order (64 KiB + a status word per rank)
 *   ahip_stream_split_finish      -> the rank's slice in d_out
 *
 * *handled = 0 (size, chain, finish): not a case for this path -- too short, too few block starts, a damaged stream, a chunk
 * that decodes differently with its true history.  Every rank reaches the same verdict at the same step (it follows from
 * gathered data); the caller then decodes the stream with ahip_inflate_raw / ahip_gzip_decode_device on one rank, which
 * has the reference's exact semantics for every malformed input.  The stream has an output of its own (nothing in front of
 * its first byte can be referenced: a gzip member that follows others in one output -- the reference's shared OutputStream --
 * is the single-device path's).  A handle belongs to the thread that created it. */
typedef struct ahip_stream_split ahip_stream_split;
int32_t ahip_stream_split_create(const void *d_in, size_t in_len, size_t data_off, uint32_t rank, uint32_t world, void *stream,
                                 ahip_stream_split **split);
/* AHIP_E_CAP with *n = the count when cap is too small (call again). */
int32_t ahip_stream_split_candidates(ahip_stream_split *split, uint64_t *cand, size_t cap, size_t *n);
int32_t ahip_stream_split_size(ahip_stream_split *split, const uint64_t *all_cand, size_t n_all, uint64_t *results, size_t cap_words,
                               int32_t *handled);
/* rank_off / rank_len: this rank's slice of the stream's output; total_out: the whole output; end_pos: the reference's
 * InputStream position behind the stream (inflate.dart:104-156). */
int32_t ahip_stream_split_chain(ahip_stream_split *split, const uint64_t *all_results, size_t n_all, int32_t *handled, uint64_t *rank_off,
                                uint64_t *rank_len, uint64_t *total_out, uint64_t *end_pos);
size_t ahip_stream_split_map_bytes(void);
int32_t ahip_stream_split_resolve(ahip_stream_split *split, void *d_map);
/* d_maps: world x ahip_stream_split_map_bytes() bytes of device memory, rank order.  d_out: rank_len bytes. */
int32_t ahip_stream_split_finish(ahip_stream_split *split, const void *d_maps, void *d_out, size_t out_cap, size_t *out_len, int32_t *handled);
void ahip_stream_split_destroy(ahip_stream_split *split);
/* The same decode with ONE process driving the devices (ahip_init_devices): shard s = rank s of n_shards, run by the context of
 * devices[s]; d_in[s] = the WHOLE compressed stream on that device (DEFLATE data from byte data_off on).  What the ranks of a job
 * all-gather is host memory here (the window maps go through the host: 64 KiB a shard each way).  d_out[s] / out_cap[s]: room for
 * shard s's slice -- slices are balanced on COMPRESSED bytes, allow for more than total / n; AHIP_E_CAP leaves the sizes needed in
 * out_len[].  offsets[s] (offsets[n] = the total): where slice s lies in the stream's output; *end_pos: the reference's stream
 * position behind the DEFLATE data.  *handled = 0: not a case for the chunked decode, nothing usable was written. */
int32_t ahip_inflate_stream_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, size_t in_len, size_t data_off,
                                   void *const *d_out, const size_t *out_cap, size_t *out_len, uint64_t *offsets, uint64_t *end_pos,
                                   int32_t *handled);
/* Diagnostics / tests (no device, no handle): the chain walk of ahip_stream_split_chain for a rank that owns candidates
 * [c0, c1) of n; out[0..5] = handled, slice offset, slice length, total, end position, the rank's chunks on the chain. */
int32_t ahip_debug_stream_split_chain(const uint64_t *cand, const uint64_t *results, size_t n, uint32_t c0, uint32_t c1, uint64_t *out);

/* ---- Deflate ----
 * ref: codecs/zlib/deflate.dart:39-48 `Deflate(bytes, level: L, windowBits: W).getBytes()`; level 0..9 as in
 * DeflateLevel (deflate.dart:10-18), window_bits 9..15 (matches reach at most 2^W - 262 bytes back, deflate.dart:105-124).
 * The stream is valid DEFLATE for any
 * inflater (and round-trips through ahip_inflate_raw); its SIZE is within the tolerance DESIGN.md
 * states of the reference's, its bytes are not the reference's (no reference test pins them).
 * An invalid level or window writes nothing and returns AHIP_OK, like the reference's silent _init.
 * *crc32 receives the CRC-32 of the input (the reference's Deflate.crc32). */
int32_t ahip_deflate_raw(const uint8_t *in, size_t in_len, int32_t level, int32_t window_bits, uint8_t *out,
                         size_t out_cap, size_t *out_len, uint32_t *crc32);
/* ref: codecs/zlib/_gzip_encoder_web.dart:27-100 (mtime supplied by the caller; the reference stamps "now") */
int32_t ahip_gzip_encode(const uint8_t *in, size_t in_len, int32_t level, int32_t window_bits, uint32_t mtime, uint8_t *out,
                         size_t out_cap, size_t *out_len);
/* ref: codecs/zlib/_zlib_encoder_web.dart:27-73 (CMF from window_bits, FLEVEL 0: `78 01` for 15; Adler-32 big-endian) */
int32_t ahip_zlib_encode(const uint8_t *in, size_t in_len, int32_t level, int32_t window_bits, uint8_t *out, size_t out_cap,
                         size_t *out_len);
/* device-resident forms of the two framed encoders (ref: _gzip_encoder_web.dart:27-100, _zlib_encoder_web.dart:27-73): the
 * same bytes as ahip_gzip_encode / ahip_zlib_encode, d_in / d_out on the current device (header, DEFLATE stream and
 * trailer are assembled in d_out; the trailer's CRC-32 / Adler-32 come from the device copy of the input);
 * synchronise `stream` before returning */
int32_t ahip_gzip_encode_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, uint32_t mtime, void *d_out,
                                size_t out_cap, size_t *out_len, void *stream);
int32_t ahip_zlib_encode_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, void *d_out, size_t out_cap,
                                size_t *out_len, void *stream);
/* device-resident form: d_in/d_out on the current device; synchronises `stream` before returning */
int32_t ahip_deflate_raw_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, void *d_out, size_t out_cap,
                                size_t *out_len, void *stream);
/* upper bound of the compressed size for in_len input bytes */
size_t ahip_deflate_bound(size_t in_len);

/* ---- many raw DEFLATE streams in one call: the ZIP entry path ----
 * ref: codecs/zip/zip_file.dart:184,229,232 -- each compressed entry is `Inflate(bytes, uncompressedSize).getBytes()`
 * on its own slice of the archive.  Entry i is the slice in[in_off[i], in_off[i] + in_size[i]) (the stream sees the
 * slice end as end of input, like the reference's InputStream over the sub-list) and is decoded exactly like
 * ahip_inflate_raw.  size_hint[i] = the directory's uncompressed size (the reference only uses it to size its
 * buffer; here it sizes entry i's output window: an entry that produces more gets status AHIP_E_CAP and can be
 * redone alone) or NULL to measure every entry first.  out_off/out_len/status: n_entries elements each, filled on
 * return; entry i's bytes are out[out_off[i], out_off[i] + out_len[i]).  Returns AHIP_OK when the batch ran (per
 * entry verdicts in status[]), AHIP_E_CAP with *out_total = required bytes when out_cap is too small. */
int32_t ahip_inflate_batch(const uint8_t *in, size_t in_len, uint32_t n_entries, const uint64_t *in_off,
                           const uint64_t *in_size, const uint64_t *size_hint, uint8_t *out, size_t out_cap,
                           uint64_t *out_off, uint64_t *out_len, int32_t *status, size_t *out_total);
/* device-resident form: d_in/d_out on the device, the entry tables on the host */
int32_t ahip_inflate_batch_device(const void *d_in, size_t in_len, uint32_t n_entries, const uint64_t *in_off,
                                  const uint64_t *in_size, const uint64_t *size_hint, void *d_out, size_t out_cap,
                                  uint64_t *out_off, uint64_t *out_len, int32_t *status, size_t *out_total, void *stream);

/* ---- BZip2 ----
 * ref: codecs/bzip2_decoder.dart:13-88 `BZip2Decoder().decodeBytes(data, verify: false)`: ONE bzip2
 * stream (the reference returns at the first end-of-stream block); blocks are decoded in parallel.
 * The obsolete randomised-block mode returns AHIP_E_UNSUPPORTED. */
int32_t ahip_bzip2_decode(const uint8_t *in, size_t in_len, int32_t verify, uint8_t *out, size_t out_cap,
                          size_t *out_len);
/* device-resident form (no reference counterpart): d_in/d_out on the current device.  Blocks are placed by the
 * host between two kernel phases, so the call synchronises; `stream` is reserved (default stream is used). */
int32_t ahip_bzip2_decode_device(const void *d_in, size_t in_len, int32_t verify, void *d_out, size_t out_cap,
                                 size_t *out_len, void *stream);

/* ---- checksums (ref: util/crc32.dart:6-27, util/adler32.dart:29-52), chainable ---- */
uint32_t ahip_crc32(const uint8_t *data, size_t len, uint32_t crc /* 0 to start */);
uint32_t ahip_adler32(const uint8_t *data, size_t len, uint32_t adler /* 1 to start */);
/* the same two on device-resident data (HIP kernels; used internally for the gzip/zlib trailers of the encoders
 * and for `verify: true` of the zlib decoder, _zlib_decoder_web.dart:90-99); synchronise `stream` */
int32_t ahip_crc32_device(const void *d_data, size_t len, uint32_t crc, uint32_t *out, void *stream);
int32_t ahip_adler32_device(const void *d_data, size_t len, uint32_t adler, uint32_t *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ARCHIVE_HIP_H */
