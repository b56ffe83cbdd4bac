// bzip2_chain.hpp -- the loop of BZip2Decoder.decodeStream (ref: lib/src/codecs/bzip2_decoder.dart:46-87) restated over
// per-block verdicts.  HOST ONLY, no HIP: archive_hip.hip drives it between the kernel phases of a batch of blocks, and
// tests/emu/bzip2_chain_emu.cc drives the very same functions on the CPU (per-block results from a CPU restatement of
// the reference's block loop or from the wave emulation of the device code) against a whole-stream CPU decoder.
//
// The reference reads block after block from ONE bit reader:
//   _readBlockType (:90-111)  six bytes, one at a time: a byte that matches neither magic is `false` AT ONCE (even when
//                             fewer than six bytes are left), a read past the end is a RangeError;
//   stored CRC, _readCompressed (:113-730): -1 is `false` -- and a block may have WRITTEN bytes before it fails (the run
//                             whose count byte lay beyond the block's data, :612-631): they stay in the output;
//   verify: a block whose CRC differs is `false` behind its bytes (:64-67); the end-of-stream block compares the folded
//                             CRC (:77-81) and ends the stream: whatever follows is not looked at.
// Here the blocks of a batch are decoded side by side, so what the chain sees per candidate magic is a BzResult, and an
// EARLIER block's CRC is only known after a later block's verdict: a verdict that ends the stream is therefore kept
// pending (`verdict`) until the CRCs of the blocks in front of it have been looked at (bz_chain_crcs) -- with `verify`, a
// CRC mismatch in front of a RangeError is what the reference reports.
#pragma once
#include <cstdint>
#include <vector>

namespace ahip {

constexpr u32 BZ_PL_PARALLEL = 0;  // bytes by bz_rle_expand, CRC by bz_block_crc (res2)
constexpr u32 BZ_PL_SERIAL = 1;    // bytes by the direct pass of bz_unbwt, CRC from its counting pass (res)
constexpr u32 BZ_PL_PARTIAL = 2;   // like SERIAL, but the block FAILED behind these bytes: they count, nothing is folded

struct BzPlaced { u32 cand; u64 off, len; u32 how; };  // cand: index inside the batch

struct BzChain {
  u64 total = 0;         // bytes placed so far (what a too-small buffer is told)
  u64 keep = 0;          // bytes that count when a CRC stopped the stream
  int32_t verdict = 0;   // AHIP_OK / AHIP_FALSE / AHIP_RANGE / AHIP_E_UNSUPPORTED: what ended the stream
  bool saw_eos = false, stopped = false, crc_stop = false;
  u32 eos_stored = 0, combined = 0;
  size_t next = 0;       // candidate the chain expects next
  u64 folded = 0;        // block CRCs folded into `combined`
  u64 end_bit = 32;      // where the bit reader stands behind the last block (or end-of-stream marker) the chain has read
  size_t fail_cand = ~(size_t)0, fail_c0 = 0;  // the candidate whose own failure ended the stream (and its batch's first): its
                         // result's end_bit is where the header stopped or the block ended -- a failure inside the symbol loop
                         // is located by the caller (bz_fail_cursor) and put into fail_bit
  u64 fail_bit = 0;      // ... and where it stood when decodeStream returned false (bz_chain_finish: the reader has pulled
                         // ceil(fail_bit / 8) bytes from its InputStream, bz2_bit_reader.dart:12-44)
};

// What _readBlockType makes of bit position `bit` when NO magic starts there.  `bytes` = the stream's bytes from bit >> 3
// on (at least 7, zeros beyond the end).  *stop_bit: where the reader stands when it returns -1 (behind the first byte
// that fits neither magic).
inline int32_t bz_no_magic_verdict(u64 bit, u64 in_len, const u8 *bytes, u64 *stop_bit = nullptr) {
  static const u8 cm[6] = {0x31, 0x41, 0x59, 0x26, 0x53, 0x59}, em[6] = {0x17, 0x72, 0x45, 0x38, 0x50, 0x90};
  const u32 sh = (u32)(bit & 7);
  bool eos = true, comp = true;
  for (u32 i = 0; i < 6; ++i) {
    if (bit + 8 * (u64)(i + 1) > in_len * 8) return 2;  // AHIP_RANGE
    const u32 b = (((u32)bytes[i] << 8 | bytes[i + 1]) >> (8 - sh)) & 0xff;
    if (b != cm[i]) comp = false;
    if (b != em[i]) eos = false;
    if (!eos && !comp) { if (stop_bit) *stop_bit = bit + 8 * (u64)(i + 1); return 1; }  // AHIP_FALSE
  }
  if (stop_bit) *stop_bit = bit + 48;
  return 1;  // (a magic after all: the scan lists every one, so this is not reached)
}

// One batch [c0, c0 + nb) of the candidate list: follow the chain through it.  res[i] = verdict of candidate c0 + i after
// the counting passes (status OK = decoded by the parallel path, OVERFLOW = by the serial path, FALSE with out_len > 0 =
// by the serial path, failing behind its bytes).  peek(bit, dst): the stream's 7 bytes from bit >> 3 on.
template <class Peek>
inline void bz_chain_walk(BzChain &ch, const BzCand *cands, size_t ncand, size_t c0, u32 nb, const BzResult *res, u64 in_len,
                          Peek &&peek, std::vector<BzPlaced> &placed) {
  placed.clear();
  while (!ch.stopped && ch.next < c0 + nb) {
    const size_t i = ch.next;
    const BzResult &r = res[i - c0];
    if (cands[i].kind == 2) {  // end of stream: combined CRC, then decodeStream returns true
      if (r.status == BZ_ST_RANGE) ch.verdict = 2;
      else { ch.saw_eos = true; ch.eos_stored = r.stored_crc; ch.end_bit = r.end_bit; }
      ch.stopped = true;
      break;
    }
    if (r.status == BZ_ST_RANGE) { ch.verdict = 2; ch.stopped = true; break; }
    if (r.status == BZ_ST_UNSUPPORTED) { ch.verdict = -3; ch.stopped = true; break; }
    if (r.status == BZ_ST_FALSE && r.out_len > 0) {  // _readCompressed wrote, then returned -1
      placed.push_back({(u32)(i - c0), ch.total, r.out_len, BZ_PL_PARTIAL});
      ch.total += r.out_len;
      ch.verdict = 1; ch.stopped = true; ch.fail_bit = r.end_bit;  // (it failed in the inverse transform: behind the block)
      break;
    }
    if (r.status != BZ_ST_OK && r.status != BZ_ST_OVERFLOW) { ch.verdict = 1; ch.stopped = true; ch.fail_bit = r.end_bit; ch.fail_cand = i; ch.fail_c0 = c0; break; }
    placed.push_back({(u32)(i - c0), ch.total, r.out_len, r.status == BZ_ST_OK ? BZ_PL_PARALLEL : BZ_PL_SERIAL});
    ch.total += r.out_len;
    ch.end_bit = r.end_bit;
    // the next block type is read at r.end_bit
    if ((r.end_bit + 7) / 8 >= in_len) { ch.stopped = true; break; }  // while (!input.isEOS): clean end without an end-of-stream block
    size_t j = i + 1;
    while (j < ncand && cands[j].bit < r.end_bit) ++j;  // (false magics inside the block's data are stepped over)
    if (j >= ncand || cands[j].bit != r.end_bit) {
      u8 b7[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      peek(r.end_bit, b7);
      ch.verdict = bz_no_magic_verdict(r.end_bit, in_len, b7, &ch.fail_bit);
      ch.stopped = true;
      break;
    }
    ch.next = j;
  }
}

// The blocks of `placed` have their bytes (and the parallel ones their CRCs, res2): CRCs in stream order.
inline void bz_chain_crcs(BzChain &ch, const std::vector<BzPlaced> &placed, const BzResult *res, const BzResult *res2, int32_t verify) {
  for (const BzPlaced &pl : placed) {
    if (pl.how == BZ_PL_PARTIAL) { ch.keep = pl.off + pl.len; break; }  // (`false`: pending in ch.verdict already)
    const u32 crc = pl.how == BZ_PL_PARALLEL ? (res2[pl.cand].crc ^ 0xffffffffu) : res[pl.cand].crc;
    if (verify && crc != res[pl.cand].stored_crc) {  // the block's bytes were already written
      ch.verdict = 1; ch.keep = pl.off + pl.len; ch.saw_eos = false; ch.crc_stop = true; ch.stopped = true;
      ch.fail_bit = res[pl.cand].end_bit;  // (the CRC is compared when _readCompressed has returned: behind the block)
      break;
    }
    ch.combined = ((ch.combined << 1) | (ch.combined >> 31)) ^ crc;
    ++ch.folded;
    ch.keep = pl.off + pl.len;
  }
}

// The whole loop over the candidates [c_lo, c_hi) in batches of `batch`, the way bzip2_device_impl runs it:
//   decode(c0, nb, res)                 the counting passes of candidates [c0, c0 + nb) -> res[0 .. nb)  (0, or an error code < 0)
//   place(c0, nb, placed, res, res2)    the bytes of the placed blocks to their offsets, the parallel ones' CRCs -> res2
// Nothing behind the point where the chain stops is decoded.  A total beyond out_cap: the walk goes on (the caller is
// told the full size), nothing more is placed -> *over_cap.
template <class Decode, class Place, class Peek>
inline int32_t bz_chain_run(BzChain &ch, const BzCand *cands, size_t ncand, size_t c_lo, size_t c_hi, u32 batch, u64 in_len, int32_t verify,
                            u64 out_cap, bool *over_cap, Decode &&decode, Place &&place, Peek &&peek) {
  ch.next = c_lo;
  *over_cap = false;
  std::vector<BzResult> res, res2;
  std::vector<BzPlaced> placed;
  for (size_t c0 = c_lo; c0 < c_hi && !ch.stopped && ch.next < c_hi; c0 += batch) {
    if (ch.next >= c0 + batch) continue;  // the chain has already stepped over this whole batch (false magics inside data)
    const u32 nb = (u32)(c_hi - c0 < batch ? c_hi - c0 : batch);
    res.assign(nb, BzResult{});
    int32_t rc = decode(c0, nb, res);
    if (rc < 0) return rc;
    // Sizes are known here; block CRCs only after the expansion, so the walk is done for placement first and the
    // verdict (first CRC mismatch stops the stream, its bytes already written) afterwards.
    bz_chain_walk(ch, cands, ncand, c0, nb, res.data(), in_len, peek, placed);
    if (ch.total > out_cap) *over_cap = true;
    // (a RangeError leaves no output: the bytes in front of it only matter for the CRCs `verify` looks at first)
    if (!placed.empty() && !*over_cap && !(ch.verdict == 2 && !verify)) {
      res2.assign(nb, BzResult{});
      rc = place(c0, nb, placed, res, res2);
      if (rc < 0) return rc;
      bz_chain_crcs(ch, placed, res.data(), res2.data(), verify);
    }
  }
  return 0;
}

// decodeStream's return value and the bytes that count, once the chain has stopped (or run out of candidates)
// *stop_bit: where the reference's bit reader stands then (true: behind the last block or marker it read; false: where the
// failing check stood)
inline int32_t bz_chain_finish(const BzChain &ch, int32_t verify, u64 *out_len, u64 *stop_bit = nullptr) {
  int32_t v = ch.verdict;
  u64 sb = v == 1 ? ch.fail_bit : ch.end_bit;
  if (v == 0 && ch.saw_eos && verify && ch.eos_stored != ch.combined) { v = 1; sb = ch.end_bit; }  // (behind the marker's CRC)
  if (out_len) *out_len = ch.crc_stop ? ch.keep : ch.total;
  if (stop_bit) *stop_bit = sb;
  return v;
}

}  // namespace ahip
