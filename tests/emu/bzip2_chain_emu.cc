// bzip2_chain_emu.cc -- the product's bzip2 host chain (archive_amd/csrc/bzip2_chain.hpp: decodeStream's loop over
// per-block verdicts, batch by batch, the way bzip2_device_impl drives it between its kernel phases) run on the CPU.
// Test infrastructure only.  What stands in for the GPU's per-block work:
//   mode 0  the oracle's block function (oracle/bzip2_oracle.c: orc_bzip2_block) on every candidate magic -- verdict, end
//           bit, size, CRC; blocks are labelled "parallel path" / "serial path" in turn so that both branches of the
//           placement and of the CRC step run;
//   mode 1  the DEVICE code of phase 1 on the 64-thread wave emulation (bz_decode_block_wave, bz_mtf_lanes_wave,
//           bz_mtf_scan_wave -- as tests/emu/bzip2_emu.cc glues them), T^-1 the plain way, then the serial inverse
//           transform bz_unbwt_block, the very function the bz_unbwt kernel wraps (counting pass, then the direct pass);
//           a block in which _getMtfVal fails (BZ_ST_NEG) goes through bz_block_exact_lane like on the device;
//   mode 2  the header by the device code on the wave emulation, then bz_block_exact_lane for EVERY block (fast: one
//           thread) -- the function that restates the reference's symbol loop, -1 symbols included, held against the oracle.
// The candidates are every bit position that holds a magic (bz_scan_magic's job), sorted; the result -- status and
// bytes -- is compared by tests/test_bzip2_chain.py with the oracle's whole-stream decoder.
//
//   g++ -std=c++17 -O2 -pthread -shared -fPIC -o libbzchain.so tests/emu/bzip2_chain_emu.cc oracle/bzip2_oracle.o
#define AHIP_HOST_EMU 1
#include "../../archive_amd/csrc/bzip2_kernels.hpp"
#include "../../archive_amd/csrc/bzip2_chain.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
using namespace ahip;

extern "C" int orc_bzip2_block(const uint8_t *in, size_t n, uint64_t bit, int level, uint8_t *out, size_t cap, uint64_t *end_bit,
                               size_t *out_len, uint32_t *crc_out, uint32_t *stored_out, int *kind_out);

namespace {
template <class F> void wave(F f) {
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l) th.emplace_back([=]() { wave_emu::lane = l; f(l); });
  for (auto &x : th) x.join();
}
u32 g_crc_table[256];
u32 g_exact_blocks = 0;  // blocks that took bz_block_exact_lane (mode 1)
void crc_init() {
  for (u32 i = 0; i < 256; ++i) {
    u32 c = i << 24;
    for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? ((c << 1) ^ 0x04c11db7u) : (c << 1);
    g_crc_table[i] = c;
  }
}
struct Block {            // what the "device" keeps of a candidate between the counting passes and the placement
  std::vector<u8> bytes;  // mode 0: the block's bytes
  std::vector<u32> tt;    // mode 1: the vector of the inverse transform
  u32 nblock = 0, orig_ptr = 0;
};

// mode 1: phase 1 by the device code on the wave emulation -> BzResult (+ tt when the block got that far)
BzResult device_block(const u8 *in, size_t n, const BzCand &c, u32 level, Block &blk, bool all_exact) {
  static BzLds LDS;
  static BzScanLds SLDS;
  static BzLaneLds LL;
  const u32 nmax = 100000u * level;
  std::vector<u16> syms(BZ_SYM_CAP);
  std::vector<u8> sel(BZ_MAX_SELECTORS + 64), list0(256), perms(BZ_CHUNKS * 256), lists(BZ_CHUNKS * 256);
  std::vector<BzChunk> chunks(BZ_CHUNKS);
  std::vector<u32> offs(BZ_CHUNKS);
  BzResult R{};
  static BzTables T;
  // (mode 2: the header only -- syms == nullptr stops in front of the symbol stream -- and bz_block_exact_lane for the rest)
  wave([&](int lane) { BzResult r; bz_decode_block_wave(LDS, in, n, c, all_exact ? (u16 *)nullptr : syms.data(), list0.data(), sel.data(), r, (u32)lane, &T); if (lane == 0) R = r; });
  if (c.kind != 0) return R;
  if (all_exact && R.status == BZ_ST_OK) R.status = BZ_ST_NEG;
  std::vector<u8> pperms((size_t)BZ_PARTS * 256), b8((size_t)nmax + 64, 0);
  std::vector<u32> pcounts(BZ_PARTS);
  if (!(R.nsyms == 0 && R.status != BZ_ST_OK)) {  // (bz_mtf_scan: nothing was decoded -- header trouble)
    for (u32 k = 0; k < BZ_CHUNKS; ++k)
      wave([&](int lane) { BzChunk r; bz_mtf_lanes_wave<false>(LL, syms.data(), R.nsyms, k, nmax, nullptr, 0, b8.data(), pperms.data() + (size_t)k * 64 * 256, pcounts.data() + k * 64, perms.data() + k * 256, r, (u32)lane); if (lane == 0) chunks[k] = r; });
    wave([&](int lane) { BzResult r = R; bz_mtf_scan_wave(SLDS, r, nmax, chunks.data(), perms.data(), list0.data(), lists.data(), offs.data(), (u32)lane); wave_emu::barrier(); if (lane == 0) R = r; });
    if (R.status == BZ_ST_OK)
      for (u32 k = 0; k < BZ_CHUNKS; ++k)
        wave([&](int lane) { BzChunk r; bz_mtf_lanes_wave<true>(LL, syms.data(), R.nsyms, k, nmax, lists.data() + k * 256, offs[k], b8.data(), pperms.data() + (size_t)k * 64 * 256, pcounts.data() + k * 64, nullptr, r, (u32)lane); });
  }
  if (R.status == BZ_ST_NEG) {  // bz_block_exact: a failing _getMtfVal inside the symbol loop -- the reference's own loop, one lane
    std::vector<u8> mtfa(4096);
    bz_block_exact_lane(T, sel.data(), list0.data(), in, n, nmax, mtfa.data(), b8.data(), R);
    ++g_exact_blocks;
  }
  // bz_fail_cursor (the product runs it for the one block the chain stops at): where the reference's reader stands when its
  // symbol loop gives up -- its own loop once more, nothing written; a header that failed knows its own bit
  auto fail_cursor = [&]() {
    if (R.status != BZ_ST_FALSE || T.sym_bit == 0) return;
    std::vector<u8> mtfa(4096);
    BzResult r2 = R;
    bz_block_exact_lane(T, sel.data(), list0.data(), in, n, nmax, mtfa.data(), (u8 *)nullptr, r2);
    R.end_bit = r2.end_bit;
  };
  if (R.status != BZ_ST_OK) { fail_cursor(); return R; }
  // T^-1 the plain way (bzip2_decoder.dart:406-439; bz_tinv_* on the device)
  blk.tt.assign((size_t)nmax + 64, 0);
  u32 cf[257] = {0};
  for (u32 i = 0; i < R.nblock; ++i) { blk.tt[i] = b8[i]; cf[b8[i] + 1]++; }
  for (int i = 1; i <= 256; ++i) cf[i] += cf[i - 1];
  for (u32 i = 0; i < R.nblock; ++i) { const u32 ch = blk.tt[i] & 0xff; blk.tt[cf[ch]++] |= i << 8; }
  blk.nblock = R.nblock; blk.orig_ptr = R.pad_orig_ptr;
  // the counting pass of bz_unbwt (no slab: a block with bytes reports BZ_ST_OVERFLOW and its size)
  u32 st, crc; u64 olen;
  bz_unbwt_block(blk.tt.data(), nmax, blk.nblock, blk.orig_ptr, (u8 *)nullptr, 0, g_crc_table, st, olen, crc);
  R.status = st; R.out_len = olen; R.crc = crc;
  if (R.out_len == 0) fail_cursor();
  return R;
}
}  // namespace

extern "C" u32 bzchain_exact_blocks() { return g_exact_blocks; }

// status: AHIP_OK 0 / AHIP_FALSE 1 / AHIP_RANGE 2 / AHIP_E_CAP -1 / AHIP_E_UNSUPPORTED -3; *out_len as ahip_bzip2_decode
// reports it.  batch: candidates per batch.  *blocks_seen (may be NULL): per-block decodes that ran (nothing behind the
// end of the chain must be touched beyond its batch).
// what ahip_last_consumed() reports after the call (the product's g_consumed, worked out the same way)
static u64 g_pos;
extern "C" size_t bzchain_last_consumed(void) { return (size_t)g_pos; }
extern "C" int bzchain_decode(const u8 *in_raw, size_t n, int verify, u32 batch, int mode, u8 *out, size_t cap, size_t *out_len, u32 *blocks_seen) {
  if (out_len) *out_len = 0;
  if (blocks_seen) *blocks_seen = 0;
  static bool once = (crc_init(), true);
  (void)once;
  // the header outcomes of bzip2_device_impl
  g_pos = 0;
  if (n < 4) {
    for (size_t i = 0; i < n && i < 3; ++i) if (in_raw[i] != "BZh"[i]) { g_pos = i + 1; return 1; }
    return 2;
  }
  for (size_t i = 0; i < 3; ++i) if (in_raw[i] != "BZh"[i]) { g_pos = i + 1; return 1; }
  const int level = (int)in_raw[3] - 0x30;
  g_pos = 4;
  if (level < 0 || level > 9) return 1;
  if (n == 4) return 0;
  if (level == 0) return 1;
  // device buffers are allocations: readable a little past the end
  std::vector<u8> padded(((n + 127) & ~(size_t)63) + 64, 0);
  memcpy(padded.data(), in_raw, n);
  const u8 *in = padded.data();
  // B0: every bit position that holds a magic (bz_scan_magic), in stream order
  std::vector<BzCand> cands;
  {
    for (u64 bit = 0; bit + 48 <= (u64)n * 8; ++bit) {
      u64 v = 0;
      for (size_t k = 0; k < 7; ++k) v = (v << 8) | in[(bit >> 3) + k];
      v = (v >> (8 - (bit & 7))) & 0xffffffffffffull;
      const u32 kind = v == 0x314159265359ull ? 0u : (v == 0x177245385090ull ? 2u : 9u);
      if (kind != 9u) cands.push_back({bit, kind, 0});
    }
  }
  const size_t ncand = cands.size();
  auto peek = [&](u64 bit, u8 *b7) { for (int k = 0; k < 7; ++k) b7[k] = (bit >> 3) + k < n ? in[(bit >> 3) + k] : 0; };
  if (ncand == 0 || cands[0].bit != 32) {
    u8 b7[8] = {0};
    peek(32, b7);
    u64 sb = 32;
    const int32_t v0 = bz_no_magic_verdict(32, n, b7, &sb);
    g_pos = std::min<u64>((u64)n, (sb + 7) / 8);
    return v0;
  }
  std::vector<Block> blocks;  // of the current batch
  u32 seen = 0;
  auto decode = [&](size_t c0, u32 nb, std::vector<BzResult> &res) -> int32_t {
    blocks.assign(nb, Block{});
    for (u32 i = 0; i < nb; ++i) {
      const BzCand &c = cands[c0 + i];
      ++seen;
      if (mode >= 1) { res[i] = device_block(in, n, c, (u32)level, blocks[i], mode == 2); continue; }
      BzResult r{};
      const size_t tmp_cap = (size_t)level * 100000 * 52 + 1024;  // (a run of 4 + 255 per five bytes at worst)
      std::unique_ptr<u8[]> tmp(new u8[tmp_cap]);
      size_t olen = 0; u32 crc = 0, stored = 0; int kind = -1; u64 end = 0;
      const int st = orc_bzip2_block(in_raw, n, c.bit, level, tmp.get(), tmp_cap, &end, &olen, &crc, &stored, &kind);
      if (st == -1) return -2;
      r.end_bit = end; r.out_len = olen; r.stored_crc = stored; r.crc = crc;
      if (c.kind == 2) r.status = st == 2 ? BZ_ST_RANGE : BZ_ST_OK;
      else if (st == 17) r.status = BZ_ST_UNSUPPORTED;
      else if (st == 2) r.status = BZ_ST_RANGE;
      else if (st == 1) r.status = BZ_ST_FALSE;  // (out_len > 0: it failed behind its bytes)
      else r.status = ((c0 + i) & 1) ? BZ_ST_OVERFLOW : BZ_ST_OK;  // "serial" / "parallel" path in turn
      if (r.status == BZ_ST_FALSE || r.status == BZ_ST_OK || r.status == BZ_ST_OVERFLOW) blocks[i].bytes.assign(tmp.get(), tmp.get() + olen);
      res[i] = r;
    }
    return 0;
  };
  auto place = [&](size_t, u32, const std::vector<BzPlaced> &placed, const std::vector<BzResult> &res, std::vector<BzResult> &res2) -> int32_t {
    for (const BzPlaced &pl : placed) {
      if (pl.off + pl.len > cap) return -2;  // (the chain does not place beyond the buffer)
      if (mode >= 1) {
        u32 st, crc; u64 olen;
        bz_unbwt_block(blocks[pl.cand].tt.data(), 100000u * (u32)level, blocks[pl.cand].nblock, blocks[pl.cand].orig_ptr, out + pl.off, ~0ull, g_crc_table, st, olen, crc);
        if (olen != pl.len) return -2;
      } else {
        if (blocks[pl.cand].bytes.size() != pl.len) return -2;
        memcpy(out + pl.off, blocks[pl.cand].bytes.data(), pl.len);
      }
      res2[pl.cand] = res[pl.cand];
      if (pl.how == BZ_PL_PARALLEL) res2[pl.cand].crc = res[pl.cand].crc ^ 0xffffffffu;  // (bz_block_crc leaves it unfinalised)
    }
    return 0;
  };
  BzChain ch;
  bool over_cap = false;
  const int32_t rc = bz_chain_run(ch, cands.data(), ncand, 0, ncand, batch ? batch : 1, (u64)n, verify, (u64)cap, &over_cap, decode, place, peek);
  if (blocks_seen) *blocks_seen = seen;
  if (rc < 0) return rc;
  if (over_cap) { if (out_len) *out_len = ch.total; return -1; }
  if (ch.verdict == -3) return -3;
  u64 got = 0, sb = 0;
  const int32_t v = bz_chain_finish(ch, verify, &got, &sb);
  if (out_len) *out_len = (size_t)got;
  g_pos = std::min<u64>((u64)n, (sb + 7) / 8);
  return v;
}
