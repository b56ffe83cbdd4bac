// bzip2_kernels.hpp -- bzip2 block decoder on gfx950: blocks in parallel, and every phase inside a block in parallel.
//
// Reference: /root/reference/lib/src/codecs/bzip2_decoder.dart (libbzip2's decompress.c in Dart).
// The reference walks blocks one after another; blocks are independent once their bit positions are
// known, so:
//   B0 bz_scan_magic   every bit position of the stream is tested for the 48-bit block magic
//                      0x314159265359 / end-of-stream magic 0x177245385090 (_readBlockType :90-111).
//   B1 per candidate block:
//        phase 1  header, selectors, code lengths, limit/base/perm tables (:114-246, :774-813): bz_header, one wave;
//                 the Huffman codes (:267-388, the part that reads bits): bz_jump_tiles (where a decoder stands 50
//                 codes after any bit position, by doubling), bz_group_starts (one look-up per group of 50 codes),
//                 bz_decode_groups (a thread per group) -> a stream of 16-bit symbols; irregular blocks through
//                 bz_decode_block, one serial wave (256 positions looked up per step, the chain on scalar registers);
//                 what the symbols mean -- move-to-front list, RUNA/RUNB zero runs -- into the block's bytes: bz_mtf_lanes
//                 (64 chunks of 64 parts per block, a part per lane: the list after a stretch is a permutation of the list
//                 before it) twice, with
//                 bz_mtf_scan chaining the permutations in between;
//        phase 2  T^-1 (:406-439) as a stable counting sort over 64 waves (bz_tinv_hist / _cursors / _scatter) -- same
//                 result as the reference's serial loop;
//        phase 3  inverse BWT (:610-727).  The pointer chase tt[t] -> t is one cycle through the block;
//                 walked serially every step waits for the previous load (~1 us x 900 000).  Instead
//                 it is list-ranked: every 16th index (and the chain head) is a splitter; the sublists
//                 from splitter to splitter are walked counting steps (bz_walk<false>: persistent
//                 workgroups, a block's vector read through ONE XCD's L2), ordered along the cycle in
//                 two levels (bz_rank), and walked again writing their bytes at their place (bz_walk<true>).
//        phase 4  run-length undo + MSB-first CRC-32: 1 024 spans per block, each span's
//                 effect on the 5 possible entry states of the "4 equal bytes, then a count" machine
//                 (bz_rle_scan), composed in order; then every span expands at its own offset
//                 (bz_rle_expand) and the block's CRC is taken over the finished bytes (bz_block_crc: the
//                 reflected CRC kernel's wave sums over bit-reversed bytes, mirrored back).
//   The host follows the chain of blocks (a block's end bit must be the next block's magic) between
//   phases 4a and 4b to place blocks, and verifies CRCs when asked.  Blocks the parallel path cannot
//   represent exactly (pointer cycle shorter than the block, data ending inside a run-length escape:
//   corrupt input only) go through the serial bz_unbwt, which restates the reference loop 1:1.
// The obsolete randomised-block mode is not implemented (BZ_ST_UNSUPPORTED).
#pragma once
#include "common.hpp"
#ifndef AHIP_HOST_EMU
#include "checksum_kernels.hpp"  // the block CRCs are taken over the finished output (bz_block_crc)
#endif

namespace ahip {

constexpr u32 BZ_MAX_SELECTORS = 18002;
constexpr u32 BZ_ST_OK = 0, BZ_ST_FALSE = 1, BZ_ST_RANGE = 2, BZ_ST_OVERFLOW = 16, BZ_ST_UNSUPPORTED = 17, BZ_ST_SERIAL = 18;
constexpr u32 BZ_ST_HUFF_SERIAL = 19;  // between kernels only: the position-parallel Huffman pass hands the block to the serial one
constexpr u32 BZ_ST_NEG = 20;          // between kernels only: _getMtfVal returned -1 inside the symbol loop -> bz_block_exact
constexpr u32 BZ_G = 16;       // splitter stride of the list ranking (see bz_walk: short sublists keep a block's walk inside one L2)
constexpr u32 BZ_SPANS = 1024;  // run-length spans per block
#ifndef AHIP_BZ_HANDOUT
#define AHIP_BZ_HANDOUT 16
#endif
constexpr u32 BZ_WALK_HANDOUT = AHIP_BZ_HANDOUT;  // idle lanes of a wave that make it worth handing out sublists
constexpr u32 BZ_WALK_BATCH = 128;  // sublists a wave takes from its XCD's queue at a time

struct BzCand { u64 bit; u32 kind; u32 pad; };  // kind 0 = compressed block, 2 = end of stream
struct BzResult {
  u64 end_bit;   // bit position just after the block
  u64 out_len;
  u32 status, crc, stored_crc, nblock;
  u32 pad_orig_ptr, nsyms;  // origPtr (phases 2-4 need it); symbols the Huffman pass recorded
};

// ---- B0: magic scan ----
#ifndef AHIP_HOST_EMU
__global__ __launch_bounds__(256) void bz_scan_magic(const u8 *__restrict__ in, u64 n, BzCand *cands, u32 *count, u32 cap) {
  const u64 p = (u64)blockIdx.x * 256 + threadIdx.x;  // byte offset
  if (p + 6 > n) return;
  u64 w = 0;
  for (int k = 0; k < 8; ++k) w = (w << 8) | (p + k < n ? in[p + k] : 0);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (p + 6 + (s ? 1 : 0) > n) break;
    const u64 v = (w >> (16 - s)) & 0xffffffffffffull;
    const u32 kind = v == 0x314159265359ull ? 0u : (v == 0x177245385090ull ? 2u : 9u);
    if (kind != 9u) {
      const u32 i = atomicAdd(count, 1u);
      if (i < cap) { cands[i].bit = p * 8 + s; cands[i].kind = kind; cands[i].pad = 0; }
    }
  }
}

#endif

// ---- MSB-first bit reader over global memory (Bz2BitReader) ----
// 64-bit left-aligned buffer refilled 32 bits at a time: one global load per ~4 symbols instead of
// one (or more) per field.  `bit` is always the position of the next unread bit.
struct BzBits {
  const u8 *in; u64 n;
  u64 bit;     // next bit to read
  bool fault;  // read past the end: RangeError in the reference
  u64 buf;     // unread bits, MSB first
  u32 cnt;     // valid bits in buf
  u64 fetch;   // byte offset of the next refill
};
AHIP_DEVINL void bz_seek(BzBits &b, u64 bit) {
  b.bit = bit; b.buf = 0; b.cnt = 0; b.fetch = bit >> 3; b.fault = false;
  // prime with the partial first byte
  const u32 sh = (u32)bit & 7;
  if (sh) {
    const u64 v = b.fetch < b.n ? b.in[b.fetch] : 0;
    b.buf = (v << (56 + sh));
    b.cnt = 8 - sh;
    b.fetch += 1;
  }
}
AHIP_DEVINL u32 bz_bits(BzBits &b, u32 nb) {  // nb <= 24
  if (nb == 0) return 0;
  if (b.bit + nb > b.n * 8) { b.fault = true; b.bit += nb; return 0; }
  if (b.cnt < nb) {
    u32 w = 0;
    if (b.fetch + 4 <= b.n) w = __builtin_bswap32(load_u32_unaligned(b.in + b.fetch));
    else for (int k = 0; k < 4; ++k) w = (w << 8) | (b.fetch + k < b.n ? b.in[b.fetch + k] : 0);
    b.buf |= (u64)w << (32 - b.cnt);
    b.cnt += 32;
    b.fetch += 4;
  }
  const u32 v = (u32)(b.buf >> (64 - nb));
  b.buf <<= nb;
  b.cnt -= nb;
  b.bit += nb;
  return v;
}

constexpr u32 BZ_FAST_BITS = 10;
constexpr u32 BZ_RING = 512;
constexpr u32 BZ_SYM_CAP = BZ_MAX_SELECTORS * 50 + 128;  // symbols of a block: 50 per selector
constexpr u32 BZ_CHUNKS = 64;                          // MTF chunks per block
constexpr u32 BZ_MISS_ENT = 1u | (1u << 9) | (1u << 18) | (1u << 31);  // chain entry without a code: the exact loop decides there (bit 31)
constexpr u32 BZ_WIN = 256;                            // bit positions looked up per window (4 per lane)
#ifdef AHIP_BZ_PROFILE
__device__ unsigned long long bz_prof[8];  // cycles: 0 header+selectors+lengths 1 tables 2 window setup 3 chain 4 cuts+stores 5 windows 6 symbols
#define BZ_TICK(v) const u64 v = __builtin_readcyclecounter()
#define BZ_ACC(i, a, b) prof[i] += (b) - (a)
#else
#define BZ_TICK(v) do { } while (0)
#define BZ_ACC(i, a, b) do { } while (0)
#endif
// What bz_header leaves in global memory for the position-parallel Huffman pass (bz_jump_tile / bz_walk_groups /
// bz_decode_groups below): the reference's limit / base / perm tables and the first-code table.
struct BzTables {
  i32 limit[6][24], base[6][24];
  u16 perm[6][258];
  i32 min_len[6];
  u16 e16[6][1u << BZ_FAST_BITS];  // first code of the 10-bit pattern: symbol << 5 | length; 0 = the exact loop decides
  u32 ngroups, nsel, eob, pad;
  u64 sym_bit;                     // where the block's symbol stream starts
};
struct BzLds {
  i32 limit[6][24], base[6][24];
  u16 perm[6][258];
  u8 len[6][258];
  i32 min_len[6];
  u8 seq2unseq[256];
  // What the reference's limit/base/perm loop does with the 10-bit pattern p, precomputed:
  //   chain[t][p]  bits 0-8 the bits ALL the codes take that lie completely inside the pattern, bits 9-17 the bits the
  //                first one takes, bits 18-27 a mask of the offsets those codes start at (bit 0 always);
  //                BZ_MISS_ENT = not even one code of <= 10 bits, an invalid index, or the end-of-block symbol: the
  //                exact loop decides there (an entry never continues across such a code either);
  //   sym[t][p]    the first code's symbol.
  u32 chain[6][1u << BZ_FAST_BITS];
  u16 sym[6][1u << BZ_FAST_BITS];
  u32 ring[BZ_RING];  // the stream ahead of the symbol loop, as big-endian dwords
};

// Wave-uniform bit reader for the symbol loop: the stream is fetched 64 dwords at a time (one per lane,
// coalesced, the next batch already in flight), words are pulled out with v_readlane, and the bit buffer
// itself lives in scalar registers -- no memory latency on the decode path.
struct BzFast {
  const u8 *in; u64 n;   // bytes
  u64 nbits, bit;        // stream length in bits, next unread bit
  u64 base;              // dword index held by lane 0 of `cur`
  u32 cur, nxt;          // this lane's dword (big-endian value) of the current / next batch
  u32 wi;                // next dword of `cur` to pull
  u64 buf; u32 cnt;      // unread bits, MSB first
  bool fault;
};
AHIP_DEVINL u32 bzf_word(const BzFast &f, u64 idx) {
  const u64 off = idx * 4;
  if (off + 4 <= f.n) return __builtin_bswap32(*(const u32 *)(f.in + off));  // `in` is device-allocated: aligned
  u32 w = 0;
  for (int k = 0; k < 4; ++k) w = (w << 8) | (off + k < f.n ? f.in[off + k] : 0u);
  return w;
}
AHIP_DEVINL void bzf_refill(BzFast &f, int lane) {  // afterwards cnt >= 33
  if (f.cnt > 32) return;
  const u32 w = lane_bcast(f.cur, (int)uniform(f.wi));
  f.buf |= (u64)w << (32 - f.cnt);
  f.cnt += 32;
  if (++f.wi == 64) {
    f.cur = f.nxt;
    f.base += 64;
    f.nxt = bzf_word(f, f.base + 64 + lane);
    f.wi = 0;
  }
}
AHIP_DEVINL void bzf_init(BzFast &f, const u8 *in, u64 n, u64 bit, int lane) {
  f.in = in; f.n = n; f.nbits = n * 8; f.bit = bit; f.fault = false;
  f.base = bit >> 5;
  f.cur = bzf_word(f, f.base + lane);
  f.nxt = bzf_word(f, f.base + 64 + lane);
  f.wi = 0; f.buf = 0; f.cnt = 0;
  bzf_refill(f, lane);
  const u32 skip = (u32)bit & 31;
  f.buf <<= skip; f.cnt -= skip;
}
AHIP_DEVINL u32 bzf_bits(BzFast &f, u32 nb, int lane) {  // nb <= 24; the reference's readBits
  if (nb == 0) return 0;
  if (f.bit + nb > f.nbits) { f.fault = true; f.bit += nb; return 0; }
  bzf_refill(f, lane);
  const u32 v = (u32)(f.buf >> (64 - nb));
  f.buf <<= nb; f.cnt -= nb; f.bit += nb;
  return v;
}

// The selectors of a block (bzip2_decoder.dart:150-184: nsel unary numbers j < ngroups, each an index into a move-to-front
// list of the tables) by the whole wave -- read one after the other they were ~ 1 ms of one wave per block, the longest
// serial step a 64-block decode had.  A zero bit ends a number: 64 dwords of the stream a round, every lane counts the
// zeros of its dword (prefix sum = which selectors they end) and knows where the zero before its first one lies (prefix
// maximum), so every number is a difference of positions.  The list is six entries, four bits each in a register: every
// lane runs its share of the numbers over the identity, the 64 results are composed in order, and every lane runs its
// share again from its true starting list.  Returns the bit behind the last selector -- or 0 when anything is out of the
// ordinary (a number >= ngroups, the input ending first): the caller's serial loop then finds the error exactly where
// the reference does.
AHIP_DEVINL u32 bz_mtf6_take(u32 &st, u32 j) {  // entry j of the packed list moves to the front
  const u32 v = (st >> (4 * j)) & 15u;
  const u32 low = st & ((1u << (4 * j)) - 1);
  st = (st & ~((1u << (4 * (j + 1))) - 1)) | (low << 4) | v;
  return v;
}
AHIP_DEVINL u32 bz_mtf6_compose(u32 r, u32 t) {  // the list r taken through t: entry k is r[t[k]]
  u32 o = 0;
#pragma unroll
  for (u32 k = 0; k < 6; ++k) o |= ((r >> (4 * ((t >> (4 * k)) & 15u))) & 15u) << (4 * k);
  return o;
}
AHIP_DEVINL u64 bz_selectors_wave(const u8 *__restrict__ in, u64 n, u64 bit0, u32 ngroups, u32 nsel, u8 *__restrict__ sel, const u32 lane) {
  const u64 nbits = n * 8;
  const u64 d0 = bit0 >> 5;                 // dword the region starts in
  const u32 skip = (u32)bit0 & 31;
  u32 count = 0;                            // zeros = selectors so far (wave-uniform)
  u32 prev1 = skip;                         // 1 + position (bits from dword d0) of the last zero so far; the virtual one in front of bit0
  u32 viol = 0, end_rel = 0;
  const u32 max_rounds = (nsel * ngroups + 31 + 2047) / 2048 + 1;
  for (u32 round = 0; round < max_rounds && count < nsel; ++round) {
    const u64 idx = d0 + (u64)round * 64 + lane;
    u32 w;
    {
      const u64 off = idx * 4;
      if (off + 4 <= n) w = __builtin_bswap32(*(const u32 *)(in + off));
      else { w = 0; for (int k = 0; k < 4; ++k) w = (w << 8) | (off + k < n ? (u32)in[off + k] : 0u); }
    }
    u32 x = ~w;                             // a one where the stream has a zero
    if (round == 0 && lane == 0 && skip) x &= 0xffffffffu >> skip;   // bits in front of the region are nobody's
    const u32 nz = (u32)__builtin_popcount(x);
    u32 total;
    const u32 base = count + wave_excl_sum(nz, total);
    const u32 rel0 = (round * 64 + lane) * 32;
    const u32 mylast1 = x ? rel0 + (31u - (u32)__builtin_ctz(x)) + 1u : 0u;  // 1 + position of my last zero
    const u32 inc = wave_incl_umax(mylast1);
    u32 before1 = lane_prev(inc);           // (lane 0: 0)
    before1 = before1 > prev1 ? before1 : prev1;
    u32 r = 0, p1 = before1;
    while (x) {
      const u32 k = (u32)__builtin_clz(x);
      x &= ~(0x80000000u >> k);
      const u32 pos = rel0 + k, j = pos + 1 - p1 - 1;  // ones between the zero before and this one
      p1 = pos + 1;
      const u32 i = base + r++;
      if (i < nsel) {
        if (j >= ngroups) viol = 1; else sel[i] = (u8)j;
        if (i == nsel - 1) end_rel = pos + 1;
      }
    }
    count += total;
    const u32 last = lane_bcast(inc, 63);
    prev1 = last > prev1 ? last : prev1;
  }
  const u32 end_all = wave_umax(end_rel);
  if (__any(viol != 0) || count < nsel || end_all == 0 || d0 * 32 + end_all > nbits) return 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the numbers -> every lane's loads
  wave_sync();
  const u32 per = (nsel + 63) / 64, lo = lane * per < nsel ? lane * per : nsel, hi = lo + per < nsel ? lo + per : nsel;
  u32 t = 0x543210u;
  for (u32 i = lo; i < hi; ++i) (void)bz_mtf6_take(t, sel[i]);
  u32 run = 0x543210u, mine = 0x543210u;
  for (u32 l = 0; l < 64; ++l) {
    if (lane == l) mine = run;
    run = bz_mtf6_compose(run, lane_bcast(t, (int)l));
  }
  for (u32 i = lo; i < hi; ++i) sel[i] = (u8)bz_mtf6_take(mine, sel[i]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the selectors -> every lane's loads
  wave_sync();
  return d0 * 32 + end_all;
}

// one wave per candidate block (a device function: tests/emu/bzip2_emu.cc runs it on the CPU wave emulation)
// syms: room for BZ_SYM_CAP symbols; list0: the block's initial MTF list (256 bytes, seqToUnseq applied)
// exp != nullptr: leave the tables there (BzTables); syms == nullptr: stop in front of the symbol stream
AHIP_DEVINL void bz_decode_block_wave(BzLds &L, const u8 *__restrict__ in, u64 n, const BzCand cand,
                                      u16 *__restrict__ syms, u8 *__restrict__ list0, u8 *__restrict__ sel, BzResult &out,
                                      const u32 lane, BzTables *__restrict__ exp = nullptr) {
  BzResult R{0, 0, BZ_ST_OK, 0, 0, 0, 0, 0};
  u32 nsyms = 0;
#ifdef AHIP_BZ_PROFILE
  u64 prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  BZ_TICK(t_begin);
  BzBits b{in, n, 0, false, 0, 0, 0};
  bz_seek(b, cand.bit + 48);
  u32 status = BZ_ST_OK;
  u32 orig_ptr = 0;
  if (exp && lane == 0) exp->sym_bit = 0;  // (nonzero only behind a header that held: bz_fail_cursor looks)
  if (cand.kind != 0) {  // end-of-stream marker: just the combined CRC
    u32 c = bz_bits(b, 16);
    c = (c << 16) | bz_bits(b, 16);
    R.stored_crc = c;
    R.end_bit = b.bit;
    R.status = b.fault ? BZ_ST_RANGE : BZ_ST_OK;
    out = R;
    return;
  }
  {
    u32 c = bz_bits(b, 16);
    c = (c << 16) | bz_bits(b, 16);
    R.stored_crc = c;
  }
  // ================= phase 1: header + MTF values =================
  do {
    const u32 randomized = bz_bits(b, 1);
    orig_ptr = bz_bits(b, 24);
    const u32 in_use16 = bz_bits(b, 16);
    u32 num_in_use = 0;
    for (u32 i = 0; i < 16; ++i) {
      u32 m = 0;
      if ((in_use16 >> (15 - i)) & 1) m = bz_bits(b, 16);
      for (u32 j = 0; j < 16; ++j)
        if ((m >> (15 - j)) & 1) { if (lane == 0) L.seq2unseq[num_in_use] = (u8)(i * 16 + j); num_in_use++; }
    }
    if (b.fault) { status = BZ_ST_RANGE; break; }
    if (randomized) { status = BZ_ST_UNSUPPORTED; break; }
    if (num_in_use == 0) { status = BZ_ST_FALSE; break; }
    const u32 alpha = num_in_use + 2;
    const u32 ngroups = bz_bits(b, 3);
    if (b.fault) { status = BZ_ST_RANGE; break; }  // (a read past the end is the reference's RangeError before any check of the value)
    if (ngroups < 2 || ngroups > 6) { status = BZ_ST_FALSE; break; }
    const u32 nsel = bz_bits(b, 15);
    if (b.fault) { status = BZ_ST_RANGE; break; }
    if (nsel < 1) { status = BZ_ST_FALSE; break; }
    if (nsel > BZ_MAX_SELECTORS) {
      // More selectors than the reference's Uint8List holds: it reads on, one unary number after the other, and the STORE of
      // number 18 002 is its RangeError -- unless a number in front of it is bad (`false`) or the input ends first
      // (RangeError as well, but where the reader stands).  Damaged input only: bit by bit.
      BzFast fo;
      bzf_init(fo, in, n, b.bit, lane);
      bool bad = false;
      for (u32 i = 0; i <= BZ_MAX_SELECTORS && !bad && !fo.fault; ++i) {
        u32 j = 0;
        while (bzf_bits(fo, 1, lane) && !fo.fault) { if (++j >= ngroups) { bad = true; break; } }
      }
      status = (bad && !fo.fault) ? BZ_ST_FALSE : BZ_ST_RANGE;
      b.bit = fo.bit;  // (where the reader stands: behind the bit that made the number too large)
      break;
    }
    // selectors and code lengths: ~30 000 bits read a few at a time -- from the stream held in registers (64 dwords a
    // lane-load, the next batch in flight), not one global load per 32 bits
    BzFast f;
    const u64 sel_end = bz_selectors_wave(in, n, b.bit, ngroups, nsel, sel, lane);
    bzf_init(f, in, n, sel_end ? sel_end : b.bit, lane);
    if (!sel_end) {  // out of the ordinary: one after the other, to the reference's exact stop
      u32 pos = 0x543210;  // MTF list of group numbers, 4 bits each
      bool bad = false;
      for (u32 i = 0; i < nsel; ++i) {
        bzf_refill(f, lane);
        const u32 top = (u32)(f.buf >> 32);
        if (!(top >> 31)) {  // zero bits: that many selectors in a row take the table at the front of the list
          u32 z = top ? (u32)__builtin_clz(top) : 32u;
          z = z < nsel - i ? z : nsel - i;
          if (f.bit + z > f.nbits) z = (u32)(f.nbits - f.bit);  // (what is left of the input; the next read is the RangeError)
          if (z) {
            if (lane < z) sel[i + lane] = (u8)(pos & 15);
            f.buf <<= z; f.cnt -= z; f.bit += z;
            i += z - 1;
            continue;
          }
        }
        const u32 j = top == 0xffffffffu ? 32u : (u32)__builtin_clz(~top);  // the unary number: ones up to a zero
        if (j >= ngroups) {  // the reference reads them one by one: the ngroups-th one is the error -- if the input lasts that long
          if (f.bit + ngroups > f.nbits) f.fault = true; else { bad = true; f.bit += ngroups; }  // (the reader stands behind that bit)
          break;
        }
        if (f.bit + j + 1 > f.nbits) { f.fault = true; break; }
        f.buf <<= j + 1; f.cnt -= j + 1; f.bit += j + 1;
        const u32 v = (pos >> (4 * j)) & 15;
        const u32 low = pos & ((1u << (4 * j)) - 1);
        pos = (pos & ~((1u << (4 * (j + 1))) - 1)) | (low << 4) | v;
        if (lane == 0) sel[i] = (u8)v;
      }
      if (f.fault) { status = BZ_ST_RANGE; break; }
      if (bad) { status = BZ_ST_FALSE; b.bit = f.bit; break; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // lane 0's selector stores -> every lane's loads
      wave_sync();
    }
    {
      bool bad = false;
      for (u32 t = 0; t < ngroups && !bad; ++t) {
        i32 c = (i32)bzf_bits(f, 5, lane);
        for (u32 i = 0; i < alpha; ++i) {
          for (;;) {
            if (c < 1 || c > 20) { bad = true; break; }
            if (!bzf_bits(f, 1, lane)) break;
            if (!bzf_bits(f, 1, lane)) c++; else c--;
            if (f.fault) { bad = true; break; }
          }
          if (bad) break;
          if (lane == 0) L.len[t][i] = (u8)c;
        }
      }
      if (f.fault) { status = BZ_ST_RANGE; break; }
      if (bad) { status = BZ_ST_FALSE; b.bit = f.bit; break; }  // (behind the last bit of the length that left 1 .. 20)
    }
    b.bit = f.bit;
    // (the reference's tables are fresh, zero-filled arrays for every block: a damaged code can index perm past its symbols)
    for (u32 i = lane; i < 6 * 258; i += 64) (&L.perm[0][0])[i] = 0;
    wave_sync();
    BZ_TICK(t_hdr);
    BZ_ACC(0, t_begin, t_hdr);
    // _hbCreateDecodeTables, one lane (tiny)
    if (lane == 0) {
      for (u32 t = 0; t < ngroups; ++t) {
        i32 minl = 32, maxl = 0;
        for (u32 i = 0; i < alpha; ++i) { const i32 l = L.len[t][i]; maxl = l > maxl ? l : maxl; minl = l < minl ? l : minl; }
        u32 pp = 0;
        for (i32 i = minl; i <= maxl; ++i)
          for (u32 j = 0; j < alpha; ++j)
            if (L.len[t][j] == i) L.perm[t][pp++] = (u16)j;
        i32 basev[24];
        for (int i = 0; i < 24; ++i) basev[i] = 0;
        for (u32 i = 0; i < alpha; ++i) basev[L.len[t][i] + 1]++;
        for (int i = 1; i < 23; ++i) basev[i] += basev[i - 1];
        for (int i = 0; i < 24; ++i) L.limit[t][i] = 0;
        i32 vec = 0;
        for (i32 i = minl; i <= maxl; ++i) { vec += basev[i + 1] - basev[i]; L.limit[t][i] = vec - 1; vec <<= 1; }
        for (i32 i = minl + 1; i <= maxl; ++i) basev[i] = ((L.limit[t][i - 1] + 1) << 1) - basev[i];
        for (int i = 0; i < 24; ++i) L.base[t][i] = basev[i];
        L.min_len[t] = minl;
      }
    }
    wave_sync();
    // fast tables: run the reference's decode loop on every 10-bit pattern once -- and again on what is left of the
    // pattern behind each code, as long as whole codes fit
    for (u32 t = 0; t < ngroups; ++t) {
      const i32 minl = L.min_len[t];
      for (u32 pat = lane; pat < (1u << BZ_FAST_BITS); pat += 64) {
        u32 used = 0, len1 = 0, starts = 0, sym1 = 0;
        for (;;) {
          i32 found = 0, fsym = 0;
          for (i32 zn = minl; used + (u32)zn <= BZ_FAST_BITS; ++zn) {
            const i32 zvec = (i32)(((pat << used) & ((1u << BZ_FAST_BITS) - 1)) >> (BZ_FAST_BITS - zn));
            if (zvec <= L.limit[t][zn]) {
              const i32 idx = zvec - L.base[t][zn];
              if (idx >= 0 && idx < 258) { found = zn; fsym = L.perm[t][idx]; }
              break;  // an index out of range is the reference's error: the exact loop reports it
            }
          }
          if (!found || (u32)fsym == num_in_use + 1) break;  // (the end-of-block symbol goes through the exact loop)
          starts |= 1u << used;
          if (used == 0) { len1 = (u32)found; sym1 = (u32)fsym; }
          used += (u32)found;
        }
        L.chain[t][pat] = used ? (used | (len1 << 9) | (starts << 18)) : BZ_MISS_ENT;
        L.sym[t][pat] = (u16)sym1;
      }
    }
    wave_sync();
    BZ_TICK(t_tab);
    BZ_ACC(1, t_hdr, t_tab);
    for (u32 i = lane; i < 256; i += 64) list0[i] = i < num_in_use ? L.seq2unseq[i] : (u8)0;
    if (exp) {
      for (u32 i = lane; i < 6 * 24; i += 64) { (&exp->limit[0][0])[i] = (&L.limit[0][0])[i]; (&exp->base[0][0])[i] = (&L.base[0][0])[i]; }
      for (u32 i = lane; i < 6 * 258; i += 64) (&exp->perm[0][0])[i] = (&L.perm[0][0])[i];
      for (u32 i = lane; i < 6 * (1u << BZ_FAST_BITS); i += 64) {
        const u32 e = (&L.chain[0][0])[i];
        (&exp->e16[0][0])[i] = (i >> BZ_FAST_BITS) < ngroups && !(e >> 31) ? (u16)(((u32)(&L.sym[0][0])[i] << 5) | ((e >> 9) & 31)) : (u16)0;
      }
      if (lane < 6) exp->min_len[lane] = L.min_len[lane];
      if (lane == 0) { exp->ngroups = ngroups; exp->nsel = nsel; exp->eob = num_in_use + 1; exp->pad = 0; exp->sym_bit = b.bit; }
      if (!syms) break;  // (bz_header: tables only; the serial wave leaves them for bz_block_exact and goes on)
    }
    // ---- the symbol loop ----
    // Only the Huffman side is serial here: a code's position is the end of the one before.  The 256 bit positions
    // from the current one are looked up at once (lane l: positions l, l + 64, l + 128, l + 192, ten bits each,
    // against the current group's table), and the chain through them -- position -> its entry -> the position
    // behind the codes the entry covers -- runs on scalar registers: a v_readlane, the entry's start mask shifted
    // into a 64-bit mask, an add (the loops below are all that is serial, about three instructions a symbol).
    // The marked positions are the symbols: they are cut at the end of the group of 50 (the next group has another
    // table) or at the end-of-block symbol and stored, compacted, as 16-bit values.  What the symbols MEAN -- the
    // move-to-front list, the zero runs -- is left to bz_mtf_*: chunks of the symbol stream, in parallel.
    // Codes longer than ten bits and the last bits of the input take the reference's bit-by-bit loop.
    const u64 nbits = n * 8;
    u64 bit = b.bit;
    auto stream_word = [&](u64 idx) -> u32 {  // dword idx of the stream as a big-endian value, zeros behind the end
      const u64 off = idx * 4;
      if (off + 4 <= n) return __builtin_bswap32(*(const u32 *)(in + off));  // `in` is device-allocated: aligned
      u32 w = 0;
      for (int k = 0; k < 4; ++k) w = (w << 8) | (off + k < n ? in[off + k] : 0u);
      return w;
    };
    constexpr u32 RM = BZ_RING - 1;
    u64 ring_hi = bit >> 5;  // dwords [ring_hi - BZ_RING, ring_hi) are in the ring; `nxt` is the batch after
    L.ring[(u32)(ring_hi + lane) & RM] = stream_word(ring_hi + lane);
    L.ring[(u32)(ring_hi + 64 + lane) & RM] = stream_word(ring_hi + 64 + lane);
    ring_hi += 128;
    u32 nxt = stream_word(ring_hi + lane);
    auto peek_global = [&](u64 at, u32 nb) -> u32 {  // 1 <= nb <= 24 bits at `at`, all inside the input
      const u64 by = at >> 3;
      u32 w = 0;
      for (int k = 0; k < 4; ++k) w = (w << 8) | (by + k < n ? (u32)in[by + k] : 0u);
      return uniform((w << ((u32)at & 7)) >> (32 - nb));
    };
    u32 selv = sel[lane < nsel ? lane : 0];  // selectors: 64 at a time, one per lane
    const u32 eob = num_in_use + 1;
    i32 group_no = -1;
    u32 group_pos = 0, gsel = 0;
    u32 stop = 0;  // 1 end of block, 2 bad data (false), 3 read past the end (RangeError)
    while (stop == 0) {
      if (group_pos == 0) {
        group_no++;
        if (group_no >= (i32)nsel) { stop = 2; break; }
        group_pos = 50;
        if (group_no && (group_no & 63) == 0) selv = sel[(u32)group_no + lane < nsel ? (u32)group_no + lane : 0];
        gsel = lane_bcast(selv, group_no & 63);
      }
      BZ_TICK(t_w0);
      while ((bit >> 5) + 10 > ring_hi) {
        L.ring[(u32)(ring_hi + lane) & RM] = nxt;
        ring_hi += 64;
        nxt = stream_word(ring_hi + lane);
      }
      wave_sync();
      u32 elen[4], epat[4], symv[4];
      u64 missm[4];
      {
        const u32 bo = ((u32)bit & 31) + lane, a = (u32)(bit >> 5) + (bo >> 5), sft = bo & 31;
        const bool near_end = bit + BZ_WIN + 32 > nbits;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32 w0 = L.ring[(a + 2 * r) & RM], w1 = L.ring[(a + 2 * r + 1) & RM];
          const u32 pt = (u32)(((((u64)w0 << 32) | w1) << sft) >> (64 - BZ_FAST_BITS));
          const u32 e = L.chain[gsel][pt];
          symv[r] = L.sym[gsel][pt];
          const u32 l1 = (e >> 9) & 511;
          bool ms = (e >> 31) != 0;
          // the last ten positions of a register step one code at a time (a start mask must not reach into the
          // next register's positions); so do the last bits of the input, where no code may end behind it
          const bool single = lane >= 54 || near_end;
          if (near_end && bit + lane + 64 * r + l1 > nbits) ms = true;
          elen[r] = single ? l1 : (e & 511);
          epat[r] = single ? 1u : ((e >> 18) & 1023);
          missm[r] = __ballot(ms);
        }
      }
      BZ_TICK(t_w1);
      BZ_ACC(2, t_w0, t_w1);
      // The chain.  A step: the entry at the current position -- its start mask into the visited mask, its length
      // onto the position.  An entry without a code steps on by one bit (what follows is rubbish, found out and cut
      // below).  Below position 44 of a register two steps need no check in between (2 x 10 bits stay inside).
      u32 p = 0, seen = 0;
      u64 m[4] = {0, 0, 0, 0};
      bool more = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (more) {
          u32 o = p - 64 * r;
          u64 mm = 0;
          while (o < 44) {
            const u32 l0 = lane_bcast(elen[r], (int)o), p0 = lane_bcast(epat[r], (int)o);
            mm |= (u64)p0 << o; o += l0;
            const u32 l1 = lane_bcast(elen[r], (int)o), p1 = lane_bcast(epat[r], (int)o);
            mm |= (u64)p1 << o; o += l1;
          }
          while (o < 64) { const u32 l0 = lane_bcast(elen[r], (int)o), p0 = lane_bcast(epat[r], (int)o); mm |= (u64)p0 << o; o += l0; }
          m[r] = mm;
          p = o + 64 * r;
          seen += (u32)__popcll(mm);
          if (seen > group_pos) more = false;  // the group ends inside what has been seen: the rest would be cut anyway
        }
      }
      BZ_TICK(t_w2);
      BZ_ACC(3, t_w1, t_w2);
      // a visited position without a code: it and everything behind it is not there; the exact loop takes over at it
      bool miss = false;
      if ((m[0] & missm[0]) | (m[1] & missm[1]) | (m[2] & missm[2]) | (m[3] & missm[3])) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (miss) m[r] = 0;
          else if (m[r] & missm[r]) {
            const u32 q = (u32)__builtin_ctzll(m[r] & missm[r]);
            m[r] &= (1ull << q) - 1;
            p = 64 * r + q;
            miss = true;
          }
        }
      }
      // the group ends with its 50th symbol: the next one belongs to the next table
      const u32 c0 = (u32)__popcll(m[0]), c1 = (u32)__popcll(m[1]), c2 = (u32)__popcll(m[2]), c3 = (u32)__popcll(m[3]);
      u32 take = c0 + c1 + c2 + c3;
      if (take + (miss ? 1u : 0u) > group_pos) {
        if (take > group_pos) {  // the symbol of rank group_pos is the first one that is not this group's
          const u32 rr = group_pos < c0 ? 0u : (group_pos < c0 + c1 ? 1u : (group_pos < c0 + c1 + c2 ? 2u : 3u));
          const u32 k = group_pos - (rr == 0 ? 0u : (rr == 1 ? c0 : (rr == 2 ? c0 + c1 : c0 + c1 + c2)));
          const u64 mm = rr == 0 ? m[0] : (rr == 1 ? m[1] : (rr == 2 ? m[2] : m[3]));
          const u32 q = (u32)__builtin_ctzll(__ballot(((mm >> lane) & 1) && wave_rank(mm) == k));
          const u64 keep = (1ull << q) - 1;
          if (rr == 0) { m[0] &= keep; m[1] = 0; m[2] = 0; m[3] = 0; }
          else if (rr == 1) { m[1] &= keep; m[2] = 0; m[3] = 0; }
          else if (rr == 2) { m[2] &= keep; m[3] = 0; }
          else m[3] &= keep;
          p = 64 * rr + q;
        }
        // (take == group_pos: the marked ones fill the group exactly and the position without a code is the next group's)
        take = group_pos;
        miss = false;
      }
      {
        u32 below = nsyms;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (m[r]) {
            if ((m[r] >> lane) & 1) syms[below + wave_rank(m[r])] = (u16)symv[r];
            below += (u32)__popcll(m[r]);
          }
        }
      }
      BZ_TICK(t_w3);
      BZ_ACC(4, t_w2, t_w3);
#ifdef AHIP_BZ_PROFILE
      prof[5] += 1;
#endif
      nsyms += take;
      group_pos -= take;
      if (miss) {  // the reference loop, bit by bit (long codes, invalid indices, end of input)
        const u64 at = bit + p;
        i32 zn = (i32)uniform((u32)L.min_len[gsel]);
        if (at + (u32)zn > nbits) { stop = 3; break; }
        i32 zvec = (i32)peek_global(at, (u32)zn);
        bool ok = true;
        for (;;) {
          if (zn > 20) { ok = false; break; }
          if (zvec <= (i32)uniform((u32)L.limit[gsel][zn])) break;
          if (at + (u32)zn + 1 > nbits) { stop = 3; break; }
          zvec = (zvec << 1) | (i32)peek_global(at + (u32)zn, 1);
          zn++;
        }
        if (stop) break;
        const i32 idx = ok ? zvec - (i32)uniform((u32)L.base[gsel][zn]) : -1;
        if (idx < 0 || idx >= 258) { stop = 2; break; }
        const u32 sym = uniform(L.perm[gsel][idx]);
        p += (u32)zn;
        group_pos--;
        if (sym == eob) stop = 1;
        else { if (lane == 0) syms[nsyms] = (u16)sym; nsyms++; }
      }
      bit += p;
    }
    b.bit = bit;
    b.fault = stop == 3;
    const bool bad = stop == 2;
    if (b.fault) { status = BZ_ST_RANGE; break; }
    if (bad) { status = BZ_ST_NEG; break; }  // a failing _getMtfVal: the reference does NOT stop there (bz_block_exact_lane)
  } while (0);

  // what the symbols mean, and everything behind that, are separate launches (bz_mtf_scan sets nblock and the final status)
#ifdef AHIP_BZ_PROFILE
  prof[6] = nsyms;
  if (lane == 0) for (int k = 0; k < 7; ++k) atomicAdd(&bz_prof[k], prof[k]);
#endif
  R.status = status;
  R.nblock = 0;
  R.nsyms = nsyms;
  R.end_bit = b.bit;
  R.crc = 0;
  R.pad_orig_ptr = orig_ptr;
  out = R;
}

// ---- the Huffman side, position-parallel ----
// bz_decode_block_wave above walks a block's codes one after the other: a code's position is the end of the one before
// it, and one wave spends ~45 cycles on each of the ~500 000 codes of a block.  Here the dependence is broken up:
//   bz_jump_tile     for EVERY bit position of the block (tiles of 4096, any number of workgroups) and every one of
//                    its coding tables: where does a decoder stand 50 codes after starting here?  The code length at
//                    every position is one table lookup; the 2-, 4-, ... 32-code jumps follow by doubling
//                    (J2k[i] = Jk[i] + Jk[i + Jk[i]], in LDS), J50 = J32 . J16 . J2.  A jump that would pass the
//                    end-of-block symbol, an invalid code or the end of the input is marked instead.
//   bz_walk_groups   one workgroup per block follows the groups: start of group g + 1 = start of g + J50[selector
//                    of g][start of g] -- one LDS lookup per 50 codes -- until it stands on a mark.
//   bz_decode_groups every group decodes its 50 codes from its start, one thread each; the marked group is decoded the
//                    same way and finds the end-of-block symbol (or the error) exactly where the reference does.
// The symbol stream that results is what bz_decode_block_wave writes; anything irregular (a mark with nothing behind it:
// a block running into its successor's bits) is handed to that function (BZ_ST_HUFF_SERIAL).
constexpr u32 BZ_TW = 4096, BZ_TOV = 1024, BZ_TN = BZ_TW + BZ_TOV;  // tile: positions written / overlap / computed
constexpr u16 BZ_TERM = 0xffff;
#ifdef AHIP_HOST_EMU
#define BZ_BLOCK_SYNC() do { } while (0)  /* the emulation runs these functions with one thread */
#else
#define BZ_BLOCK_SYNC() __syncthreads()
#endif
struct BzTileLds {
  u16 x[3][BZ_TN + 2];
  u32 bits[BZ_TN / 32 + 4];  // the tile's stream as big-endian dwords, from the dword its first bit lies in
  i32 limit[6][24], base[6][24];
  u16 perm[6][258];
  i32 min_len[6];
  u16 e16[6][1u << BZ_FAST_BITS];
};
AHIP_DEVINL u32 bz_stream_word(const u8 *__restrict__ in, u64 n, u64 idx) {  // dword idx as a big-endian value, zeros behind the end
  const u64 off = idx * 4;
  if (off + 4 <= n) return __builtin_bswap32(*(const u32 *)(in + off));  // `in` is device-allocated: aligned
  u32 w = 0;
  for (int k = 0; k < 4; ++k) w = (w << 8) | (off + k < n ? in[off + k] : 0u);
  return w;
}
// The length of the code at a position whose next 32 bits are w32 (table t), the way the reference finds it; 0 = no
// code there: none of <= 20 bits, an invalid index, the end-of-block symbol (sym_out says which when asked).
template <class TB>
AHIP_DEVINL u32 bz_code_at(const TB &T, u32 t, u32 w32, u32 eob, u32 &sym_out) {
  const u32 e = T.e16[t][w32 >> (32 - BZ_FAST_BITS)];
  if (e) { sym_out = e >> 5; return e & 31; }
  i32 zn = T.min_len[t];
  for (;;) {
    if (zn > 20) { sym_out = 0xffffu; return 0; }
    const i32 zvec = (i32)(w32 >> (32 - zn));
    if (zvec <= T.limit[t][zn]) {
      const i32 idx = zvec - T.base[t][zn];
      if (idx < 0 || idx >= 258) { sym_out = 0xffffu; return 0; }
      sym_out = T.perm[t][idx];
      return sym_out == eob ? 0u : (u32)zn;
    }
    zn++;
  }
}
// One tile: positions [tile_bit, tile_bit + BZ_TW) of the block whose tables are T; `lim` = the first bit that is not
// the block's any more (the next block's magic, or the end of the input).  j50[t * tstride + i] for position tile_bit + i.
AHIP_DEVINL void bz_jump_tile(BzTileLds &S, const u8 *__restrict__ in, u64 n, const BzTables *__restrict__ T, u64 tile_bit,
                              u64 lim, u16 *__restrict__ j50, u64 tstride, const u32 tid, const u32 nthreads) {
  const u32 ngroups = T->ngroups, eob = T->eob;
  for (u32 i = tid; i < 6 * 24; i += nthreads) { (&S.limit[0][0])[i] = (&T->limit[0][0])[i]; (&S.base[0][0])[i] = (&T->base[0][0])[i]; }
  for (u32 i = tid; i < 6 * 258; i += nthreads) (&S.perm[0][0])[i] = (&T->perm[0][0])[i];
  for (u32 i = tid; i < 6 * (1u << BZ_FAST_BITS) / 2; i += nthreads) ((u32 *)&S.e16[0][0])[i] = ((const u32 *)&T->e16[0][0])[i];
  for (u32 i = tid; i < 6; i += nthreads) S.min_len[i] = T->min_len[i];
  const u64 d0 = tile_bit >> 5;
  const u32 sft0 = (u32)tile_bit & 31;
  for (u32 i = tid; i < BZ_TN / 32 + 4; i += nthreads) S.bits[i] = bz_stream_word(in, n, d0 + i);
  BZ_BLOCK_SYNC();
  const u64 nbits = n * 8;
  // (a mark is the largest value: a sum with one in it saturates to it; x[.][BZ_TN] stays a mark, and every index is
  //  clamped to it -- no branches)
  for (u32 i = tid; i < 3; i += nthreads) S.x[i][BZ_TN] = BZ_TERM;
  // (Taking a thread's ten positions through each look-up together -- one LDS round trip per look-up and pass instead of one
  //  per position -- changed nothing, 2.92 against 2.82 ms for 64 blocks: the 24 waves per CU hide that latency already; the
  //  kernel is bound by the LDS's throughput on scattered 16-bit reads.)
  auto dbl = [&](const u16 *src, u16 *dst) {
    for (u32 i = tid; i < BZ_TN; i += nthreads) {
      const u32 a = src[i];
      const u32 j = i + a < BZ_TN ? i + a : BZ_TN;
      const u32 r = a + src[j];
      dst[i] = (u16)(r < BZ_TERM ? r : BZ_TERM);
    }
    BZ_BLOCK_SYNC();
  };
  for (u32 t = 0; t < ngroups; ++t) {
    for (u32 i = tid; i < BZ_TN; i += nthreads) {
      const u64 at = tile_bit + i;
      u32 r = BZ_TERM;
      if (at < lim) {
        const u32 bo = sft0 + i;
        const u32 w32 = (u32)(((((u64)S.bits[bo >> 5] << 32) | S.bits[(bo >> 5) + 1]) << (bo & 31)) >> 32);
        u32 sym;
        const u32 len = bz_code_at(S, t, w32, eob, sym);
        if (len && at + len <= nbits) r = len;
      }
      S.x[0][i] = (u16)r;
    }
    BZ_BLOCK_SYNC();
    dbl(S.x[0], S.x[1]);  // J2 (kept)
    dbl(S.x[1], S.x[2]);  // J4
    dbl(S.x[2], S.x[0]);  // J8
    dbl(S.x[0], S.x[2]);  // J16 (kept)
    dbl(S.x[2], S.x[0]);  // J32 (kept; J8 is not needed any more)
    for (u32 i = tid; i < BZ_TW; i += nthreads) {
      if (tile_bit + i >= lim) break;
      const u32 a = S.x[0][i];
      const u32 j = i + a < BZ_TN ? i + a : BZ_TN;
      const u32 ab = a + S.x[2][j];
      const u32 k = i + ab < BZ_TN ? i + ab : BZ_TN;
      const u32 r = ab + S.x[1][k];
      j50[t * tstride + i] = (u16)(r < BZ_TERM ? r : BZ_TERM);
    }
    BZ_BLOCK_SYNC();
  }
}

// One workgroup per block: the starts of its groups (bits from sym_bit).  jt = LDS room for 6 x BZ_TW jumps.
// Returns (thread 0) the number of groups that have a start: the last of them is the one standing on a mark, or -- no
// mark within nsel groups -- there is none and the block is `false` (the reference runs out of selectors).
struct BzWalkState { u32 pos_lo, pos_hi, g, done; };
struct alignas(16) BzWalkLds { u16 jt[2][6][BZ_TW + 8]; u8 sel[BZ_MAX_SELECTORS + 14]; BzWalkState st[2]; };  // two tiles, two states: see bz_walk_groups
AHIP_DEVINL void bz_walk_groups(BzWalkLds &S, const BzTables *__restrict__ T, const u8 *__restrict__ sel, u64 lim,
                                const u16 *__restrict__ j50, u64 j50_bit0, u64 tstride, u32 *__restrict__ gstart,
                                u32 &found, u32 &marked, const u32 tid, const u32 nthreads) {  // marked: see below
  const u32 ngroups = T->ngroups, nsel = T->nsel;
  const u64 sym_bit = T->sym_bit;
  for (u32 i = tid; i < nsel; i += nthreads) S.sel[i] = sel[i];
  if (tid == 0) { S.st[0].pos_lo = (u32)sym_bit; S.st[0].pos_hi = (u32)(sym_bit >> 32); S.st[0].g = 0; S.st[0].done = 0; }
  BZ_BLOCK_SYNC();
  // a tile's jumps: 6 rows of BZ_TW 16-bit values, fetched 8 at a time.  While thread 0 walks through tile t (in one half of
  // S.jt, reading the state the tile before left in S.st[t & 1] and leaving its own in the other), everybody stores the
  // jumps of tile t + 1 into the other half and asks for those of tile t + 3: ONE barrier a tile.  (With one buffer it was
  // three -- before the deposit, before the walk, behind it: 4 us a tile where the walk itself, ~ 16 dependent LDS
  // look-ups, is one.)  A jump is < 1024 bits: the walk never skips a tile.
  constexpr u32 VEC = 8, PER_ROW = BZ_TW / VEC, MAX_SLOTS = 6;  // 6 * 512 vectors over >= 512 threads
  uint4 holdA[MAX_SLOTS], holdB[MAX_SLOTS];
  auto request = [&](uint4 (&hold)[MAX_SLOTS], u64 base) {  // tile at `base` -> registers
#pragma unroll
    for (u32 k = 0; k < MAX_SLOTS; ++k) {
      const u32 v = tid + k * nthreads;
      hold[k] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
      if (v < 6 * PER_ROW && base < lim) {
        const u32 t = v / PER_ROW, i = (v % PER_ROW) * VEC;
        if (t < ngroups && base + i + VEC <= lim) hold[k] = load_u128_unaligned((const u8 *)(j50 + t * tstride + (base - j50_bit0) + i));
        else if (t < ngroups && base + i < lim) {
          u16 tmp[VEC];
          for (u32 q = 0; q < VEC; ++q) tmp[q] = base + i + q < lim ? j50[t * tstride + (base - j50_bit0) + i + q] : BZ_TERM;
          hold[k] = make_uint4(tmp[0] | (u32)tmp[1] << 16, tmp[2] | (u32)tmp[3] << 16, tmp[4] | (u32)tmp[5] << 16, tmp[6] | (u32)tmp[7] << 16);
        }
      }
    }
  };
  const bool wide = nthreads * MAX_SLOTS >= 6 * PER_ROW;  // (false: few threads -- the CPU emulation: plain copies)
  auto fill = [&](u32 half, const uint4 (&hold)[MAX_SLOTS], u64 tbase) {  // the tile at tbase -> S.jt[half]
    if (wide) {
#pragma unroll
      for (u32 k = 0; k < MAX_SLOTS; ++k) {
        const u32 v = tid + k * nthreads;
        if (v < 6 * PER_ROW) { const u32 t = v / PER_ROW, i = (v % PER_ROW) * VEC; *(uint4 *)&S.jt[half][t][i] = hold[k]; }
      }
    } else {
      for (u32 t = 0; t < ngroups; ++t)
        for (u32 i = tid; i < BZ_TW; i += nthreads) S.jt[half][t][i] = tbase + i < lim ? j50[t * tstride + (tbase - j50_bit0) + i] : BZ_TERM;
    }
  };
  u64 base = sym_bit;
  if (wide) { request(holdA, base); request(holdB, base + BZ_TW); }
  fill(0, holdA, base);
  if (wide) request(holdA, base + 2 * BZ_TW);
  BZ_BLOCK_SYNC();
  u32 last = 0;  // the state the walk ended in
  // tile number `par` mod 2, held in S.jt[par]; `hold` has the tile behind it; returns false when the walk is over
  auto tile = [&](u32 par, uint4 (&hold)[MAX_SLOTS]) -> bool {
    const BzWalkState in = S.st[par];
    const u64 pos = ((u64)in.pos_hi << 32) | in.pos_lo;
    last = par;
    if (in.done || pos >= lim) return false;
    if (tid == 0) {
      u32 o = (u32)(pos - base);  // position inside the tile
      const u32 rel = (u32)(base - sym_bit);
      u32 g = in.g, done = 0;
      u32 sg = g < nsel ? S.sel[g] : 0u;
      while (o < BZ_TW) {
        if (g >= nsel) { done = 2; break; }  // out of selectors
        const u32 j = S.jt[par][sg][o];
        sg = S.sel[g + 1];  // (room for one past the end) -- independent of the jump just requested
        gstart[g] = rel + o;
        ++g;
        if (j == BZ_TERM) { done = 1; break; }
        o += j;
      }
      const u64 q = base + o;
      BzWalkState out;
      out.pos_lo = (u32)q; out.pos_hi = (u32)(q >> 32); out.g = g; out.done = done;
      S.st[par ^ 1] = out;
    }
    fill(par ^ 1, hold, base + BZ_TW);
    if (wide) request(hold, base + 3 * BZ_TW);
    base += BZ_TW;
    BZ_BLOCK_SYNC();
    return true;
  };
  for (;;) {
    if (!tile(0, holdB)) break;
    if (!tile(1, holdA)) break;
  }
  const BzWalkState fin = S.st[last];
  found = fin.g;
  marked = fin.done == 1 ? 1u : (fin.done == 2 ? 0u : 2u);  // 1 a marked group ends the walk, 0 out of selectors, 2 irregular
}
struct BzGroupLds {
  i32 limit[6][24], base[6][24];
  u16 perm[6][258];
  i32 min_len[6];
  u16 e16[6][1u << BZ_FAST_BITS];
};

// Group g of a block (one thread): its <= 50 codes from gstart[g] into syms[50 g ...].  The marked group (last = true)
// stops where the reference stops and reports how: its symbols before that, the bit behind the end-of-block symbol.
struct BzGroupEnd { u32 status, nsyms; u64 end_bit; };
template <class TB>
AHIP_DEVINL void bz_decode_group(const TB &T, u32 eob, const u8 *__restrict__ in, u64 n, u64 sym_bit, u32 g, u32 start, u32 t,
                                 bool last, u16 *__restrict__ syms, BzGroupEnd &end) {
  const u64 nbits = n * 8;
  u64 pos = sym_bit + start;
  u32 k = 0;
  u32 status = BZ_ST_HUFF_SERIAL;  // (a marked group that turns out to have nothing in it: left to the serial decoder)
  for (; k < 50; ++k) {
    const u64 by = pos >> 3;
    u64 w = 0;
    if (by + 8 <= n) w = __builtin_bswap64(load_u64_unaligned(in + by));
    else for (int q = 0; q < 8; ++q) w = (w << 8) | (by + q < n ? (u64)in[by + q] : 0ull);
    const u32 w32 = (u32)((w << ((u32)pos & 7)) >> 32);
    u32 sym;
    u32 len = bz_code_at(T, t, w32, eob, sym);
    if (len && pos + len > nbits) len = 0;
    if (!len) {
      if (!last) break;  // (cannot happen: the jump over this group was not marked)
      // the reference's bit-by-bit loop: where exactly does it stop?
      i32 zn = T.min_len[t];
      if (pos + (u32)zn > nbits) { status = BZ_ST_RANGE; break; }
      bool ok = true, fault = false;
      for (;;) {
        if (zn > 20) { ok = false; break; }
        if ((i32)(w32 >> (32 - zn)) <= T.limit[t][zn]) break;
        if (pos + (u32)zn + 1 > nbits) { fault = true; break; }
        zn++;
      }
      if (fault) { status = BZ_ST_RANGE; break; }
      const i32 idx = ok ? (i32)(w32 >> (32 - zn)) - T.base[t][zn] : -1;
      if (idx < 0 || idx >= 258) { status = BZ_ST_NEG; break; }  // a failing _getMtfVal: bz_block_exact_lane goes on like the reference
      if ((u32)T.perm[t][idx] == eob) { status = BZ_ST_OK; pos += (u32)zn; break; }
      break;  // (cannot happen: a code after all)
    }
    syms[(u64)g * 50 + k] = (u16)sym;
    pos += len;
  }
  if (last) { end.status = status; end.nsyms = g * 50 + k; end.end_bit = pos; }
}

// ---- the symbol stream's meaning: move-to-front list and zero runs (bzip2_decoder.dart:267-388), by chunks ----
// The list after a stretch of symbols is a permutation of the list before it, whatever that was: every chunk first runs
// its symbols over the identity (pass 0: the permutation, and how many bytes the chunk makes), one wave per block
// chains the permutations (bz_mtf_scan: every chunk's true list and output offset, the block's size and verdict), and
// the chunks run again with the bytes (pass 1: tt[]).  A chunk starts where the even split says, moved behind the
// digits of a zero run that began before it: runs are never cut.
struct BzChunk { u32 count, bad; };

AHIP_DEVINL u32 bz_chunk_start(const u16 *__restrict__ syms, u32 nsyms, u32 c) {
  const u32 per = (nsyms + BZ_CHUNKS - 1) / BZ_CHUNKS;
  const u64 nominal = (u64)c * per;
  if (nominal >= nsyms) return nsyms;
  u32 st = (u32)nominal;
  if (st > 0 && syms[st - 1] <= 1)
    for (u32 k = 0; k < 32 && st < nsyms && syms[st] <= 1; ++k) ++st;  // (more than 21 digits: the run's owner reports it)
  return st;
}

// (The wave-per-symbol form below was the device's until round 4; it is compiled for the CPU emulation only, where
//  tests/emu/bzip2_emu.cc holds the part-per-lane form that replaced it against it, byte for byte.)
#ifdef AHIP_HOST_EMU
// list: 256 bytes the chunk starts from (pass 1) / nullptr = the identity (pass 0, result to perm_out)
template <bool WRITE>
AHIP_DEVINL void bz_mtf_chunk_wave(const u16 *__restrict__ syms, u32 nsyms, u32 c, u32 limit, const u8 *__restrict__ list,
                                   u32 out_off, u32 *__restrict__ tt, u8 *__restrict__ perm_out, BzChunk &res, const u32 lane) {
  const u32 s0 = uniform(bz_chunk_start(syms, nsyms, c)), s1 = uniform(c + 1 < BZ_CHUNKS ? bz_chunk_start(syms, nsyms, c + 1) : nsyms);
  u32 m0, m1, m2, m3;  // the list: entry 64 j + lane in m[j]
  if (WRITE) { m0 = list[lane]; m1 = list[lane + 64]; m2 = list[lane + 128]; m3 = list[lane + 192]; }
  else { m0 = lane; m1 = lane + 64; m2 = lane + 128; m3 = lane + 192; }
  auto mtf_take = [&](u32 nn) -> u32 {  // entry nn (< 256) moves to the front
    if (nn < 64) {
      const u32 v = lane_bcast(m0, (int)nn);
      u32 sh = lane_prev(m0);
      sh = lane == 0 ? v : sh;
      m0 = lane <= nn ? sh : m0;
      return v;
    }
    const u32 q = nn >> 6, r = nn & 63;
    const u32 v = q == 1 ? lane_bcast(m1, (int)r) : (q == 2 ? lane_bcast(m2, (int)r) : lane_bcast(m3, (int)r));
    const u32 c1 = lane_bcast(m0, 63), c2 = lane_bcast(m1, 63), c3 = lane_bcast(m2, 63);
    const u32 u0 = lane_prev(m0), u1 = lane_prev(m1), u2 = lane_prev(m2), u3 = lane_prev(m3);
    const u32 t0 = lane == 0 ? v : u0, t1 = lane == 0 ? c1 : u1, t2 = lane == 0 ? c2 : u2, t3 = lane == 0 ? c3 : u3;
    const bool low = lane <= r;
    m0 = t0;
    if (q == 1) { if (low) m1 = t1; }
    else if (q == 2) { m1 = t1; if (low) m2 = t2; }
    else { m1 = t1; m2 = t2; if (low) m3 = t3; }
    return v;
  };
  u32 cnt = 0;   // bytes made so far
  u32 pend = 0;  // tt[] stores are gathered 64 at a time: lane k keeps the byte for index 64 j + k
  auto flush_full = [&](u32 end_idx) {  // end_idx: a multiple of 64 just reached
    const u32 i = end_idx - 64 + lane;
    if (i >= out_off) tt[i] = pend;
  };
  auto put_run = [&](u32 v, u32 k) {
    while (k) {
      const u32 idx = out_off + cnt, off = idx & 63, take = k < 64 - off ? k : 64 - off;
      if (lane - off < take) pend = v;
      cnt += take; k -= take;
      if (((idx + take) & 63) == 0) flush_full(idx + take);
    }
  };
  u32 r0 = 0, es = 0, bad = 0;  // the zero run being read: digits so far, value so far
  for (u32 base = s0; base < s1 && !bad; base += 64) {
    const u32 here = s1 - base < 64 ? s1 - base : 64;
    const u32 v = lane < here ? (u32)syms[base + lane] : 0xffffu;
    const u64 dm = __ballot(v <= 1), bm = __ballot(v == 1);
    u32 k = 0;
    while (k < here) {
      if ((dm >> k) & 1) {  // RUNA / RUNB digits: the whole stretch at once
        const u64 rest = ~(dm >> k);
        const u32 r = rest ? (u32)__builtin_ctzll(rest) : 64u;
        if (r0 + r > 21) { bad = 1; break; }  // the 22nd digit finds N >= 2M
        const u32 B = (u32)(bm >> k) & ((1u << r) - 1);
        es += (((1u << r) - 1) + B) << r0;
        r0 += r;
        k += r;
        continue;
      }
      if (r0) {
        if (cnt + es > limit) { bad = 1; break; }
        if (WRITE) put_run(lane_bcast(m0, 0), es); else cnt += es;
        r0 = 0; es = 0;
      }
      // list symbols up to the next digit (or the end of the batch), in pieces that stay inside one 64-entry line of tt[]
      const u64 nd = dm >> k;
      u32 len = nd ? (u32)__builtin_ctzll(nd) : 64u;
      len = len < here - k ? len : here - k;
      if (cnt + len > limit) { bad = 1; break; }  // (some symbol of the stretch finds nblock >= nblockMAX)
      while (len) {
        const u32 idx = out_off + cnt, off = idx & 63;
        const u32 piece = WRITE ? (len < 64 - off ? len : 64 - off) : len;
        for (u32 q = 0; q < piece; ++q) {
          const u32 sym = lane_bcast(v, (int)(k + q));
          const u32 byte = mtf_take(sym - 1);
          if (WRITE) { if (lane == off + q) pend = byte; }
        }
        cnt += piece; k += piece; len -= piece;
        if (WRITE && ((idx + piece) & 63) == 0) flush_full(idx + piece);
      }
    }
  }
  if (r0 && !bad) {  // a run ends with its chunk
    if (cnt + es > limit) bad = 1;
    else if (WRITE) put_run(lane_bcast(m0, 0), es);
    else cnt += es;
  }
  if (WRITE) {
    const u32 end = out_off + cnt, i = (end & ~63u) + lane;
    if (i >= out_off && i < end) tt[i] = pend;
  } else {
    perm_out[lane] = (u8)m0; perm_out[lane + 64] = (u8)m1; perm_out[lane + 128] = (u8)m2; perm_out[lane + 192] = (u8)m3;
  }
  res.count = cnt;
  res.bad = bad;
}

#endif  // AHIP_HOST_EMU

// ---- the same, a PART per lane ----
// bz_mtf_chunk_wave spends a whole wave on one symbol at a time: the list lies across the lanes and a symbol's move to
// the front is a handful of cross-lane operations -- 403 M symbols x two passes x ~16 instructions were 19.5 of the 66 ms
// of a 448-block batch.  Here every LANE walks its own part of the chunk (a chunk = 64 parts, cut by the same rule as the
// chunks) with its own list: 256 bytes in LDS, the lane's dwords BZ_LANE_STRIDE apart from the next lane's (any one
// dword index is conflict-free across the lanes, and so is one lane's list across the dword indices).  A symbol's move
// to the front shifts the dwords in front of it by one byte; the wave takes as many shift steps as its deepest lane
// needs.  Pass 0 runs on the identity and leaves every part's permutation (global, for pass 1), its byte count, and the
// chunk's permutation = the 64 composed (for bz_mtf_scan, unchanged); pass 1 unrolls the chunk's true list through the
// parts' permutations into every lane's starting list and runs again with the bytes.
constexpr u32 BZ_PARTS = BZ_CHUNKS * 64;      // parts per block
constexpr u32 BZ_LANE_STRIDE = 65;            // dwords between two lanes' lists
struct BzLaneLds {
  u32 list[64 * BZ_LANE_STRIDE];
  u8 cur[256] __attribute__((aligned(4)));  // the list being taken through the parts
};
AHIP_DEVINL u32 bz_part_start(const u16 *__restrict__ syms, u32 nsyms, u32 c) {
  const u32 per = (nsyms + BZ_PARTS - 1) / BZ_PARTS;
  const u64 nominal = (u64)c * per;
  if (nominal >= nsyms) return nsyms;
  u32 st = (u32)nominal;
  if (st > 0 && syms[st - 1] <= 1)
    for (u32 k = 0; k < 32 && st < nsyms && syms[st] <= 1; ++k) ++st;  // (more than 21 digits: the run's owner reports it)
  return st;
}
// chunk c of a block.  part_perms: the chunk's 64 x 256 bytes (written by pass 0, read by pass 1); part_counts: its 64 counts.
template <bool WRITE>
AHIP_DEVINL void bz_mtf_lanes_wave(BzLaneLds &S, const u16 *__restrict__ syms, u32 nsyms, u32 c, u32 limit,
                                   const u8 *__restrict__ list /* pass 1: the chunk's true list */, u32 out_off, u8 *__restrict__ b8,
                                   u8 *__restrict__ part_perms, u32 *__restrict__ part_counts, u8 *__restrict__ perm_out, BzChunk &res,
                                   const u32 lane) {
  u32 *L = S.list + lane * BZ_LANE_STRIDE;  // this lane's list, four entries a dword
  u32 my_off = 0;
  if (WRITE) {
    // every lane's starting list: the chunk's, taken through the parts in front of the lane
    for (u32 i = lane; i < 256; i += 64) S.cur[i] = list[i];
    wave_sync();
    const u32 *pp = (const u32 *)part_perms;
    u32 p_next = pp[lane];  // dword `lane` of part 0's permutation
    for (u32 l = 0; l < 64; ++l) {
      const u32 p = p_next;
      if (l + 1 < 64) p_next = pp[(l + 1) * 64 + lane];
      const u32 mine = ((const u32 *)S.cur)[lane];
      S.list[l * BZ_LANE_STRIDE + lane] = mine;  // lane l starts from the list as it stands
      const u32 y = (u32)S.cur[p & 0xff] | ((u32)S.cur[(p >> 8) & 0xff] << 8) | ((u32)S.cur[(p >> 16) & 0xff] << 16) | ((u32)S.cur[p >> 24] << 24);
      wave_sync();
      ((u32 *)S.cur)[lane] = y;
      wave_sync();
    }
    u32 total;
    my_off = out_off + wave_excl_sum(part_counts[lane], total);
  } else {
#pragma unroll 4
    for (u32 d = 0; d < 64; ++d) L[d] = (4 * d) | ((4 * d + 1) << 8) | ((4 * d + 2) << 16) | ((4 * d + 3) << 24);
  }
  wave_sync();
  const u32 part = c * 64 + lane;
  const u32 s0 = bz_part_start(syms, nsyms, part);
  u32 s1 = part + 1 < BZ_PARTS ? bz_part_start(syms, nsyms, part + 1) : nsyms;
  if (s1 < s0) s1 = s0;
  const u32 steps = wave_umax(s1 - s0);
  u32 cnt = 0, r0 = 0, es = 0, bad = 0;  // bytes made; the zero run being read: digits so far, value so far
  u32 sb[4] = {0, 0, 0, 0};              // the next (up to) eight symbols
  // pass 1: the block's bytes go to b8[] (bz_tinv_scatter spreads them into tt[]); a lane gathers four and stores an aligned
  // dword (a store per symbol was 64 cache lines per instruction for one byte of each: the pass took three times pass 0)
  u32 ob = 0, nob = 0;  // bytes b8[my_off + cnt - nob .. my_off + cnt) are still here, the oldest lowest
  auto spill = [&]() {
    u8 *dst = b8 + (my_off + cnt - nob);
    if (nob == 4) *(u32 *)dst = ob;
    else for (u32 k = 0; k < nob; ++k) dst[k] = (u8)(ob >> (8 * k));
    nob = 0; ob = 0;
  };
  auto emit = [&](u32 byte) {  // counts the byte
    ob |= byte << (8 * nob);
    ++nob; ++cnt;
    if (((my_off + cnt) & 3u) == 0) spill();
  };
  // n copies of a byte (a zero run): whole dwords once the position is aligned -- the wave waits for its longest run at
  // every step, so what a run costs per byte is what the pass costs
  auto emit_run = [&](u32 byte, u32 n) {
    while (n && ((my_off + cnt) & 3u)) { emit(byte); --n; }
    if (n >= 4) {  // (aligned: nothing is pending)
      u32 *d = (u32 *)(b8 + my_off + cnt);
      const u32 nd = n >> 2, w = byte * 0x01010101u;
      for (u32 k = 0; k < nd; ++k) d[k] = w;
      cnt += 4 * nd;
      n &= 3u;
    }
    while (n) { emit(byte); --n; }
  };
  for (u32 j = 0; j < steps; ++j) {
    const bool on = s0 + j < s1 && !bad;
    if ((j & 7) == 0 && on) {
      const u32 at = s0 + j;
      if (at + 8 <= nsyms) { const uint4 v = load_u128_unaligned((const u8 *)(syms + at)); sb[0] = v.x; sb[1] = v.y; sb[2] = v.z; sb[3] = v.w; }
      else for (u32 k = 0; k < 8; ++k) { const u32 x = at + k < nsyms ? (u32)syms[at + k] : 0u; if (k & 1) sb[k >> 1] |= x << 16; else sb[k >> 1] = x; }
    }
    const u32 jj = j & 7;
    const u32 w2 = jj < 2 ? sb[0] : (jj < 4 ? sb[1] : (jj < 6 ? sb[2] : sb[3]));
    const u32 sym = (jj & 1) ? w2 >> 16 : w2 & 0xffffu;
    const bool digit = on && sym <= 1;
    if (digit) {
      if (r0 >= 21) bad = 1;  // the 22nd digit finds N >= 2M
      else { es += (1u + sym) << r0; ++r0; }
    }
    // a run ends in front of a list symbol
    const bool flush = on && !digit && r0 != 0;
    if (flush && cnt + es > limit) bad = 1;
    if (WRITE) {
      const bool big = flush && !bad && es >= 512;  // a long run is written by the whole wave
      if (__any(big)) {
        if (big) spill();
        for (u64 bm = __ballot(big); bm; bm &= bm - 1) {
          const int src = __builtin_ctzll(bm);
          const u32 b0 = lane_bcast(my_off + cnt, src), n = lane_bcast(es, src), byte = lane_bcast(L[0] & 0xffu, src);
          for (u32 i = lane; i < n; i += 64) b8[b0 + i] = (u8)byte;
        }
      }
      if (flush && !bad && !big) {
        const u32 byte = L[0] & 0xffu;
        emit_run(byte, es);
        es = 0;  // (counted)
      }
    }
    if (flush && !bad) { cnt += es; r0 = 0; es = 0; }
    // the list symbol: entry nn moves to the front
    bool mtf = on && !digit && !bad;
    if (mtf && cnt + 1 > limit) { bad = 1; mtf = false; }  // (the symbol finds nblock >= nblockMAX)
    u32 q = 0, xq = 0, m = 0, carry = 0;
    if (mtf) {
      const u32 nn = sym - 1, r = nn & 3;
      q = nn >> 2;
      xq = L[q];
      carry = (xq >> (8 * r)) & 0xffu;  // the byte itself: it goes to the front
      m = r == 3 ? 0xffffffffu : (1u << (8 * (r + 1))) - 1;  // the bytes of dword q that move
      if (WRITE) emit(carry); else ++cnt;
    }
    for (u32 d = 0; __any(mtf && d < q); d += 2) {
      if (mtf && d < q) {
        const bool two = d + 1 < q;
        const u32 x0 = L[d], x1 = two ? L[d + 1] : 0u;
        L[d] = (x0 << 8) | carry;
        carry = x0 >> 24;
        if (two) { L[d + 1] = (x1 << 8) | carry; carry = x1 >> 24; }
      }
    }
    if (mtf) L[q] = ((((xq << 8) | carry) & m) | (xq & ~m));
  }
  if (r0 && !bad) {  // a run ends with its part
    if (cnt + es > limit) bad = 1;
    else {
      if (WRITE) emit_run(L[0] & 0xffu, es);
      else cnt += es;
    }
  }
  if (WRITE) spill();
  wave_sync();
  if (!WRITE) {
    // the parts' permutations (lane t writes dword t of part l: 256 bytes a store), and the chunk's = all of them composed
    u32 *pp = (u32 *)part_perms;
    S.cur[lane] = (u8)lane; S.cur[lane + 64] = (u8)(lane + 64); S.cur[lane + 128] = (u8)(lane + 128); S.cur[lane + 192] = (u8)(lane + 192);
    wave_sync();
    for (u32 l = 0; l < 64; ++l) {
      const u32 p = S.list[l * BZ_LANE_STRIDE + lane];
      pp[l * 64 + lane] = p;
      const u32 y = (u32)S.cur[p & 0xff] | ((u32)S.cur[(p >> 8) & 0xff] << 8) | ((u32)S.cur[(p >> 16) & 0xff] << 16) | ((u32)S.cur[p >> 24] << 24);
      wave_sync();
      ((u32 *)S.cur)[lane] = y;
      wave_sync();
    }
    ((u32 *)perm_out)[lane] = ((const u32 *)S.cur)[lane];
    part_counts[lane] = cnt;
  }
  u32 total;
  (void)wave_excl_sum(cnt, total);
  res.count = total;
  res.bad = __any(bad != 0) ? 1u : 0u;
}

// one wave per block: chunk offsets and lists, the block's size and verdict
struct BzScanLds { u8 cur[256], nxt[256]; };
AHIP_DEVINL void bz_mtf_scan_wave(BzScanLds &S, BzResult &R, u32 nblock_max, const BzChunk *__restrict__ chunks,
                                  const u8 *__restrict__ perms, const u8 *__restrict__ list0, u8 *__restrict__ lists,
                                  u32 *__restrict__ offs, const u32 lane) {
  if (R.nsyms == 0 && R.status != BZ_ST_OK) return;  // nothing was decoded (header trouble, end-of-stream marker, BZ_ST_NEG at once)
  for (u32 i = lane; i < 256; i += 64) S.cur[i] = list0[i];
  wave_sync();
  u64 running = 0;
  u32 bad = 0;
  for (u32 c = 0; c < BZ_CHUNKS; ++c) {
    for (u32 i = lane; i < 256; i += 64) lists[c * 256 + i] = S.cur[i];
    if (lane == 0) offs[c] = running > 0xffffffffull ? 0xffffffffu : (u32)running;
    bad |= uniform(chunks[c].bad);
    running += uniform(chunks[c].count);
    for (u32 i = lane; i < 256; i += 64) S.nxt[i] = S.cur[perms[c * 256 + i]];
    wave_sync();
    for (u32 i = lane; i < 256; i += 64) S.cur[i] = S.nxt[i];
    wave_sync();
  }
  // every error of the list / run side comes before the one the Huffman side stopped at, and reads as `false`
  // (BZ_ST_NEG stays: the reference goes on behind a failing _getMtfVal, and so does bz_block_exact)
  u32 status = R.status;
  if (bad || running > nblock_max) status = BZ_ST_FALSE;
  if (status == BZ_ST_OK && R.pad_orig_ptr >= running) status = BZ_ST_FALSE;
  R.status = status;
  R.nblock = status == BZ_ST_OK ? (u32)running : 0u;
}

// ---- blocks whose _getMtfVal fails: the reference's own loop, symbol by symbol ----
// _getMtfVal returns -1 when the selectors run out, when a code is longer than 20 bits or when its index lies outside the
// alphabet (bzip2_decoder.dart:735, :753, :766) -- and only its FIRST result is checked (:271).  Inside the loop (:304,
// :385) the reference goes on with -1 as a symbol: nn = -2 takes the short-list branch, reads _mtfa[_mtfbase[0] - 2]
// (the array is fresh, zero-filled, for every block; below the list it holds zeros, or what the list left there before
// its last rebuild), shifts nothing, makes that byte the front of the list, stores it, and decodes on from wherever the
// bit reader stands -- until an end-of-block symbol, nblockMAX (`false`), the end of the input or index -1 (RangeError).
// What such a block makes depends on the physical layout of the 4096-byte array, so this is the reference loop itself,
// one lane, for exactly those (damaged) blocks: BZ_ST_NEG -> status, nblock, end bit and the block's bytes in b8.
// mtfa: 4096 bytes of scratch.  sel / list0 (= seqToUnseq, zeros behind the symbols in use) as bz_header left them.
// b8 == nullptr: nothing is written (bz_fail_cursor: only where the reader stands when the loop ends is wanted).
AHIP_DEVINL void bz_block_exact_lane(const BzTables &T, const u8 *__restrict__ sel, const u8 *__restrict__ list0, const u8 *__restrict__ in,
                                     u64 n, u32 nblock_max, u8 *__restrict__ mtfa, u8 *__restrict__ b8, BzResult &R) {
  BzBits b{in, n, 0, false, 0, 0, 0};
  bz_seek(b, T.sym_bit);
  const i32 nsel = (i32)T.nsel, eob = (i32)T.eob;
  i32 group_no = -1, group_pos = 0, gsel = 0, gmin = 0;
  auto get = [&]() -> i32 {  // _getMtfVal, :732-772
    if (group_pos == 0) {
      group_no++;
      if (group_no >= nsel) return -1;
      group_pos = 50;
      gsel = sel[group_no];
      gmin = T.min_len[gsel];
    }
    group_pos--;
    i32 zn = gmin;
    i32 zvec = (i32)bz_bits(b, (u32)zn);
    for (;;) {
      if (zn > 20) return -1;
      if (zvec <= T.limit[gsel][zn]) break;
      zn++;
      zvec = (zvec << 1) | (i32)bz_bits(b, 1);
    }
    const i32 idx = zvec - T.base[gsel][zn];
    if (idx < 0 || idx >= 258) return -1;
    return (i32)T.perm[gsel][idx];
  };
  i32 mtfbase[16];
  for (u32 i = 0; i < 4096; ++i) mtfa[i] = 0;
  {
    i32 kk = 4095;
    for (i32 ii = 15; ii >= 0; ii--) {
      for (i32 jj = 15; jj >= 0; jj--) { mtfa[kk] = (u8)(ii * 16 + jj); kk--; }
      mtfbase[ii] = kk + 1;
    }
  }
  u32 nblock = 0, status = BZ_ST_OK;
  i32 next_sym = get();
  if (b.fault) status = BZ_ST_RANGE;
  else if (next_sym < 0) status = BZ_ST_FALSE;
  while (status == BZ_ST_OK) {
    if (next_sym == eob) break;
    if (next_sym == 0 || next_sym == 1) {
      i32 es = -1, N = 1;
      do {
        if (N >= 2 * 1024 * 1024) { status = BZ_ST_FALSE; break; }
        es += next_sym == 0 ? N : 2 * N;
        N *= 2;
        next_sym = get();
        if (b.fault) { status = BZ_ST_RANGE; break; }
      } while (next_sym == 0 || next_sym == 1);
      if (status != BZ_ST_OK) break;
      es++;
      const u8 uc = list0[mtfa[mtfbase[0]]];
      if ((u64)nblock + (u64)es > nblock_max) { status = BZ_ST_FALSE; break; }
      if (b8) for (i32 k = 0; k < es; ++k) b8[nblock + k] = uc;
      nblock += (u32)es;
      continue;
    }
    if (nblock >= nblock_max) { status = BZ_ST_FALSE; break; }
    i32 nn = next_sym - 1;  // (-2 for the symbol -1)
    u8 uc;
    if (nn < 16) {
      const i32 pp = mtfbase[0];
      if (pp + nn < 0) { status = BZ_ST_RANGE; break; }  // Uint8List[-1]
      uc = mtfa[pp + nn];
      while (nn > 0) { mtfa[pp + nn] = mtfa[pp + nn - 1]; nn--; }
      mtfa[pp] = uc;
    } else {
      i32 lno = nn / 16;
      const i32 off = nn % 16;
      i32 pp = mtfbase[lno] + off;
      uc = mtfa[pp];
      while (pp > mtfbase[lno]) { mtfa[pp] = mtfa[pp - 1]; pp--; }
      mtfbase[lno]++;
      while (lno > 0) {
        mtfbase[lno]--;
        mtfa[mtfbase[lno]] = mtfa[mtfbase[lno - 1] + 15];
        lno--;
      }
      mtfbase[0]--;
      mtfa[mtfbase[0]] = uc;
      if (mtfbase[0] == 0) {
        i32 kk = 4095;
        for (i32 ii = 15; ii >= 0; ii--) {
          for (i32 jj = 15; jj >= 0; jj--) { mtfa[kk] = mtfa[mtfbase[ii] + jj]; kk--; }
          mtfbase[ii] = kk + 1;
        }
      }
    }
    if (b8) b8[nblock] = list0[uc];
    nblock++;
    next_sym = get();
    if (b.fault) { status = BZ_ST_RANGE; break; }
  }
  if (status == BZ_ST_OK && R.pad_orig_ptr >= nblock) status = BZ_ST_FALSE;  // (origPtr, :392; the count checks :399-432 cannot fail)
  R.status = status;
  R.nblock = status == BZ_ST_OK ? nblock : 0u;
  R.end_bit = b.bit;
}

#ifndef AHIP_HOST_EMU
// only_serial: just the blocks the position-parallel pass handed back (BZ_ST_HUFF_SERIAL)
__global__ __launch_bounds__(64) void bz_decode_block(const u8 *__restrict__ in, u64 n, const BzCand *__restrict__ cands,
                                                      u32 ncand, u16 *__restrict__ syms_all, u8 *__restrict__ list0_all,
                                                      u8 *__restrict__ sel_all, BzResult *__restrict__ results, u32 only_serial,
                                                      BzTables *__restrict__ tables) {
  __shared__ BzLds L;
  const u32 blk = blockIdx.x, lane = threadIdx.x;
  if (blk >= ncand) return;
  if (only_serial && results[blk].status != BZ_ST_HUFF_SERIAL) return;
  BzResult R;
  bz_decode_block_wave(L, in, n, cands[blk], syms_all + (u64)blk * BZ_SYM_CAP, list0_all + (u64)blk * 256,
                       sel_all + (u64)blk * BZ_MAX_SELECTORS, R, lane, tables + blk);  // (the tables too: bz_block_exact may need them)
  if (lane == 0) results[blk] = R;
}
// one workgroup per candidate; only the blocks marked BZ_ST_NEG do anything (damaged input), one lane of them
__global__ __launch_bounds__(64) void bz_block_exact(const u8 *__restrict__ in, u64 n, u32 ncand, const BzTables *__restrict__ tables,
                                                     const u8 *__restrict__ list0_all, const u8 *__restrict__ sel_all,
                                                     BzResult *__restrict__ results, u8 *__restrict__ b8_all, u32 block_size100k) {
  __shared__ u8 mtfa[4096];
  const u32 blk = blockIdx.x;
  if (blk >= ncand || results[blk].status != BZ_ST_NEG) return;
  if (threadIdx.x != 0) return;
  const u32 nblock_max = 100000u * block_size100k;
  BzResult R = results[blk];
  bz_block_exact_lane(tables[blk], sel_all + (u64)blk * BZ_MAX_SELECTORS, list0_all + (u64)blk * 256, in, n, nblock_max, mtfa,
                      b8_all + (u64)blk * nblock_max, R);
  results[blk] = R;
}
// Where the reference's bit reader stands when _readCompressed gives up on block `blk` somewhere in its symbol loop (the
// stream position decodeStream leaves behind, ahip_last_consumed): the position-parallel passes know THAT a block fails
// there -- the list outgrowing the block, :292-326 -- not at which bit, so the reference's own loop is run once more, one
// lane, nothing written.  Only for the one block a damaged stream stops at, and only when its header held (sym_bit != 0:
// a header that fails knows its own bit, BzResult::end_bit).  *cursor = 0: no answer.
__global__ __launch_bounds__(64) void bz_fail_cursor(const u8 *__restrict__ in, u64 n, u32 blk, const BzTables *__restrict__ tables,
                                                     const u8 *__restrict__ list0_all, const u8 *__restrict__ sel_all,
                                                     const BzResult *__restrict__ results, u32 block_size100k, u64 *__restrict__ cursor) {
  __shared__ u8 mtfa[4096];
  if (threadIdx.x != 0) return;
  *cursor = 0;
  if (tables[blk].sym_bit == 0) return;
  BzResult R = results[blk];
  bz_block_exact_lane(tables[blk], sel_all + (u64)blk * BZ_MAX_SELECTORS, list0_all + (u64)blk * 256, in, n, 100000u * block_size100k, mtfa,
                      (u8 *)nullptr, R);
  *cursor = R.end_bit;
}
// bz_header: the block headers and tables for the position-parallel pass (one wave per candidate)
__global__ __launch_bounds__(64) void bz_header(const u8 *__restrict__ in, u64 n, const BzCand *__restrict__ cands, u32 ncand,
                                                BzTables *__restrict__ tables, u8 *__restrict__ list0_all,
                                                u8 *__restrict__ sel_all, BzResult *__restrict__ results) {
  __shared__ BzLds L;
  const u32 blk = blockIdx.x, lane = threadIdx.x;
  if (blk >= ncand) return;
  BzResult R;
  bz_decode_block_wave(L, in, n, cands[blk], (u16 *)nullptr, list0_all + (u64)blk * 256, sel_all + (u64)blk * BZ_MAX_SELECTORS, R, lane,
                       tables + blk);
  if (lane == 0) {
    results[blk] = R;
    if (R.status != BZ_ST_OK || cands[blk].kind != 0) tables[blk].ngroups = 0;  // nothing to decode
  }
}
// grid (tiles, blocks).  cands_all / ncand_all: the whole sorted candidate list (a block's bits end at the next candidate)
__global__ __launch_bounds__(512) void bz_jump_tiles(const u8 *__restrict__ in, u64 n, const BzCand *__restrict__ cands_all,
                                                     u32 ncand_all, u32 first, const BzTables *__restrict__ tables,
                                                     u16 *__restrict__ j50, u64 j50_bit0, u64 tstride) {
  __shared__ BzTileLds S;
  const u32 blk = blockIdx.y;
  const BzTables *T = tables + blk;
  if (T->ngroups == 0) return;
  const u64 lim = first + blk + 1 < ncand_all ? cands_all[first + blk + 1].bit : n * 8;
  const u64 tile_bit = T->sym_bit + (u64)blockIdx.x * BZ_TW;
  if (tile_bit >= lim) return;
  bz_jump_tile(S, in, n, T, tile_bit, lim, j50 + (tile_bit - j50_bit0), tstride, threadIdx.x, blockDim.x);
}
__global__ __launch_bounds__(512) void bz_group_starts(const u8 *__restrict__ in, u64 n, const BzCand *__restrict__ cands_all, u32 ncand_all,
                                               u32 first, const BzTables *__restrict__ tables, const u8 *__restrict__ sel_all,
                                               const u16 *__restrict__ j50, u64 j50_bit0, u64 tstride, u32 *__restrict__ gstart_all,
                                               u32 *__restrict__ gcount, BzResult *__restrict__ results) {
  __shared__ BzWalkLds S;
  const u32 blk = blockIdx.x;
  const BzTables *T = tables + blk;
  if (T->ngroups == 0) { if (threadIdx.x == 0) gcount[blk] = 0; return; }
  const u64 lim = first + blk + 1 < ncand_all ? cands_all[first + blk + 1].bit : n * 8;
  u32 found = 0, marked = 0;
  bz_walk_groups(S, T, sel_all + (u64)blk * BZ_MAX_SELECTORS, lim, j50, j50_bit0, tstride, gstart_all + (u64)blk * BZ_MAX_SELECTORS,
                 found, marked, threadIdx.x, blockDim.x);
  if (threadIdx.x == 0) {
    gcount[blk] = marked == 1 ? found : 0u;
    if (marked != 1) { results[blk].status = marked == 0 ? BZ_ST_NEG : BZ_ST_HUFF_SERIAL; results[blk].nsyms = 0; }  // (out of selectors: _getMtfVal's first -1)
  }
}
// grid (ceil(BZ_MAX_SELECTORS / 256), blocks): one thread per group
__global__ __launch_bounds__(256) void bz_decode_groups(const u8 *__restrict__ in, u64 n, const BzTables *__restrict__ tables,
                                                        const u8 *__restrict__ sel_all, const u32 *__restrict__ gstart_all,
                                                        const u32 *__restrict__ gcount, u16 *__restrict__ syms_all,
                                                        BzResult *__restrict__ results) {
  __shared__ BzGroupLds S;
  const u32 blk = blockIdx.y, found = gcount[blk];
  if (blockIdx.x * 256 >= found) return;
  const BzTables *T = tables + blk;
  for (u32 i = threadIdx.x; i < 6 * 24; i += 256) { (&S.limit[0][0])[i] = (&T->limit[0][0])[i]; (&S.base[0][0])[i] = (&T->base[0][0])[i]; }
  for (u32 i = threadIdx.x; i < 6 * 258; i += 256) (&S.perm[0][0])[i] = (&T->perm[0][0])[i];
  for (u32 i = threadIdx.x; i < 6 * (1u << BZ_FAST_BITS) / 2; i += 256) ((u32 *)&S.e16[0][0])[i] = ((const u32 *)&T->e16[0][0])[i];
  if (threadIdx.x < 6) S.min_len[threadIdx.x] = T->min_len[threadIdx.x];
  __syncthreads();
  const u32 g = blockIdx.x * 256 + threadIdx.x;
  if (g >= found) return;
  BzGroupEnd end;
  const bool last = g + 1 == found;
  bz_decode_group(S, T->eob, in, n, T->sym_bit, g, gstart_all[(u64)blk * BZ_MAX_SELECTORS + g], sel_all[(u64)blk * BZ_MAX_SELECTORS + g],
                  last, syms_all + (u64)blk * BZ_SYM_CAP, end);
  if (last) { results[blk].status = end.status; results[blk].nsyms = end.nsyms; results[blk].end_bit = end.end_bit; }
}
// grid (BZ_CHUNKS, blocks), 64 threads: one wave per chunk, one part of it per lane
template <bool WRITE>
__global__ __launch_bounds__(64) void bz_mtf_lanes(const u16 *__restrict__ syms_all, const BzResult *__restrict__ results,
                                                   u32 block_size100k, BzChunk *__restrict__ chunks_all, u8 *__restrict__ perms_all,
                                                   const u8 *__restrict__ lists_all, const u32 *__restrict__ offs_all,
                                                   u8 *__restrict__ b8_all, u8 *__restrict__ part_perms_all, u32 *__restrict__ part_counts_all) {
  __shared__ BzLaneLds S;
  const u32 blk = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
  const u32 nblock_max = 100000u * block_size100k;
  const BzResult &R = results[blk];
  if (WRITE && R.status != BZ_ST_OK) return;
  BzChunk res;
  const u64 ci = (u64)blk * BZ_CHUNKS + c;
  bz_mtf_lanes_wave<WRITE>(S, syms_all + (u64)blk * BZ_SYM_CAP, R.nsyms, c, nblock_max, WRITE ? lists_all + ci * 256 : nullptr,
                           WRITE ? offs_all[ci] : 0u, b8_all + (u64)blk * nblock_max, part_perms_all + ci * 64 * 256, part_counts_all + ci * 64,
                           perms_all + ci * 256, res, lane);
  if (!WRITE && lane == 0) chunks_all[ci] = res;
}
__global__ __launch_bounds__(64) void bz_mtf_scan(BzResult *__restrict__ results, const BzCand *__restrict__ cands, u32 ncand,
                                                  u32 block_size100k, const BzChunk *__restrict__ chunks_all, const u8 *__restrict__ perms_all,
                                                  const u8 *__restrict__ list0_all, u8 *__restrict__ lists_all,
                                                  u32 *__restrict__ offs_all) {
  __shared__ BzScanLds S;
  const u32 blk = blockIdx.x, lane = threadIdx.x;
  if (blk >= ncand || cands[blk].kind != 0) return;
  BzResult R = results[blk];
  bz_mtf_scan_wave(S, R, 100000u * block_size100k, chunks_all + (u64)blk * BZ_CHUNKS, perms_all + (u64)blk * BZ_CHUNKS * 256,
                   list0_all + (u64)blk * 256, lists_all + (u64)blk * BZ_CHUNKS * 256, offs_all + (u64)blk * BZ_CHUNKS, lane);
  if (lane == 0) { results[blk].status = R.status; results[blk].nblock = R.nblock; }
}

// ---- phase 2 (own launch): T^-1 ----
// tt[i] holds the block's bytes (low 8 bits).  Result: tt[j] |= i << 8 for the j-th smallest (byte, i) --
// a stable counting sort, i.e. the reference's serial loop (:406-439).  64 waves per block, each owning a
// contiguous sixteenth: per-wave histograms, a (symbol, wave) exclusive scan, then 64 elements per step:
// eight ballots give every lane the set of lanes holding the same byte, so its rank inside the step is a
// popcount and only the first lane of each byte value bumps the wave's cursor.
// Round 4: BZ_TINV_PARTS workgroups per block (a workgroup's 16 waves own a contiguous 1 / (16 PARTS) of the block each), in
// three launches -- the histograms (and tt[i] = the byte), the cursors of every (wave, symbol) from a scan over the 64
// waves and the 256 symbols, the placement -- because one workgroup per block was 1.4 ms however few blocks there were.
constexpr u32 BZ_TINV_PARTS = 4, BZ_TINV_WAVES = 16 * BZ_TINV_PARTS;
// grid (PARTS, blocks) x 1024: hist[blk][wave][sym]
__global__ __launch_bounds__(1024) void bz_tinv_hist(u32 *__restrict__ tt_all, const u8 *__restrict__ b8_all, u32 block_size100k,
                                                     const BzCand *__restrict__ cands, const BzResult *__restrict__ results,
                                                     u32 *__restrict__ hist_all) {
  __shared__ u32 cur[16][256];
  const u32 blk = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK) return;
  const u32 nblock = results[blk].nblock;
  u32 *tt = tt_all + (u64)blk * (100000u * block_size100k);
  const u8 *b8 = b8_all + (u64)blk * (100000u * block_size100k);  // the block's bytes as bz_mtf_lanes<true> left them
  for (u32 i = tid; i < 16 * 256; i += 1024) (&cur[0][0])[i] = 0;
  __syncthreads();
  const u32 seg = (((nblock + BZ_TINV_WAVES - 1) / BZ_TINV_WAVES) + 63) & ~63u;
  const u32 gw = part * 16 + w;
  const u32 lo = gw * seg < nblock ? gw * seg : nblock;
  const u32 hi = lo + seg < nblock ? lo + seg : nblock;
  for (u32 i = lo + lane; i < hi; i += 64) { const u32 b = b8[i]; tt[i] = b; atomicAdd(&cur[w][b], 1u); }  // (tt[i]: the byte now, the link later)
  __syncthreads();
  u32 *hist = hist_all + ((u64)blk * BZ_TINV_WAVES + part * 16) * 256;
  for (u32 i = tid; i < 16 * 256; i += 1024) hist[i] = (&cur[0][0])[i];
}
// grid (blocks) x 256: hist[blk][wave][sym] -> the first place of that wave's bytes of that symbol
__global__ __launch_bounds__(256) void bz_tinv_cursors(const BzCand *__restrict__ cands, const BzResult *__restrict__ results,
                                                       u32 *__restrict__ hist_all) {
  __shared__ u32 tot[256];
  const u32 blk = blockIdx.x, tid = threadIdx.x;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK) return;
  u32 *hist = hist_all + (u64)blk * BZ_TINV_WAVES * 256;
  u32 acc = 0;
  for (u32 k = 0; k < BZ_TINV_WAVES; ++k) { const u32 c = hist[k * 256 + tid]; hist[k * 256 + tid] = acc; acc += c; }
  tot[tid] = acc;
  __syncthreads();
  if (tid < 64) {  // cftab: exclusive scan of the 256 symbol totals, 4 per lane
    u32 t0 = tot[4 * tid], t1 = tot[4 * tid + 1], t2 = tot[4 * tid + 2], t3 = tot[4 * tid + 3];
    u32 total;
    const u32 ex = wave_excl_sum(t0 + t1 + t2 + t3, total);
    tot[4 * tid] = ex; tot[4 * tid + 1] = ex + t0; tot[4 * tid + 2] = ex + t0 + t1; tot[4 * tid + 3] = ex + t0 + t1 + t2;
  }
  __syncthreads();
  const u32 basep = tot[tid];
  for (u32 k = 0; k < BZ_TINV_WAVES; ++k) hist[k * 256 + tid] += basep;
}
// grid (PARTS, blocks) x 1024
__global__ __launch_bounds__(1024) void bz_tinv_scatter(u32 *__restrict__ tt_all, const u8 *__restrict__ b8_all, u32 block_size100k,
                                                        const BzCand *__restrict__ cands, const BzResult *__restrict__ results,
                                                        const u32 *__restrict__ hist_all) {
  __shared__ u32 cur[16][256];
  const u32 blk = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK) return;
  const u32 nblock = results[blk].nblock;
  u32 *tt = tt_all + (u64)blk * (100000u * block_size100k);
  const u8 *b8 = b8_all + (u64)blk * (100000u * block_size100k);
  const u32 *hist = hist_all + ((u64)blk * BZ_TINV_WAVES + part * 16) * 256;
  for (u32 i = tid; i < 16 * 256; i += 1024) (&cur[0][0])[i] = hist[i];
  __syncthreads();
  const u32 seg = (((nblock + BZ_TINV_WAVES - 1) / BZ_TINV_WAVES) + 63) & ~63u;
  const u32 gw = part * 16 + w;
  const u32 lo = gw * seg < nblock ? gw * seg : nblock;
  const u32 hi = lo + seg < nblock ? lo + seg : nblock;
  const u64 below = (1ull << lane) - 1;
  for (u32 i0 = lo; i0 < hi; i0 += 64) {
    const u32 i = i0 + lane;
    const bool act = i < hi;
    const u32 sym = act ? (u32)b8[i] : 0u;
    u64 same = __ballot(act);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u64 bal = __ballot((sym >> k) & 1);
      same &= ((sym >> k) & 1) ? bal : ~bal;
    }
    const u32 rank = (u32)__popcll(same & below);
    u32 pos = 0;
    if (act) pos = cur[w][sym] + rank;
    wave_sync();  // every lane has read its cursor before the leaders move them
    if (act && rank == 0) cur[w][sym] += (u32)__popcll(same);
    wave_sync();
    // (tt[pos] holds its byte since bz_tinv_hist: only the upper 24 bits are touched)
    if (act) atomicOr(&tt[pos], i << 8);
  }
}

#endif  // AHIP_HOST_EMU (the serial inverse transform below also runs on the host: tests/emu/bzip2_chain_emu.cc)

// ---- phase 3 (own launch): inverse BWT + un-RLE + CRC, one lane per block ----
// bzip2_decoder.dart:610-727 restated 1:1 (the reference writes its bytes as it goes and may fail AFTER writing:
// `cNBlockUsed > sSaveNBlockPP` behind a run whose count byte lay beyond the block's data, :612-631 -- those bytes stay
// in the output, the verdict is `false`; the host chain places them: bzip2_chain.hpp).  out_cap: bytes beyond it are
// counted, not written (BZ_ST_OVERFLOW).
AHIP_DEVINL void bz_unbwt_block(const u32 *__restrict__ tt, u32 nblock_max, u32 nblock, u32 orig_ptr, u8 *__restrict__ out, u64 out_cap,
                                const u32 *__restrict__ crc_table, u32 &status_out, u64 &olen_out, u32 &crc_out) {
  u32 crc = 0xffffffffu;
  u64 olen = 0;
  u32 status = BZ_ST_OK;
  u32 t_pos = tt[orig_ptr] >> 8;
  u32 n_used = 0;
  const u32 save_pp = nblock + 1;
#define BZ_EMIT(ch)                                                     \
  do {                                                                  \
    if (olen < out_cap) out[olen] = (u8)(ch); else status = BZ_ST_OVERFLOW; \
    olen++;                                                             \
    crc = (crc << 8) ^ crc_table[((crc >> 24) & 0xff) ^ ((ch) & 0xff)]; \
  } while (0)
#define BZ_STEP(var)                                    \
  do {                                                  \
    if (t_pos >= nblock_max) { status = BZ_ST_FALSE; goto fin; } \
    t_pos = tt[t_pos];                                  \
    (var) = t_pos & 0xff;                               \
    t_pos >>= 8;                                        \
    n_used++;                                           \
  } while (0)
  if (t_pos < nblock_max) {
    u32 k0, k1;
    t_pos = tt[t_pos];
    k0 = t_pos & 0xff;
    t_pos >>= 8;
    n_used++;
    u32 out_len = 0, out_ch = 0, c_k0 = k0;
    for (;;) {
      if (out_len > 0) { for (u32 r = 0; r < out_len; ++r) BZ_EMIT(out_ch); }
      if (n_used > save_pp) { status = BZ_ST_FALSE; break; }
      if (n_used == save_pp) break;
      out_ch = c_k0;
      BZ_STEP(k1);
      if (k1 != c_k0) { c_k0 = k1; BZ_EMIT(out_ch); out_len = 0; continue; }
      if (n_used == save_pp) { BZ_EMIT(out_ch); out_len = 0; continue; }
      out_len = 2;
      BZ_STEP(k1);
      if (n_used == save_pp) continue;
      if (k1 != c_k0) { c_k0 = k1; continue; }
      out_len = 3;
      BZ_STEP(k1);
      if (n_used == save_pp) continue;
      if (k1 != c_k0) { c_k0 = k1; continue; }
      BZ_STEP(k1);
      out_len = k1 + 4;
      BZ_STEP(c_k0);
    }
  }
fin:
#undef BZ_EMIT
#undef BZ_STEP
  status_out = status; olen_out = olen; crc_out = crc ^ 0xffffffffu;
}

#ifndef AHIP_HOST_EMU
// direct_off == nullptr: the blocks the parallel path handed back (BZ_ST_SERIAL) are walked COUNTING only (slab_cap 0:
// a block with bytes reports BZ_ST_OVERFLOW and its true size; one that fails behind its bytes BZ_ST_FALSE and the bytes
// it had written).  direct_off != nullptr: second pass for exactly the blocks the chain placed, written straight to
// direct_out + direct_off[blk] (what a failing block wrote before it failed included).
__global__ __launch_bounds__(64) void bz_unbwt(const u32 *__restrict__ tt_all, u32 block_size100k, u32 ncand,
                                               const BzCand *__restrict__ cands, u8 *__restrict__ slabs, u64 slab_cap,
                                               BzResult *__restrict__ results, const u32 *__restrict__ crc_table,
                                               const u64 *__restrict__ direct_off, u8 *__restrict__ direct_out) {
  const u32 blk = blockIdx.x * 64 + threadIdx.x;
  if (blk >= ncand) return;
  if (cands[blk].kind != 0) return;
  if (direct_off) { if (direct_off[blk] == ~0ull) return; }
  else if (results[blk].status != BZ_ST_SERIAL) return;
  const u32 nblock_max = 100000u * block_size100k;
  const u32 *tt = tt_all + (u64)blk * nblock_max;
  u8 *out = direct_off ? direct_out + direct_off[blk] : slabs + (u64)blk * slab_cap;
  if (direct_off) slab_cap = ~0ull;
  u32 status, crc;
  u64 olen;
  bz_unbwt_block(tt, nblock_max, results[blk].nblock, results[blk].pad_orig_ptr, out, slab_cap, crc_table, status, olen, crc);
  if (direct_off) return;  // sizes and CRC were established by the first pass
  results[blk].status = status;
  results[blk].out_len = olen;
  results[blk].crc = crc;
}
// ---- phase 3 (parallel): list-ranked inverse BWT ----
// Splitter s < S sits at index s * BZ_G; splitter S is the chain head p0 = tt[origPtr] >> 8.
struct BzWalk { u32 next, len; };
AHIP_DEVINL u32 bz_nsplit(u32 nblock) { return (nblock + BZ_G - 1) / BZ_G; }

// Who walks what.  The walk is 0.9 M dependent 4-byte reads at random places of the block's 3.6 MB vector -- every entry
// exactly once, so a 128-byte line is worth fetching only if it stays cached until its 32 entries have been read -- and
// a sublist ends where the chain happens to reach the next multiple of BZ_G: its length is geometric (mean BZ_G, the
// longest of a block's some eleven times that).
//  * With every block of a batch in flight at once (a 2-D grid, one thread per splitter) the vectors of some seventy
//    blocks -- 260 MB -- were walked side by side and nearly every read went to memory (60 G reads/s: the rate of
//    tools/micro/xcc_chase.hip on a 64 MB vector; 230 G/s on one that fits an L2).  Workgroups are dealt to the
//    eight XCDs in turn (workgroup i runs on XCD i mod 8) and each XCD has its own 4 MB L2, so the grid is 8 x `wpx`
//    PERSISTENT workgroups and the ones with blockIdx.x mod 8 == x take the blocks x, x + 8, ... one after the other:
//    a block's vector is read by one XCD only, through an L2 it (nearly) fits in.
//  * That alone changed little (70 % of the reads still missed, TCC_HIT / TCC_MISS): with BZ_G = 128 a block had as many
//    sublists as the XCD has lanes at work, the lanes that were done went on to the next block while the long sublists of
//    this one were still being walked, and three vectors shared the L2.  The share of the reads that fall into such an
//    overlap is (lanes at work) / (sublists of a block): BZ_G = 16 makes it an eighth.  (The price is 56 thousand
//    sublists to rank per block: bz_rank does it in two levels.)
//  * A lane whose sublist has ended takes the next one from the XCD's queue (queue[x]: the next item, an item = block
//    x + 8 (q / stride), splitter q mod stride; one atomic for all the idle lanes of a wave) instead of waiting for the
//    longest sublist of its workgroup.
//    What a sublist needs to start -- the block's verdict, size and head, its rank in the WRITE pass -- is fetched when the
//    wave takes a batch from the queue (a batch lies inside one block: BZ_WALK_IPB items per block), not when a lane
//    starts: those loads, three deep, stalled the whole wave at four steps out of ten.
template <bool WRITE>
__global__ __launch_bounds__(256) void bz_walk(const u32 *__restrict__ tt_all, u32 block_size100k,
                                               const BzCand *__restrict__ cands, const BzResult *__restrict__ results,
                                               BzWalk *__restrict__ walk_all, const u32 *__restrict__ rank_all,
                                               u8 *__restrict__ pre_all, u32 nb, u32 *__restrict__ queue) {
  const u32 xcd = blockIdx.x & 7;
  const u32 nblock_max = 100000u * block_size100k;
  const u32 stride = nblock_max / BZ_G + 2;
  const u32 ipb = (stride + BZ_WALK_BATCH - 1) / BZ_WALK_BATCH * BZ_WALK_BATCH;  // queue items per block
  const u32 nitems = (nb > xcd ? (nb - xcd + 7) / 8 : 0u) * ipb;
  const int lane = threadIdx.x & 63;
  const u64 below = (1ull << lane) - 1;
  bool active = false;
  // the chain a lane is on
  const u32 *tt = tt_all;
  u8 *pre = pre_all;
  u32 cur = 0, len = 0, p0 = 0, nblock = 0, s = 0, blk = 0;
  u32 acc = 0, na = 0, at = 0;  // WRITE: bytes [at, at + na) are still in `acc`; whole aligned dwords are stored as such
  // the wave's batch (wave-uniform): items [pool_next, pool_end) of block b_blk are not handed to a lane yet
  u32 pool_base = 0, pool_next = 0, pool_end = 0, b_blk = 0, b_nblock = 0, b_S = 0, b_p0 = 0, b_s0 = 0;
  u32 rk[BZ_WALK_BATCH / 64];  // WRITE: rank of item pool_base + 64 i + lane
  bool out_of_work = false;
  for (;;) {
    const u64 idle = __ballot(!active);
    // (sublists are handed out to sixteen lanes at a time: the hand-out is some eighty instructions, a step of the walk
    //  fifteen, and with sublists of sixteen entries some lane of a wave is done at nearly every step)
    if (((u32)__popcll(idle) >= BZ_WALK_HANDOUT || idle == ~0ull) && !out_of_work) {
      if (pool_next == pool_end) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&queue[xcd], BZ_WALK_BATCH);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= nitems) out_of_work = true;
        else {
          pool_base = pool_next = base;
          pool_end = base + BZ_WALK_BATCH;
          b_blk = xcd + 8 * (base / ipb);
          const u32 s0 = b_s0 = base % ipb;
          const bool ok = cands[b_blk].kind == 0 && results[b_blk].status == BZ_ST_OK && results[b_blk].nblock != 0;
          b_nblock = ok ? (u32)__builtin_amdgcn_readfirstlane((int)results[b_blk].nblock) : 0u;
          b_S = bz_nsplit(b_nblock);
          if (!ok || s0 > b_S) pool_next = pool_end;  // nothing to walk in this batch
          else {
            b_p0 = (u32)__builtin_amdgcn_readfirstlane((int)(tt_all[(u64)b_blk * nblock_max + results[b_blk].pad_orig_ptr] >> 8));
            if (WRITE) {
#pragma unroll
              for (u32 i = 0; i < BZ_WALK_BATCH / 64; ++i) {
                const u32 si = s0 + 64 * i + (u32)lane;
                rk[i] = si <= b_S ? rank_all[(u64)b_blk * stride + si] : ~0u;
              }
            }
          }
        }
      }
      const u32 avail = pool_end - pool_next;
      if (avail) {
        const u32 rank = (u32)__popcll(idle & below), need = (u32)__popcll(idle);
        const u32 q = pool_next + rank;
        const bool served = !active && rank < avail;  // the others are served from the next batch
        // (every lane takes part in the shuffles)
        u32 my_at = 0;
        if (WRITE) {
          const u32 rel = served ? q - pool_base : 0u;
#pragma unroll
          for (u32 i = 0; i < BZ_WALK_BATCH / 64; ++i) { const u32 v = __shfl(rk[i], (int)(rel & 63)); if ((rel >> 6) == i) my_at = v; }
        }
        pool_next += need < avail ? need : avail;
        if (served) {
          s = b_s0 + (q - pool_base);
          if (s <= b_S && !(WRITE && my_at == ~0u)) {  // (~0: not on the head's cycle -- a duplicate of the head, or corrupt data)
            blk = b_blk;
            nblock = b_nblock;
            p0 = b_p0;
            tt = tt_all + (u64)blk * nblock_max;
            pre = pre_all + (u64)blk * nblock_max;
            cur = s == b_S ? p0 : s * BZ_G;
            len = 0;
            at = my_at; acc = 0; na = 0;
            active = true;
          }
        }
      }
    }
    if (!__any(active)) {
      if (out_of_work) break;
      continue;
    }
    if (active) {
      const u32 w = tt[cur];
      if (WRITE) {
        acc |= (w & 0xffu) << (8 * na);
        ++na;
        if (((at + na) & 3u) == 0) {
#ifdef AHIP_BZ_NT_PRE  // dev: the walk's output past the L2 (it is not read again by this kernel, and the vector needs the room)
          if (na == 4) __builtin_nontemporal_store(acc, (u32 *)(pre + at));
#else
          if (na == 4) *(u32 *)(pre + at) = acc;
#endif
          else for (u32 b = 0; b < na; ++b) pre[at + b] = (u8)(acc >> (8 * b));
          at += na; na = 0; acc = 0;
        }
      }
      cur = w >> 8;
      ++len;
      if (!((cur & (BZ_G - 1)) != 0 && cur != p0 && len < nblock)) {  // the sublist ends here
        if (WRITE) { for (u32 b = 0; b < na; ++b) pre[at + b] = (u8)(acc >> (8 * b)); }
        else {
          const u32 S = bz_nsplit(nblock);
          BzWalk r;
          r.next = cur == p0 ? S : cur / BZ_G;
          r.len = len;
          walk_all[(u64)blk * (nblock_max / BZ_G + 2) + s] = r;
        }
        active = false;
      }
    }
  }
}

// One workgroup per block: the place of every sublist along the cycle, starting at the head.  The sublists (a block has
// nblock / BZ_G + 1 of them, 56 thousand at BZ_G = 16) are themselves a linked list (BzWalk::next), ranked the same way
// one level up: every BZ_G2-th splitter and the head are the second-level splitters (at most 1 024: a thread each), a
// thread follows the sublists from its splitter to the next second-level one adding up their lengths, thread 0 orders
// the second-level splitters along the cycle (in LDS), and every thread goes over its stretch again handing out the places.
// rank_all is ~0 beforehand (the host's memset) and stays so for what is not on the head's cycle.
constexpr u32 BZ_G2 = 64;
static_assert((900000 / BZ_G + 1 + BZ_G2 - 1) / BZ_G2 + 1 <= 1024, "second-level splitters: one thread each");
__global__ __launch_bounds__(1024) void bz_rank(u32 block_size100k, const BzCand *__restrict__ cands,
                                                BzResult *__restrict__ results, const BzWalk *__restrict__ walk_all,
                                                u32 *__restrict__ rank_all) {
  __shared__ u16 nxt2[1024];
  __shared__ u32 ln2[1024];  // stretch length (saturated); overwritten by 0x80000000 | place once the splitter is placed
  const u32 blk = blockIdx.x, t = threadIdx.x;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK) return;
  const u32 nblock = results[blk].nblock;
  if (nblock == 0) return;
  const u32 nblock_max = 100000u * block_size100k;
  const u32 S = bz_nsplit(nblock), stride = nblock_max / BZ_G + 2;
  const u32 S2 = (S + BZ_G2 - 1) / BZ_G2;  // second-level splitters t < S2 are the sublists t * BZ_G2; t == S2 is the head (sublist S)
  const BzWalk *walk = walk_all + (u64)blk * stride;
  const bool mine = t <= S2;
  const u32 start = t == S2 ? S : t * BZ_G2;
  if (mine) {
    u32 cur = start, hops = 0;
    u64 sum = 0;
    do {
      const BzWalk w = walk[cur];
      sum += w.len;
      cur = w.next;
    } while (cur != S && (cur % BZ_G2) != 0 && cur <= S && ++hops <= S);
    nxt2[t] = (u16)(cur == S ? S2 : (cur <= S ? cur / BZ_G2 : S2 + 1));  // (S2 + 1: nowhere -- a damaged record)
    ln2[t] = sum < 0x7fffffffull ? (u32)sum : 0x7fffffffu;
  }
  __syncthreads();
  if (t == 0) {
    u32 s = S2, pos = 0, hops = 0;
    do {
      const u32 l = ln2[s];
      if (l & 0x80000000u) break;  // already placed: not a simple cycle through the head
      ln2[s] = 0x80000000u | pos;
      pos += l;
      s = nxt2[s];
    } while (s != S2 && s <= S2 && ++hops <= S2 && pos < nblock);
    // a valid block is ONE cycle of length nblock; anything else takes the serial path
    if (s != S2 || pos != nblock) results[blk].status = BZ_ST_SERIAL;
  }
  __syncthreads();
  if (mine && (ln2[t] & 0x80000000u)) {
    u32 cur = start, hops = 0, pos = ln2[t] & 0x7fffffffu;
    do {
      rank_all[(u64)blk * stride + cur] = pos;
      const BzWalk w = walk[cur];
      pos += w.len;
      cur = w.next;
    } while (cur != S && (cur % BZ_G2) != 0 && cur <= S && ++hops <= S);
  }
}

// ---- phase 4: run-length undo as a scan ----
// Machine (bzip2_decoder.dart:640-727): a byte is output as it comes; after 4 equal bytes in a row the
// next byte is a repeat count (that many more copies), and the run detector starts afresh.
// State: prev byte + cnt (0 fresh, 1..3 equal bytes so far, 4 = next byte is a count).
struct RleState { u32 prev, cnt; };
AHIP_DEVINL u32 rle_class(const RleState &s, u32 x0) {  // what a span starting with byte x0 sees
  if (s.cnt == 4) return 4;
  return (s.cnt > 0 && s.prev == x0) ? s.cnt : 0u;
}
AHIP_DEVINL RleState rle_entry(u32 cls, u32 x0, u32 prev_if_count) {
  RleState s;
  s.cnt = cls;
  s.prev = cls == 4 ? prev_if_count : x0;
  if (cls == 0) s.prev = 0x100;  // fresh: matches no byte
  return s;
}
// advance by one input byte; returns the number of bytes it puts out (the byte value is s.prev afterwards,
// except for a count, where it is the run byte)
AHIP_DEVINL u32 rle_step(RleState &s, u32 x) {
  if (s.cnt == 4) { s.cnt = 0; return x; }  // prev stays the run byte for the caller; state is fresh
  if (s.cnt > 0 && x == s.prev) { s.cnt += 1; return 1; }
  s.prev = x; s.cnt = 1;
  return 1;
}

struct BzSpan { u32 prev, cnt; u64 off; };  // entry state and output offset of a span

__global__ __launch_bounds__(1024) void bz_rle_scan(u32 block_size100k, const BzCand *__restrict__ cands,
                                                    BzResult *__restrict__ results, const u8 *__restrict__ pre_all,
                                                    BzSpan *__restrict__ spans_all) {
  __shared__ u32 t_state[BZ_SPANS][5];  // packed prev << 8 | cnt
  __shared__ u32 t_len[BZ_SPANS][5];
  __shared__ u16 x0s[BZ_SPANS];
  const u32 blk = blockIdx.x, t = threadIdx.x;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK) return;
  const u32 nblock = results[blk].nblock;
  const u32 nblock_max = 100000u * block_size100k;
  const u8 *pre = pre_all + (u64)blk * nblock_max;
  const u32 sp = (nblock + BZ_SPANS - 1) / BZ_SPANS;
  const u32 lo = t * sp < nblock ? t * sp : nblock, hi = lo + sp < nblock ? lo + sp : nblock;
  const u32 x0 = lo < hi ? pre[lo] : 0x100u;
  x0s[t] = (u16)x0;
  // the run byte of an entering "count" state is unknown here; it only matters for the output bytes, not for
  // lengths or exit states (after a count the machine is fresh), so 0 stands in
  RleState st[5];
  u32 len[5];
#pragma unroll
  for (u32 c = 0; c < 5; ++c) { st[c] = rle_entry(c, x0, 0); len[c] = 0; }
  // The five entry classes are walked side by side only until they have MERGED (the same state: from then on they stay
  // together -- a byte that differs from the one before it does that, i.e. nearly always within the first few bytes);
  // after that one machine runs and its output count goes to all five.  The span is read 16 bytes at a time: a lane's
  // span is its own ~0.9 KB, so a byte load a step was 64 cache lines per load instruction and one byte of each.
  bool merged = false;
  RleState ms{0x100, 0};
  u32 mlen = 0;
  auto eat = [&](u32 x) {
    if (merged) mlen += rle_step(ms, x);
    else {
#pragma unroll
      for (u32 c = 0; c < 5; ++c) len[c] += rle_step(st[c], x);
    }
  };
  auto settle = [&]() {
    if (merged) return;
    bool same = true;
#pragma unroll
    for (u32 c = 1; c < 5; ++c) same = same && st[c].prev == st[0].prev && st[c].cnt == st[0].cnt;
    if (same) { merged = true; ms = st[0]; }
  };
  u32 i = lo;
  for (; i + 16 <= hi; i += 16) {
    const uint4 v = load_u128_unaligned(pre + i);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (u32 j = 0; j < 16; ++j) eat((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
    settle();
  }
  for (; i < hi; ++i) eat(pre[i]);
  if (merged) {
#pragma unroll
    for (u32 c = 0; c < 5; ++c) { st[c] = ms; len[c] += mlen; }
  }
#pragma unroll
  for (u32 c = 0; c < 5; ++c) {
    t_state[t][c] = (st[c].prev << 8) | st[c].cnt;
    t_len[t][c] = len[c];
  }
  __syncthreads();
  __shared__ u32 e_prev[BZ_SPANS], e_cnt[BZ_SPANS];
  __shared__ u64 e_off[BZ_SPANS];
  if (t == 0) {
    RleState s{0x100, 0};
    u64 off = 0;
    for (u32 k = 0; k < BZ_SPANS; ++k) {
      e_prev[k] = s.prev; e_cnt[k] = s.cnt; e_off[k] = off;
      const u32 x = x0s[k];
      if (x == 0x100u) continue;  // empty span
      const u32 c = rle_class(s, x);
      const u32 ps = t_state[k][c];
      off += t_len[k][c];
      // a span entered in "count" state consumes its first byte as the count and is fresh right after, so
      // its recorded exit state does not depend on the (here unknown) run byte
      s.prev = ps >> 8; s.cnt = ps & 0xff;
    }
    results[blk].out_len = off;
    if (s.cnt == 4) results[blk].status = BZ_ST_SERIAL;  // data ends inside an escape: the reference reads on (serial path)
  }
  __syncthreads();
  BzSpan o;
  o.prev = e_prev[t]; o.cnt = e_cnt[t]; o.off = e_off[t];
  spans_all[(u64)blk * BZ_SPANS + t] = o;
}

// GF(2) helpers for the MSB-first CRC-32 (polynomial 0x04c11db7)
AHIP_DEVINL u32 gf_mulmod(u32 a, u32 b) {
  u32 r = 0;
  for (int i = 31; i >= 0; --i) {
    r = (r << 1) ^ ((r >> 31) ? 0x04c11db7u : 0u);
    if ((b >> i) & 1) r ^= a;
  }
  return r;
}
// x^(8 * nbytes) mod P; pw[k] = x^(8 * 2^k) mod P
AHIP_DEVINL u32 gf_xpow8(u64 nbytes, const u32 *pw) {
  u32 r = 1;  // the polynomial "1"
  bool first = true;
  for (u32 k = 0; nbytes; ++k, nbytes >>= 1)
    if (nbytes & 1) { r = first ? pw[k] : gf_mulmod(r, pw[k]); first = false; }
  return r;
}

// The output of every span is gathered eight bytes at a time and stored as one (unaligned) 8-byte word: a byte store a
// step was, like the loads, 64 cache lines per instruction.  The block's CRC is taken afterwards over the finished
// bytes (bz_block_crc): a table-driven CRC inside this loop was a dependent LDS look-up per output byte.
__global__ __launch_bounds__(256) void bz_rle_expand(u32 block_size100k, const BzCand *__restrict__ cands,
                                                     BzResult *__restrict__ results, const u8 *__restrict__ pre_all,
                                                     const BzSpan *__restrict__ spans_all, const u64 *__restrict__ dst_off,
                                                     u8 *__restrict__ out) {
  const u32 blk = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK || dst_off[blk] == ~0ull) return;
  const u32 nblock = results[blk].nblock;
  const u32 nblock_max = 100000u * block_size100k;
  const u8 *pre = pre_all + (u64)blk * nblock_max;
  const u32 sp = (nblock + BZ_SPANS - 1) / BZ_SPANS;
  const u32 lo = t * sp < nblock ? t * sp : nblock, hi = lo + sp < nblock ? lo + sp : nblock;
  const BzSpan e = spans_all[(u64)blk * BZ_SPANS + t];
  RleState s{e.prev, e.cnt};
  u8 *dst = out + dst_off[blk] + e.off;
  u64 acc = 0;  // bytes not stored yet, the oldest lowest
  u32 na = 0;
  auto eat = [&](u32 x) {
    const bool is_count = s.cnt == 4;
    const u32 run_byte = s.prev;
    u32 reps = rle_step(s, x);
    const u64 rep8 = 0x0101010101010101ull * (is_count ? run_byte : x);
    if (is_count) s.prev = 0x100;  // fresh
    while (reps) {
      const u32 room = 8 - na, k = reps < room ? reps : room;
      acc |= (k == 8 ? rep8 : (rep8 & ((1ull << (8 * k)) - 1))) << (8 * na);
      na += k;
      reps -= k;
      if (na == 8) { ((unaligned_u64 *)dst)->v = acc; dst += 8; acc = 0; na = 0; }
    }
  };
  u32 i = lo;
  for (; i + 16 <= hi; i += 16) {
    const uint4 v = load_u128_unaligned(pre + i);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (u32 j = 0; j < 16; ++j) eat((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
  }
  for (; i < hi; ++i) eat(pre[i]);
  for (u32 k = 0; k < na; ++k) dst[k] = (u8)(acc >> (8 * k));
}

// results[blk].crc ^= the MSB-first CRC-32 register (before the final inversion) over the block's output bytes: the waves'
// shares of the zero-start register (ck_raw_wave on bit-reversed bytes, mirrored back) and, once, the contribution of the
// 0xffffffff it starts from.  ck_tab: the tables of checksum_kernels.hpp; bz_tab: 256 MSB-first table entries followed by
// the 64 powers pw[k] = x^(8 * 2^k).
__global__ __launch_bounds__(256) void bz_block_crc(const BzCand *__restrict__ cands, BzResult *__restrict__ results,
                                                    const u64 *__restrict__ dst_off, const u8 *__restrict__ out,
                                                    const u32 *__restrict__ ck_tab, const u32 *__restrict__ bz_tab) {
  __shared__ u32 T[256 * 5];
  for (u32 i = threadIdx.x; i < 256 * 5; i += 256) T[i] = ck_tab[i];
  __syncthreads();
  const u32 blk = blockIdx.y;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK || dst_off[blk] == ~0ull) return;
  const u64 total = results[blk].out_len;
  const u32 lane = threadIdx.x & 63;
  bool any;
  const u32 raw = ck_raw_wave<true>(out + dst_off[blk], total, T, ck_tab + 1280, (u64)blockIdx.x * 4 + (threadIdx.x >> 6), (u64)gridDim.x * 4, lane, &any);
  u32 term = any ? __brev(raw) : 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) term ^= gf_mulmod(0xffffffffu, gf_xpow8(total, bz_tab + 256));  // the 0xffffffff initial value
  if (lane == 0 && term) atomicXor(&results[blk].crc, term);
}

#endif  // AHIP_HOST_EMU

}  // namespace ahip
