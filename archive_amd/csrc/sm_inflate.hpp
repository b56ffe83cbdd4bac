// sm_inflate.hpp -- ONE long DEFLATE stream decoded by many waves.
//
// The reference (lib/src/codecs/zlib/inflate.dart) walks a stream block after block; nothing in the format
// says where a block starts, and a back-reference may reach 32 KiB into what the previous block produced.
// Multi-member input side-steps both (members are independent).  For a single long stream:
//
//   S1 sm_find_kernel      the compressed data is cut every `chunk_bytes`; waves behind each cut look for the first
//                          block start: a DYNAMIC block header (a cheap bit-parallel filter on BTYPE and the HLIT / HDIST
//                          ranges, then a complete code-length code, then survivors are run-length decoded one per
//                          lane and must describe complete trees with an end-of-block code), or a STORED block on a
//                          byte boundary (header byte 0 / 1, LEN, ~LEN: inflate.dart:213-237 -- what an incompressible
//                          stretch is made of: stored blocks back to back, each starting where the data of the one
//                          before ends).  What still slips through is judged by S2.  Fixed blocks carry no signature
//                          and are not searched; a cut without a find simply extends the previous chunk.
//   S2 sm_tokenize_kernel  the ordinary tokenizer (inflate_member<.., CHUNK>) from each find to the block that
//                          starts on the next find.  Run twice: sizes first (a false find decodes garbage and is
//                          dropped when the host follows the chain of ends == starts), then recording tokens at
//                          exact offsets.  A back-reference may reach `hist` bytes in front of its chunk.
//   S3 sm_resolve_kernel   the ordinary resolver on 16-bit symbols: a byte, or a marker "byte j of the 32 KiB in
//                          front of this chunk" for what is not known yet.
//   S4 sm_windows_*        the last 32 KiB behind every chunk: window k is a function of window k-1, composed in
//                          groups (symbolic inside a group, groups in parallel), linked group after group by one
//                          workgroup (the only serial step), then made concrete per chunk.
//   S5 sm_translate_kernel every chunk turns its symbols into bytes with its (now known) window.
//
// Anything unexpected (no finds, a broken chain, an error status inside a chunk) makes the host fall back to the
// one-wave path, which restates the reference exactly; so malformed streams keep their reference verdicts.
#pragma once
#include "inflate_par.hpp"

namespace ahip {

constexpr u32 SM_WINDOW = 32768;

struct SmLds {
  WaveLds w;
  TokLds p;
};

// bits [q, q + 64) of the stream (zero past the end)
AHIP_DEVINL u64 sm_bits64(const u8 *in, u64 in_len, u64 q) {
  const u64 byte = q >> 3;
  u64 lo = 0, hi = 0;
  if (byte + 16 <= in_len) { lo = load_u64_unaligned(in + byte); hi = load_u64_unaligned(in + byte + 8); }
  else {
    for (u32 k = 0; k < 8; ++k) { if (byte + k < in_len) lo |= (u64)in[byte + k] << (8 * k); if (byte + 8 + k < in_len) hi |= (u64)in[byte + 8 + k] << (8 * k); }
  }
  const u32 s = (u32)q & 7;
  return s ? (lo >> s) | (hi << (64 - s)) : lo;
}

// Second filter, one candidate per lane: build the code-length code (canonical, LSB-first table like the
// reference's), run-length decode the HLIT + HDIST code lengths and keep only headers whose literal/length code
// has an end-of-block code and is complete and whose distance code is complete or has at most one code -- what
// every compressor writes and random bits almost never do.  Nothing is stored: only the Kraft sums are kept.
AHIP_DEVINL bool sm_header_plausible(const u8 *in, u64 in_len, u64 q, u8 *tab /* this lane's 128 entries */) {
  const u64 v = sm_bits64(in, in_len, q);
  const u32 hlit = ((u32)(v >> 3) & 31) + 257, hdist = ((u32)(v >> 8) & 31) + 1, ncl = ((u32)(v >> 13) & 15) + 4;
  const u64 w = sm_bits64(in, in_len, q + 17);
  // position of symbol s in the transmitted order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
  const u8 inv[19] = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2};
  // The code-length code, canonical, WITHOUT a look-up table: how many codes of each length (cnt), the first code of each
  // length (first) and where its symbols start in the list of symbols sorted by (length, symbol) (offs) -- 8-bit fields of a
  // u64 each; the caller's first filter has made sure the code is complete, so a code fits 7 bits and a first code 8 -- and
  // that list in this lane's 128 bytes.  A symbol is then decoded by comparing the next 1 .. 7 bits with the lengths' code
  // ranges.  (A 128-entry table per candidate -- seven passes over the 19 symbols, each code replicated 2^(7 - len) times in
  // loops whose trip count is the longest of the 32 lanes' -- was what a call of this function cost; nearly every candidate
  // dies within a few symbols, and the calls are four fifths of the finder.)
  u64 cnt = 0;
#pragma unroll
  for (u32 sym = 0; sym < 19; ++sym) {
    const u32 l = inv[sym] < ncl ? (u32)(w >> (3 * inv[sym])) & 7 : 0u;
    cnt += 1ull << (8 * l);
  }
  cnt &= ~0xffull;  // (length 0 = not used)
  u64 first = 0, offs = 0;
  {
    u32 code = 0, at = 0;
#pragma unroll
    for (u32 bl = 1; bl <= 7; ++bl) {
      const u32 before = bl > 1 ? (u32)(cnt >> (8 * (bl - 1))) & 0xffu : 0u;
      code = (code + before) << 1;
      at += before;
      first |= (u64)(code & 0xffu) << (8 * bl);
      offs |= (u64)at << (8 * bl);
    }
  }
  {
    u64 put = offs;  // field l: where the next symbol of length l goes
#pragma unroll
    for (u32 sym = 0; sym < 19; ++sym) {
      const u32 l = inv[sym] < ncl ? (u32)(w >> (3 * inv[sym])) & 7 : 0u;
      if (l) { tab[(u32)(put >> (8 * l)) & 0xffu] = (u8)sym; put += 1ull << (8 * l); }
    }
  }
  // (The header's bits come from global memory, 64 at a time.  Fetching its first 96 bytes into this lane's LDS at once changed
  //  nothing -- a call of this function takes ~ 80 us, 1.2 calls per searching wave, and it is the serial decode of the ONE true
  //  header among the candidates that the call waits for: ~ 150 dependent steps on a wave with two or three neighbours per SIMD.)
  // length and symbol of the code in the low bits of `bits` (0: none -- cannot happen with a complete code)
  auto decode = [&](u32 bits, u32 &sym) -> u32 {
    const u32 r = __brev(bits) >> 25;  // the next seven bits, the first one read on top: a code of length l is r >> (7 - l)
    u32 len = 0, idx = 0;
#pragma unroll
    for (u32 l = 7; l >= 1; --l) {  // (downwards: the shortest length that fits is the one that stays)
      const u32 c = (r >> (7 - l)) - ((u32)(first >> (8 * l)) & 0xffu);
      const bool hit = c < ((u32)(cnt >> (8 * l)) & 0xffu);
      len = hit ? l : len;
      idx = hit ? ((u32)(offs >> (8 * l)) & 0xffu) + c : idx;
    }
    sym = len ? tab[idx] : 0u;
    return len;
  };
  const u32 total = hlit + hdist;
  const u64 end_bits = in_len * 8;
  u64 pos = q + 17 + 3 * ncl;
  u64 buf = 0;
  u32 have = 0, idx = 0, prev = 0, kl = 0, kd = 0, nd = 0;
  bool eob = false;
  while (idx < total) {
    if (have < 16) { if (pos + 16 > end_bits) return false; buf = sm_bits64(in, in_len, pos); have = 64; }
    u32 sym;
    u32 len = decode((u32)buf & 127u, sym);
    if (len == 0) return false;
    u32 rep = 1, val = sym;
    if (sym >= 16) {
      const u32 xb = sym == 16 ? 2u : (sym == 17 ? 3u : 7u);
      const u32 x = (u32)(buf >> len) & ((1u << xb) - 1);
      len += xb;
      if (sym == 16) { if (idx == 0) return false; rep = 3 + x; val = prev; }
      else { rep = (sym == 17 ? 3u : 11u) + x; val = 0; }
    }
    buf >>= len; have -= len; pos += len;
    if (idx + rep > total) return false;
    if (val) {
      const u32 nl = idx < hlit ? (idx + rep <= hlit ? rep : hlit - idx) : 0u;  // how many of them are literal/length codes
      kl += nl * (32768u >> val);
      kd += (rep - nl) * (32768u >> val);
      nd += rep - nl;
      if (kl > 32768u || kd > 32768u) return false;  // over-subscribed already: random bits die here after a few symbols
      if (idx <= 256 && idx + rep > 256) eob = true;
    }
    prev = val;
    idx += rep;
  }
  return eob && kl == 32768u && (kd == 32768u || nd <= 1);
}

// cand[k] (k >= 1): bit position of the first dynamic block header at or behind data_start + k * chunk_bytes
// (searched up to the next cut), ~0 if there is none.
// The range behind a cut is searched by `split` waves (equal parts, cand[k * split + part]); the host keeps the
// first find of each cut.
// The stream is read from LDS: a slab of SM_SLAB bytes (+ what a window at its last bit position needs) is staged
// with coalesced 16-byte loads.  Three filters, each on what the one before lets through:
//   0  bit-parallel, 2 048 positions a step (every lane a dword = 32 positions): BTYPE == 2 and HLIT, HDIST <= 29 are
//      a dozen shifts and ANDs on the dword and its successor -- 22 % of random positions pass; they are compacted
//      into a list (in stream order);
//   1  one listed position per lane: the code-length code is complete (Kraft sum of the 3-bit lengths == 1: five
//      look-ups in a table of four fields each) -- about one in 300;
//   2  sm_header_plausible, one per lane, 64 at a time.
// (Round 3's first version ran filter 1 on all 64 positions of a step: 3.5 of the 7.6 ms of a 256 MiB member.)
constexpr u32 SM_SLAB = 2048, SM_STEP = 2048;  // bytes per slab; bit positions per filter-0 step
// one wave (a device function: tests/emu/sm_find_emu.cc runs it on the CPU wave emulation)
// (15 KB a wave: ten waves per CU.  With 64 second-filter tables and 16-bit Kraft sums it was 23 KB -- six waves --
//  and the kernel is latency-bound.)
constexpr u32 SM_SECOND = 32;  // candidates the second filter takes at a time (a 128-byte table each)
struct SmFindLds {
  u8 cl_tab[SM_SECOND][128];  // per candidate of a second-filter call: its symbols sorted by code length (19 bytes used)
  u64 queue[128];
  u8 kraft4[4096];  // sum of 128 >> len over four 3-bit code-length fields (len 0 counts nothing), saturated at 255
  u32 slab[(SM_SLAB + 64) / 4];
  u16 list[SM_STEP];  // filter 0's survivors: bit offsets inside the slab
};
// first_part (device only; nullptr in the emulation): the lowest part of this cut that has found a block start so far.
// The host keeps the first find behind a cut, so a wave searching a LATER part of the same cut gives up as soon as an
// earlier part has one (its own find could not be used): about half of the finder's work on text.
AHIP_DEVINL u64 sm_find_wave(SmFindLds &S, const u8 *__restrict__ in, u64 in_len, u64 q0, u64 q1, const int lane,
                             u32 *first_part = nullptr, u32 part = 0, u64 *stats = nullptr) {
  for (u32 i = lane; i < 4096; i += 64) {
    u32 t = 0;
    for (u32 f = 0; f < 4; ++f) { const u32 l = (i >> (3 * f)) & 7; t += l ? (128u >> l) : 0u; }
    S.kraft4[i] = (u8)(t < 255 ? t : 255);  // (256 = four 1-bit codes: over the mark like everything above 128)
  }
  wave_sync();
  const u64 below = (1ull << lane) - 1;
  const u64 end_bits = in_len * 8;
  u64 found = ~0ull;
  u32 qn = 0;
  u64 slab_byte = ~0ull;  // stream byte of slab[0] (a multiple of 4)
  // (Where the finder's time goes, ablation builds on a 64 MiB member: 0.70 ms; 0.155 with this check compiled out -- and then
  //  every wave scans its whole part --, 0.06 without filter 1 as well.  Settling the queue slab by slab with the header's bits
  //  read from the slab in LDS instead of global memory was slower, 0.82 ms: the calls cost, not their memory accesses.)
  auto second = [&]() {  // sm_header_plausible on the first <= SM_SECOND queued positions; lowest position first
    const u32 nb = qn < SM_SECOND ? qn : SM_SECOND;
#ifdef AHIP_PROFILE
    const u64 t_in = __builtin_amdgcn_s_memtime();
#endif
    bool pass = false;
    if ((u32)lane < nb) pass = sm_header_plausible(in, in_len, S.queue[lane], S.cl_tab[lane]);
    const u64 pm = __ballot(pass);
    if (pm) found = S.queue[__builtin_ctzll(pm)];
    wave_sync();
    const u64 m0 = (u32)lane + nb < qn ? S.queue[lane + nb] : 0, m1 = (u32)lane + 64 + nb < qn ? S.queue[lane + 64 + nb] : 0;
    wave_sync();
    if ((u32)lane + nb < qn) S.queue[lane] = m0;
    if ((u32)lane + 64 + nb < qn) S.queue[lane + 64] = m1;
    qn -= nb;
    wave_sync();
#ifdef AHIP_PROFILE
    if (stats && lane == 0) { stats[0] += 1; stats[1] += nb; stats[2] += __builtin_amdgcn_s_memtime() - t_in; }
#endif
  };
  for (u64 base = q0 & ~31ull; found == ~0ull && base < q1; base += SM_STEP) {
    if (first_part && uniform(__atomic_load_n(first_part, __ATOMIC_RELAXED)) < part) return ~0ull;
    // the step's windows need bytes [base / 8, (base + SM_STEP - 1 + 17 + 64) / 8]: restage when they leave the slab
    if (slab_byte == ~0ull || ((base + SM_STEP - 1 + 17 + 64) >> 3) + 4 > slab_byte + SM_SLAB + 64) {
      slab_byte = (base >> 3) & ~3ull;
      wave_sync();
      for (u32 o = (u32)lane * 16; o < SM_SLAB + 64; o += 1024) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (slab_byte + o + 16 <= in_len) v = load_u128_unaligned(in + slab_byte + o);
        else {
          u32 w[4] = {0, 0, 0, 0};
          for (u32 b = 0; b < 16; ++b) if (slab_byte + o + b < in_len) w[b >> 2] |= (u32)in[slab_byte + o + b] << (8 * (b & 3));
          v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *(uint4 *)((u8 *)S.slab + o) = v;
      }
      wave_sync();
    }
    // ---- filter 0: this lane's dword, positions wq .. wq + 31 ----
    const u32 wi = (u32)((base - slab_byte * 8) >> 5) + (u32)lane;  // dword of the slab
    const u64 wq = base + 32ull * lane;
    const u64 v = (u64)S.slab[wi] | ((u64)S.slab[wi + 1] << 32);
    u32 m = (u32)(~(v >> 1) & (v >> 2));                                   // BTYPE == 2: bit 1 clear, bit 2 set
    m &= ~(u32)((v >> 4) & (v >> 5) & (v >> 6) & (v >> 7));                // HLIT <= 29: not 1111x
    m &= ~(u32)((v >> 9) & (v >> 10) & (v >> 11) & (v >> 12));             // HDIST <= 29
    // ---- a stored block on a byte boundary: BFINAL + BTYPE 00 + zero padding in one byte, then LEN and its complement ----
    // (the four bytes of this lane's dword; 2^-23 of random positions look like one: false finds die in S2 like false headers)
    u32 sj = 4;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const u64 x = v >> (8 * j);
      if (((u32)x & 0xfeu) == 0 && ((((u32)(x >> 8)) ^ ((u32)(x >> 24))) & 0xffffu) == 0xffffu) sj = (u32)j;
    }
    const u64 sq = wq + 8ull * sj;
    bool shit = sj < 4 && sq >= q0 && sq < q1 && sq + 40 <= end_bits;
    if (AHIP_ANY_HINT(shit)) {
      AHIP_ASM_NOTE("stored block candidate");
      // ... and the stored block it announces is followed by another one (same test where its data ends): the 2^-23 of
      // random positions that pass the first test become 2^-46 -- a false find inside a chunk costs that chunk the
      // tokens its sizing pass keeps (its area is cut short), so it has to be rare.  The last block of a stored stretch
      // is not found; a chunk simply starts one block earlier.
      if (shit) {
        const u64 h0 = sm_bits64(in, in_len, sq);
        const u64 q2 = sq + 40 + 8ull * (u32)((h0 >> 8) & 0xffffu);
        shit = q2 + 40 <= end_bits;
        if (shit) {
          const u64 h1 = sm_bits64(in, in_len, q2);
          shit = ((u32)h1 & 0xfeu) == 0 && ((((u32)(h1 >> 8)) ^ ((u32)(h1 >> 24))) & 0xffffu) == 0xffffu;
        }
      }
    }
    const u64 sm_ = __ballot(shit);
    const u64 stored_at = sm_ ? lane_bcast64(sq, __builtin_ctzll(sm_)) : ~0ull;  // (lanes ascend with the position)
    // inside [q0, q1) and with the 29 bits a header needs at least
    if (wq < q0) m &= q0 - wq >= 32 ? 0u : ~0u << (u32)(q0 - wq);
    const u64 lastp1 = end_bits >= 29 ? end_bits - 28 : 0, hi = q1 < lastp1 ? q1 : lastp1;  // first position not to test
    if (wq + 32 > hi) m &= wq >= hi ? 0u : ~0u >> (32 - (u32)(hi - wq));
    u32 total;
    u32 at = wave_excl_sum((u32)__builtin_popcount(m), total);
    const u32 rel0 = wi * 32;
    for (u32 mm = m; mm; mm &= mm - 1) S.list[at++] = (u16)(rel0 + (u32)__builtin_ctz(mm));
    wave_sync();
    // ---- filter 1, 64 listed positions at a time: a complete code-length code ----
    for (u32 b0 = 0; b0 < total && found == ~0ull; b0 += 64) {
      bool ok = b0 + (u32)lane < total;
      u32 rel = 0;
      if (ok) {
        rel = S.list[b0 + lane];
        const u32 di = rel >> 5, sh = rel & 31;
        const u32 d0 = S.slab[di], d1 = S.slab[di + 1], d2 = S.slab[di + 2], d3 = S.slab[di + 3];
        const u32 x0 = __builtin_amdgcn_alignbit(d1, d0, sh);                       // bits [q, q + 32)
        const u64 x = (u64)__builtin_amdgcn_alignbit(d2, d1, sh) | ((u64)__builtin_amdgcn_alignbit(d3, d2, sh) << 32);  // [q + 32, q + 96)
        u64 w = ((u64)x0 >> 17) | (x << 15);                                          // [q + 17, q + 81)
        const u32 ncl = ((x0 >> 13) & 15) + 4;
        // sum of 2^(7 - len) over the transmitted lengths == 2^7 (19 x 3 = 57 bits, five table look-ups)
        w &= (1ull << (3 * ncl)) - 1;
        const u32 kraft = (u32)S.kraft4[(u32)w & 4095] + S.kraft4[(u32)(w >> 12) & 4095] + S.kraft4[(u32)(w >> 24) & 4095] +
                          S.kraft4[(u32)(w >> 36) & 4095] + S.kraft4[(u32)(w >> 48) & 4095];
        ok = kraft == 128;
      }
      const u64 pm = __ballot(ok);
      if (ok) S.queue[qn + (u32)__popcll(pm & below)] = slab_byte * 8 + rel;
      qn += (u32)__popcll(pm);
      wave_sync();
      while (qn >= 64 && found == ~0ull) second();
    }
    if (stored_at != ~0ull) {  // nothing behind it matters: settle what is queued, the lower position wins
      while (qn && found == ~0ull) second();
      return found < stored_at ? found : stored_at;
    }
  }
  while (qn && found == ~0ull) second();
  return found;
}
#ifndef AHIP_HOST_EMU
__global__ __launch_bounds__(64) void sm_find_kernel(const u8 *__restrict__ in, u64 in_len, u64 data_start, u64 chunk_bytes,
                                                     u32 n_chunks, u32 split, u64 *__restrict__ cand) {
  __shared__ SmFindLds S;
  const int lane = threadIdx.x;
  // Part-major: the workgroups searching the FIRST part behind every cut are dispatched first, and by the time those of a
  // later part get a slot most cuts have their find (first_part): they leave at once.  (Cut-major, the parts of a cut ran
  // side by side and the early exit saved nothing.)
  const u32 k = blockIdx.x % (n_chunks - 1) + 1, part = blockIdx.x / (n_chunks - 1);
  if (part >= split) return;
  const u64 part_bits = chunk_bytes * 8 / split;
  const u64 q0 = (data_start + (u64)k * chunk_bytes) * 8 + part * part_bits;
  u32 *first_part = (u32 *)(cand + (u64)n_chunks * split * 4) + k;  // (behind the finds and the profile slots; set to "none" by the host)
#ifdef AHIP_PROFILE
  __shared__ u64 stats[4];
  if (lane < 4) stats[lane] = 0;
  const u64 t_all = __builtin_amdgcn_s_memtime();
  const u64 found = sm_find_wave(S, in, in_len, q0, q0 + part_bits, lane, first_part, part, stats);
  if (lane == 0) {  // per workgroup behind the finds: calls of the header check, candidates it took, its cycles (the host prints means)
    const u64 nn = (u64)n_chunks * split;
    cand[nn + blockIdx.x] = stats[0]; cand[2 * nn + blockIdx.x] = stats[1]; cand[3 * nn + blockIdx.x] = stats[2] | ((__builtin_amdgcn_s_memtime() - t_all) << 32);
  }
#else
  const u64 found = sm_find_wave(S, in, in_len, q0, q0 + part_bits, lane, first_part, part);
#endif
  if (lane == 0) {
    cand[(u64)k * split + part] = found;
    if (found != ~0ull) atomicMin(first_part, part);
  }
}
#endif

#ifndef AHIP_HOST_EMU  // (the emulation only runs the block finder of this file)
// Token area / run directory of candidate c laid out along the INPUT (tok_layout_in): the sizing pass keeps its tokens
// there, and the write pass resolves the chunks of the chain straight from them (one tokenizer pass instead of two).
// base: a rank that decodes only a range of the candidates (ahip_stream_split_*) counts input bytes and candidates from the
// first of its own, so that its areas fill a buffer sized for the range (all zero: the whole stream).
struct SmBase {
  u64 byte0; u32 cand0; u32 k0;  // first byte / first candidate of the range; k0: list entry 0 is entry k0 of the stream's
  u32 ways;                      // 0 / 1: an area reaches to the next candidate.  W > 1: candidate c keeps its tokens in buffer c % W
  u64 way_words, way_dirs;       //   (way_words token words, way_dirs directory entries each) and its area reaches to candidate c + W
};
// W ways: with cuts as close as the stream split makes them (16 KiB) the finder also returns starts that are not on the chain of
// blocks; the chunk in front of one decodes across it, and an area that ended there would be full (MR_FAR: tokenized again,
// + 0.9 ms for the whole rank).  The candidates that share a buffer are W apart: a chunk may run over W - 1 false starts.
AHIP_DEVINL void sm_layout_in(const u64 *cand_bits, u32 n_cand, u64 in_len, u32 c, const SmBase &base, u64 &toff, u32 &col_cap, u64 &doff, u32 &dir_cap) {
  const u32 W = base.ways > 1 ? base.ways : 1u;
  const u64 p0 = uniform64(cand_bits[c]) >> 3;
  const u64 p1 = c + W < n_cand ? (uniform64(cand_bits[c + W]) >> 3) + 1 : in_len;
  tok_layout_in(p0 - base.byte0, p1 > p0 ? p1 - p0 : 0, c - base.cand0, toff, col_cap, doff, dir_cap);
  if (W > 1) { const u32 way = (c - base.cand0) % W; toff += (u64)way * base.way_words; doff += (u64)way * base.way_dirs; }
}
// The next chunk of a workgroup: the next value of a device counter (lane 0's atomicAdd, like next_member of the member
// kernels), or -- next == nullptr -- what the grid's stride says (`strided`).
AHIP_DEVINL u32 sm_next_chunk(u32 *next, u32 strided, int lane) {
  if (!next) return strided;
  u32 k = 0;
  if (lane == 0) k = atomicAdd(next, 1u);
  return uniform(k);
}
// the tokenizer on chunks (persistent grid like inflate_tokenize_kernel).  lay_in = 0: chunk k's token area / run
// directory follow tok_layout(out_off, out_limit, k) (exact offsets known); lay_in = 1: a sizing pass over ALL candidates
// that keeps its tokens, laid out along the input -- of a rank's range of the candidates (base): list entry k is candidate
// base.cand0 + k.
__global__ __launch_bounds__(64) void sm_tokenize_kernel(const u8 *__restrict__ in, u64 in_len,
                                                        const ChunkDesc *__restrict__ chunks, u32 n_chunks,
                                                        const u64 *__restrict__ cand_bits, u32 n_cand,
                                                        u32 *__restrict__ tokens, DirEnt *__restrict__ dir,
                                                        MemberResult *__restrict__ results, u32 lay_in, u32 *__restrict__ next,
                                                        SmBase base) {
  __shared__ SmLds lds;
  const int lane = threadIdx.x;
  // chunks are handed out by a counter like the members of inflate_tokenize_kernel (next != nullptr), or by the grid's stride
  for (u32 k = sm_next_chunk(next, blockIdx.x, lane); k < n_chunks; k = sm_next_chunk(next, k + gridDim.x, lane)) {
    const ChunkDesc c = chunks[k];
    MemberDesc d;
    d.in_off = uniform64(c.start_bit) >> 3;
    d.out_off = uniform64(c.out_off);
    d.out_limit = uniform64(c.out_limit);
    d.expect_end = POS_UNKNOWN;
    d.in_end = 0;
    ChunkCtx cx{cand_bits, n_cand, (u32)uniform64(c.start_bit) & 7, uniform(c.hist), k + base.k0 == 0 ? 1u : 0u};  // (chunk 0 = the stream's first bytes)
    TokSink sk{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false};
    if (tokens) {
      u64 toff, doff;
      if (lay_in) { sm_layout_in(cand_bits, n_cand, in_len, k + base.cand0, base, toff, sk.col_cap, doff, sk.dir_cap); sk.sizing = true; }
      else tok_layout(d.out_off, d.out_limit, k, toff, sk.col_cap, doff, sk.dir_cap);
      sk.col_cap = uniform(sk.col_cap);
      sk.dir_cap = uniform(sk.dir_cap);
      sk.area = tokens + toff;
      sk.dir = dir + doff;
    }
    HeaderLds &hdr = *(HeaderLds *)((u8 *)lds.p.inbuf + 1024);
    inflate_member<false, true, true>(lds.w, hdr, &lds.p, in, in_len, d, (u8 *)nullptr, sk, results[k], lane, &cx);
  }
}

// tokens -> symbols.  lay_in = 1: chunk k's tokens are those the sizing pass kept for candidate chunks[k].pad.
__global__ __launch_bounds__(64) void sm_resolve_kernel(const u8 *__restrict__ in, u64 in_len, const ChunkDesc *__restrict__ chunks,
                                                       u32 n_chunks, u16 *sym, const u32 *__restrict__ tokens,
                                                       const DirEnt *__restrict__ dir, const MemberResult *__restrict__ results,
                                                       const u64 *__restrict__ cand_bits, u32 n_cand, u32 lay_in, u32 *__restrict__ err,
                                                       u32 *__restrict__ next, SmBase base) {
  __shared__ ResLdsT<u16> lds;
  const int lane = threadIdx.x;
  for (u32 k = sm_next_chunk(next, blockIdx.x, lane); k < n_chunks; k = sm_next_chunk(next, k + gridDim.x, lane)) {
    const u64 out_off = uniform64(chunks[k].out_off), out_limit = uniform64(chunks[k].out_limit);
    const u32 ndir = (u32)uniform64(results[k].tok_words);
    u64 toff, doff;
    u32 cc, dc;
    if (lay_in) sm_layout_in(cand_bits, n_cand, in_len, uniform(chunks[k].pad), base, toff, cc, doff, dc);
    else tok_layout(out_off, out_limit, k, toff, cc, doff, dc);
    u32 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!resolve_member<u16>(lds, in, tokens + toff, dir + doff, ndir, sym + out_off, cyc, lane) && lane == 0) atomicAdd(err, 1u);  // (never silent)
  }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for the prefetched global loads below
// (the threads of these kernels talk to each other through LDS alone).
AHIP_DEVINL void sm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Windows.  windows[k] = the 32 KiB of output that end with chunk k (bytes; positions before the stream start are
// unused).  Window k is a function of window k-1 (every element is a byte of chunk k or a look-up into window k-1),
// and such functions compose, so the chain is cut into groups:
//   A  sm_windows_group   every group walks its chunks with the group's (unknown) input window as markers:
//                         symbolic windows wsym[k], groups in parallel;
//   B  sm_windows_link    one workgroup makes the groups' last windows concrete, group after group;
//   C  sm_windows_apply   every chunk's symbolic window becomes bytes with its group's input window.
__global__ __launch_bounds__(1024) void sm_windows_group(const ChunkDesc *__restrict__ chunks, const MemberResult *__restrict__ results,
                                                         u32 n_chunks, u32 group_size, const u16 *__restrict__ sym,
                                                         u16 *__restrict__ wsym) {
  __shared__ u16 W[2][SM_WINDOW];
  const u32 tid = threadIdx.x, k0 = blockIdx.x * group_size;
  constexpr u32 PER = SM_WINDOW / 1024;  // 32 window elements per thread
  for (u32 j = tid; j < SM_WINDOW; j += 1024) W[1][j] = (u16)(SYM_MARK + j);  // identity: "element j of the group's input window"
  __syncthreads();
  for (u32 g = 0; g < group_size && k0 + g < n_chunks; ++g) {
    const u32 k = k0 + g;
    const u16 *prev = W[(g + 1) & 1];
    u16 *cur = W[g & 1];
    const u64 off = chunks[k].out_off, len = results[k].out_len;
    // window element j = output position (end - 32768 + j) of the stream; all loads first, they are independent
    // (issuing the next chunk's loads ahead as sm_windows_link does was measured slower here: the loads are conditional,
    //  and 64 symbols in registers per thread spill at 1 024 threads)
    u32 s[PER];
#pragma unroll
    for (u32 u = 0; u < PER; ++u) {
      const u32 j = tid + u * 1024;
      s[u] = len >= SM_WINDOW - j ? (u32)sym[off + len - (SM_WINDOW - j)] : 0xffffffffu;
    }
#pragma unroll
    for (u32 u = 0; u < PER; ++u) {
      const u32 j = tid + u * 1024;
      u16 v;
      if (s[u] != 0xffffffffu) v = s[u] < SYM_MARK ? (u16)s[u] : prev[s[u] - SYM_MARK];
      else v = prev[j + len];  // this chunk is shorter than the window: the rest slides over from the one before
      cur[j] = v;
      wsym[(u64)k * SM_WINDOW + j] = v;
    }
    __syncthreads();
  }
}
// gwin[g] = concrete window at the END of group g
// hist_win / hist0: the window in front of the whole stream -- element j = the byte 32768 - j in front of its first output
// byte; only its last hist0 elements exist (what EARLIER gzip members wrote into the shared output, quirk q8; 0 for a
// stream with an output of its own: nothing in front of it is ever referenced)
AHIP_DEVINL u8 sm_hist_byte(const u8 *hist_win, u32 hist0, u32 j) { return j >= SM_WINDOW - hist0 ? hist_win[j] : (u8)0; }
// stride: elements from one chunk's symbolic window to the next (SM_WINDOW; more where the maps sit in an exchange buffer)
__global__ __launch_bounds__(1024) void sm_windows_link(u32 n_chunks, u32 group_size, const u16 *__restrict__ wsym, u8 *__restrict__ gwin,
                                                        const u8 *hist_win, u32 hist0, u32 stride) {
  __shared__ u8 W[2][SM_WINDOW];
  const u32 tid = threadIdx.x, n_groups = (n_chunks + group_size - 1) / group_size;
  for (u32 j = tid; j < SM_WINDOW; j += 1024) W[1][j] = sm_hist_byte(hist_win, hist0, j);
  __syncthreads();
  constexpr u32 PER = SM_WINDOW / 1024;
  u32 s[PER], sn[PER];  // (the symbols of group g + 1 are on their way while group g is looked up: see sm_windows_group)
  auto last_of = [&](u32 g) -> u32 { return (g + 1) * group_size - 1 < n_chunks ? (g + 1) * group_size - 1 : n_chunks - 1; };
  auto request = [&](u32 g, u32 (&r)[PER]) {
    const u16 *ws = wsym + (u64)last_of(g) * stride;
#pragma unroll
    for (u32 u = 0; u < PER; ++u) r[u] = ws[tid + u * 1024];
  };
  if (n_groups) request(0, sn);
  for (u32 g = 0; g < n_groups; ++g) {
    const u8 *prev = W[(g + 1) & 1];
    u8 *cur = W[g & 1];
#pragma unroll
    for (u32 u = 0; u < PER; ++u) s[u] = sn[u];
    if (g + 1 < n_groups) request(g + 1, sn);
#pragma unroll
    for (u32 u = 0; u < PER; ++u) {
      const u32 j = tid + u * 1024;
      const u8 v = s[u] < SYM_MARK ? (u8)s[u] : prev[s[u] - SYM_MARK];
      cur[j] = v;
      gwin[(u64)g * SM_WINDOW + j] = v;
    }
    sm_lds_barrier();
  }
}
// The same link on SYMBOLS: gsym[g] = the window at the end of group g as a function of the window in front of group 0 (an
// element is a byte, or a marker into THAT window).  A rank that decodes a range of one stream's chunks
// (ahip_stream_split_*) does not know the bytes in front of its range until the ranks have exchanged these maps: its last
// one, gsym[n_groups - 1], is what it contributes.
__global__ __launch_bounds__(1024) void sm_windows_link_sym(u32 n_chunks, u32 group_size, const u16 *__restrict__ wsym, u16 *__restrict__ gsym) {
  __shared__ u16 W[2][SM_WINDOW];
  const u32 tid = threadIdx.x, n_groups = (n_chunks + group_size - 1) / group_size;
  for (u32 j = tid; j < SM_WINDOW; j += 1024) W[1][j] = (u16)(SYM_MARK + j);
  __syncthreads();
  constexpr u32 PER = SM_WINDOW / 1024;
  for (u32 g = 0; g < n_groups; ++g) {
    const u16 *prev = W[(g + 1) & 1];
    u16 *cur = W[g & 1];
    const u32 last = (g + 1) * group_size - 1 < n_chunks ? (g + 1) * group_size - 1 : n_chunks - 1;
    const u16 *ws = wsym + (u64)last * SM_WINDOW;
    u32 s[PER];
#pragma unroll
    for (u32 u = 0; u < PER; ++u) s[u] = ws[tid + u * 1024];
#pragma unroll
    for (u32 u = 0; u < PER; ++u) {
      const u32 j = tid + u * 1024;
      const u16 v = s[u] < SYM_MARK ? (u16)s[u] : prev[s[u] - SYM_MARK];
      cur[j] = v;
      gsym[(u64)g * SM_WINDOW + j] = v;
    }
    __syncthreads();
  }
  if (n_groups == 0) for (u32 j = tid; j < SM_WINDOW; j += 1024) gsym[j] = (u16)(SYM_MARK + j);  // no chunks: the identity
}
// (one workgroup per chunk; the group's input window is copied into LDS first -- the look-ups are scattered byte reads, see
//  sm_translate_kernel)
// entry (nullptr: none): the 32 KiB in front of chunk 0 where they are not the stream's history but another rank's output
__global__ __launch_bounds__(1024) void sm_windows_apply(u32 group_size, const u16 *__restrict__ wsym, const u8 *__restrict__ gwin,
                                                         u8 *__restrict__ windows, const u8 *hist_win, u32 hist0, const u8 *__restrict__ entry) {
  __shared__ u8 W[SM_WINDOW] __attribute__((aligned(16)));
  const u32 k = blockIdx.x, g = k / group_size, tid = threadIdx.x;
  const u8 *in_win = g ? gwin + (u64)(g - 1) * SM_WINDOW : entry;
  // this chunk's symbols, eight a thread and step (on their way while the window is copied)
  constexpr u32 PER = SM_WINDOW / 8 / 1024;  // 4
  uint4 q[PER];
#pragma unroll
  for (u32 u = 0; u < PER; ++u) q[u] = ((const uint4 *)(wsym + (u64)k * SM_WINDOW))[tid + u * 1024];
  if (in_win) for (u32 i = tid; i < SM_WINDOW / 16; i += 1024) ((uint4 *)W)[i] = ((const uint4 *)in_win)[i];
  else for (u32 i = tid; i < SM_WINDOW; i += 1024) W[i] = sm_hist_byte(hist_win, hist0, i);
  __syncthreads();
#pragma unroll
  for (u32 u = 0; u < PER; ++u) {
    const u32 x[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
    u32 b[8];
#pragma unroll
    for (u32 e = 0; e < 4; ++e) { b[2 * e] = x[e] & 0xffffu; b[2 * e + 1] = x[e] >> 16; }
    u32 gth[8];
#pragma unroll
    for (u32 e = 0; e < 8; ++e) gth[e] = W[b[e] & (SM_WINDOW - 1)];
#pragma unroll
    for (u32 e = 0; e < 8; ++e) b[e] = b[e] < SYM_MARK ? b[e] : gth[e];
    const u64 lo = (b[0] & 0xffu) | ((b[1] & 0xffu) << 8) | ((b[2] & 0xffu) << 16) | (b[3] << 24);
    const u64 hi = (b[4] & 0xffu) | ((b[5] & 0xffu) << 8) | ((b[6] & 0xffu) << 16) | (b[7] << 24);
    ((u64 *)(windows + (u64)k * SM_WINDOW))[tid + u * 1024] = lo | (hi << 32);
  }
}

// symbols -> bytes, every chunk with the window of the chunk before it.  Eight symbols a thread and step: one 16-byte load
// (the symbol array is the library's own: aligned; the vectors start where the chunk's offset reaches a multiple of 8),
// one 8-byte store (the output is the caller's pointer: whatever alignment it has).  The window look-ups are what the
// kernel's time goes to -- scattered byte reads, 64 different lines an instruction: each workgroup copies the window into
// LDS first (a few workgroups per chunk: the copy is a quarter of what the workgroup then translates) and looks up there.
// (A byte per thread and step, look-ups in global memory: 0.58 ms for 256 MiB; vectors alone: 0.51.)
__global__ __launch_bounds__(256) void sm_translate_kernel(const ChunkDesc *__restrict__ chunks, const MemberResult *__restrict__ results,
                                                           const u16 *__restrict__ sym, const u8 *__restrict__ windows,
                                                           u8 *__restrict__ out, const u8 *hist_win, u32 hist0, const u8 *__restrict__ entry) {
  __shared__ u8 W[SM_WINDOW] __attribute__((aligned(16)));
  const u32 k = blockIdx.y;
  const u64 off = chunks[k].out_off, len = results[k].out_len;
  const u8 *w = k ? windows + (u64)(k - 1) * SM_WINDOW : entry;  // chunk 0: the window in front of the stream (entry: of this rank's range)
  auto byte_of = [&](u32 s) -> u32 { return s < SYM_MARK ? (s & 0xffu) : (u32)(w ? w[s - SYM_MARK] : sm_hist_byte(hist_win, hist0, s - SYM_MARK)); };
  u64 head = (8 - (off & 7)) & 7;
  head = head < len ? head : len;
  const u64 nvec = (len - head) >> 3, tail0 = head + (nvec << 3);
  if ((u64)blockIdx.x * 256 >= nvec && blockIdx.x) return;  // (nothing for this workgroup: not even the copy)
  if (w) {
    for (u32 i = threadIdx.x; i < SM_WINDOW / 16; i += 256) ((uint4 *)W)[i] = ((const uint4 *)w)[i];
    __syncthreads();
  }
  if (blockIdx.x == 0) {  // the elements in front of the first and behind the last whole vector
    const u32 t = threadIdx.x;
    if (t < head) out[off + t] = (u8)byte_of(sym[off + t]);
    if (t >= 8 && tail0 + (t - 8) < len) out[off + tail0 + (t - 8)] = (u8)byte_of(sym[off + tail0 + (t - 8)]);
  }
  const uint4 *sv = (const uint4 *)(sym + off + head);
  u8 *ov = out + off + head;
  for (u64 v = (u64)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (u64)gridDim.x * 256) {
    const uint4 q = sv[v];
    const u32 x[4] = {q.x, q.y, q.z, q.w};
    u32 b[8];
#pragma unroll
    for (u32 e = 0; e < 4; ++e) { b[2 * e] = x[e] & 0xffffu; b[2 * e + 1] = x[e] >> 16; }
    if (((q.x | q.y | q.z | q.w) & (SYM_MARK * 0x10001u)) != 0) {  // (SYM_MARK is the symbols' top bit)
      if (w) {  // eight look-ups, all on their way before the first is used (a byte value looks up W[byte]: harmless)
        u32 g[8];
#pragma unroll
        for (u32 e = 0; e < 8; ++e) g[e] = W[b[e] & (SM_WINDOW - 1)];
#pragma unroll
        for (u32 e = 0; e < 8; ++e) b[e] = b[e] < SYM_MARK ? b[e] : g[e];
      } else {
#pragma unroll
        for (u32 e = 0; e < 8; ++e) b[e] = byte_of(b[e]);
      }
    }
    const u64 lo = (b[0] & 0xffu) | ((b[1] & 0xffu) << 8) | ((b[2] & 0xffu) << 16) | (b[3] << 24);
    const u64 hi = (b[4] & 0xffu) | ((b[5] & 0xffu) << 8) | ((b[6] & 0xffu) << 16) | (b[7] << 24);
    ((unaligned_u64 *)(ov + (v << 3)))->v = lo | (hi << 32);
  }
}

#endif  // AHIP_HOST_EMU

}  // namespace ahip
