// deflate_kernels.hpp -- DEFLATE encoder on gfx950.
//
// The reference encoder (/root/reference/lib/src/codecs/zlib/deflate.dart) is one sequential
// pass: hash-chain LZ77 with lazy matching (_deflateSlow :997-1118, _longestMatch :1120-1206),
// a symbol buffer (_trTally :531-568) and per-block Huffman coding (_trFlushBlock :747-807,
// _buildTree :2656, _compressBlock :571-614, _sendBits :487-499).  Its output bytes are not pinned
// by any reference test; what must hold is (a) the stream inflates to the input through the
// reference's Inflate and (b) the size stays within a stated tolerance of the reference's
// (<= + 5 % on the benchmark corpora at levels 1 / 6 / 9).
//
// Here the input is cut into independent 32 KiB chunks (each may still reference the 32 KiB of
// raw input before it) and every chunk becomes one DEFLATE block, closed -- except the last --
// by an empty stored block so that it ends on a byte boundary (the Z_SYNC_FLUSH marker the
// reference itself emits through _trStoredBlock(0, 0, false), deflate.dart:219):
//
//   D1 deflate_match_kernel   persistent workgroups (one per CU at levels 4-9: 1 024 threads and the CU's whole LDS),
//                             each taking a run of consecutive chunks.  The sliding window is a 36 KiB ring in LDS;
//                             the reference's hash-chain DEPTH (128 / 4 096 candidates) is replaced by CONTEXT: a
//                             4-way bucketed hash of 4-byte strings (the four most recent occurrences) plus one-way
//                             tables keyed by the hash of the 8- and of the 16-byte string at the position.  A step
//                             probes, then inserts, one position per thread; all seven candidates are verified and
//                             extended together, 8 bytes per LDS round trip, to 32 bytes (ties go on to nice_length).
//                             Insertion is an LDS atomicMax on a key that orders positions (the lowest position of
//                             the youngest step wins): two encodes of the same input are byte-identical.  The tables
//                             are carried from chunk to chunk (the window of chunk c + 1 is the window of chunk c moved
//                             on by 32 KiB).  Best (len, dist) per position goes to scratch.
//   D2 deflate_parse_kernel   one wave per chunk follows the reference's one-step lazy rule as the
//                             orbit of position 0 (63 positions a block, the visited ones found wave-wide by
//                             pointer doubling) and writes the token list; a capped match the parse lands on is
//                             extended by the whole wave.
//   D3 deflate_encode_kernel  one workgroup per chunk: symbol histogram (LDS atomics), zlib's
//                             heap Huffman construction with the 15/7-bit limit (restated from
//                             deflate.dart:2567-2784, run by one lane), the dynamic header's code-length
//                             tokens by the whole workgroup (df_cl_tokens: what the reference's run-length state
//                             machine makes of a run of L equal lengths is a closed form), then the block's
//                             tokens 256 at a time: workgroup prefix sum of code lengths -> bit
//                             offsets -> atomicOr into the LDS output image.  A chunk that does
//                             not shrink is emitted as a stored block.
//   D4 deflate_offsets_kernel + deflate_concat_kernel  exclusive scan of chunk sizes (device) -> byte-granular gather.
// Measured (config 3, 1 GiB of log text, level 6): 26.0 ms = 41.2 GB/s in, match 19.4 / parse 2.6 / encode 4.0 ms
// (profiles/r05_df_kernel_stats.md; round 4: 36 ms, 23.1 / 6.3 / 6.4 -- same bytes out).
#pragma once
#include "common.hpp"

namespace ahip {

constexpr u32 DF_CHUNK = 32768;          // bytes per chunk = one DEFLATE block
constexpr u32 DF_SLAB = DF_CHUNK + 512;  // per-chunk output slab (a stored block needs CHUNK + 5 + flush)
// Table shapes per level group (the reference's level table, deflate.dart:1253-1272, trades chain depth for speed; here
// the knob is what the LDS tables can index -- compressed size follows the ENTRY COUNT and the context length, not the
// associativity).  The instances archive_hip.hip launches (deflate_match_kernel<HASH_BITS, WAYS, LA, LB, SUB>):
//   levels 1-3: 4 096 buckets x 2 ways, no long-context tables, 256 positions a step (three workgroups per CU)
//   levels 4-7: 8 192 x 4 + 16 384 / 8 192 entries keyed by the 8- / 16-byte string, 1 024 positions a step (one workgroup per CU)
//   levels 8-9: the same tables, 512 positions a step (closer candidates are seen; AHIP_DF_SUB512=1 selects it for all levels)
constexpr u32 DF_MINLEN = 4;             // 4-byte hash: 3-byte matches are not searched
constexpr u32 DF_EMPTY = 0;              // a table slot nobody wrote
#ifndef AHIP_DF_CAP
#define AHIP_DF_CAP 32
#endif
constexpr u32 DF_CAP = AHIP_DF_CAP;      // the match kernel compares this far; the parse extends the matches it uses

struct DeflateParams {
  u64 n;        // input bytes
  u32 chunks;
  u32 lazy;     // 1 = one-step lazy evaluation (levels 4..9), 0 = greedy (levels 1..3)
  u32 store;    // 1 = level 0: stored blocks only
  u32 max_cmp;  // longest match searched (258)
  u32 max_dist; // farthest match: 2^windowBits - 262 (deflate.dart:1120-1131, MAX_DIST)
  u32 nice;     // candidates that still tie after DF_CAP bytes are compared on up to this length (the reference's
                // nice_length, deflate.dart:1253-1272: 128 at level 6, 258 at level 9); DF_CAP = never
  u32 open;     // 1 = this input is a SHARD of a longer one and not its last: its last chunk is not the stream's final
                // block either -- it ends with the byte-aligning empty stored block like every other chunk, so that the next
                // shard's stream can be spliced on behind it (ahip_deflate_shards)
};

template <u32 HB> AHIP_DEVINL u32 df_hash4(u32 w) { return (w * 2654435761u) >> (32 - HB); }
// Insertion into a table of 16-bit window positions, two to a dword, such that among the writers of one slot in one
// step a DEFINED one wins -- not whichever wave the hardware happens to serve last: two encodes of the same input are
// byte for byte the same (the reference is one sequential pass and trivially so, deflate.dart:997-1118).  A position is
// stored as key = pos ^ (SUB - 1) (SUB = positions per step, a power of two; 0 = empty: position SUB - 1 is never found):
// keys grow from step to step and, inside a step, towards the LOWER positions, so the largest key of a slot is the
// lowest position of the youngest step -- the one every later position of that step can still use as a candidate.
// Entry `e` is one half of dword e >> 1; `word` = that dword as read since the last barrier.  During a step only ONE
// half of any dword is written -- the ways of a bucket take turns step by step, the two halves of a long-string bucket
// alternate with the step's parity -- so every writer of the slot carries the same other half, and an LDS atomicMax on
// the whole dword compares just the keys.
AHIP_DEVINL void df_insert(u16 *tab, u32 e, u32 key, u32 word) {
  const u32 nw = (e & 1) ? ((word & 0xffffu) | (key << 16)) : ((word & 0xffff0000u) | key);
  atomicMax((u32 *)tab + (e >> 1), nw);
}
AHIP_DEVINL u32 df_word(const u16 *tab, u32 e) { return ((const u32 *)tab)[e >> 1]; }
// hashes of LONGER strings (8 and 16 bytes): a candidate that shares that much context is worth what a deep walk
// down the reference's hash chain finds (_longestMatch, deflate.dart:1120-1206, up to 128 / 4096 candidates)
template <u32 HB> AHIP_DEVINL u32 df_hash8(u64 w) { return (u32)(((w * 0x9E3779B97F4A7C15ull) >> 32) * 2654435761u) >> (32 - HB); }
template <u32 HB> AHIP_DEVINL u32 df_hash16(u64 a, u64 b) { return df_hash8<HB>(a * 0xC2B2AE3D27D4EB4Full + (b ^ (b >> 29))); }

// ------------------------------------------------------------------------------------------
// D1: per-position best match
// ------------------------------------------------------------------------------------------
// The sliding window lives in LDS: a 36 KiB ring (32 KiB of history + 1.25 KiB of lookahead + slack) whose
// first 272 bytes are mirrored behind its end, so a compare that starts near the end of the ring reads
// straight on.  Candidate verification and match extension are LDS reads (two aligned dwords + v_alignbyte
// per unaligned 4 bytes) instead of scattered global loads -- the first version spent its time in the
// address coalescer (64 different lines per load instruction).
constexpr u32 DF_RING = 36864, DF_MIRROR = 272, DF_AHEAD = 1024;
AHIP_DEVINL u32 df_rc(u32 x) { return x >= DF_RING ? x - DF_RING : x; }  // window position -> ring offset (x < 2 * DF_RING)
// (gfx950's LDS also serves unaligned 4- and 8-byte reads directly, tools/micro/lds_unaligned.hip -- one DS instruction
// instead of two / three aligned dwords + v_alignbyte -- but not faster:)
#ifdef AHIP_DF_UNALIGNED_READS  // measured SLOWER (config 3 level 6: 20.6 against 25.2 GB/s): unaligned 8-byte LDS reads are split by the hardware
AHIP_DEVINL u32 df_rd4(const u32 *ring, u32 r) { return ((const unaligned_u32 *)((const u8 *)ring + r))->v; }
AHIP_DEVINL u64 df_rd8(const u32 *ring, u32 r) { return ((const unaligned_u64 *)((const u8 *)ring + r))->v; }
#else
AHIP_DEVINL u32 df_rd4(const u32 *ring, u32 r) {
  const u32 i = r >> 2;
  return __builtin_amdgcn_alignbyte(ring[i + 1], ring[i], r & 3);
}
AHIP_DEVINL u64 df_rd8(const u32 *ring, u32 r) {
  const u32 i = r >> 2;
  const u32 a = ring[i], b = ring[i + 1], c = ring[i + 2];
  return (u64)__builtin_amdgcn_alignbyte(b, a, r & 3) | ((u64)__builtin_amdgcn_alignbyte(c, b, r & 3) << 32);
}
#endif
// Lengths of up to NW candidate matches at once.  rc[k] = ring offset of candidate k (alive[k]: there is one;
// a bucket-mate that does not even share 4 bytes ends with len < 4).  The candidates advance together, 8 bytes per round, so one
// round costs ONE LDS round trip for all of them.  The kernel issues VALU 84 % of the time (profiles/r05_experiments.md section 6)
// and three quarters of that is this loop, so a round is written for its instruction count: a candidate is a dword-aligned
// LDS address and a byte shift, both fixed (a round moves every address by 8); "alive" is a lane mask in a register (no
// branches around the candidates: dead ones compute along, masking them per lane or per candidate with a wave vote was
// measured slower); the first differing byte of 8 is v_ffbl on the two XORed dwords -- which answers -1 = "none" for
// equal dwords, so min3 with the bytes that are left needs no compare.  The lengths are what the straightforward loop gives.
AHIP_DEVINL u32 df_ffbl(u32 x) {  // v_ffbl_b32: index of the lowest set bit, 0xffffffff for 0 (__builtin_ctz(0) is undefined)
  u32 r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
typedef const u32 __attribute__((address_space(3))) *df_lds_u32p;
#ifndef AHIP_DF_ROUND
#define AHIP_DF_ROUND 8
#endif
constexpr u32 DF_ROUND = AHIP_DF_ROUND;  // bytes compared per round: 8 or 16 (the lengths are the same).  16 halves what is paid per round, but most
                                         // waves are done after one or two rounds of 8: measured slower (256 MiB of log text, level 6: 8.95 against 8.73 ms)
static_assert((DF_ROUND == 8 || DF_ROUND == 16) && DF_CAP % DF_ROUND == 0, "compare rounds");
template <int NW>
AHIP_DEVINL void df_match_lens(const u32 *ring, const u32 (&rc)[NW], u32 rp, u32 maxl, bool (&alive)[NW], u32 (&len)[NW], u32 l0 = 0) {
  // LDS byte addresses, made opaque: left to itself the compiler re-adds the ring's LDS offset (beyond a DS instruction's
  // 16-bit offset field) to every address in every round -- 24 v_add per round for what is one add per candidate
  const u32 rb = (u32)(uintptr_t)(const __attribute__((address_space(3))) u8 *)ring;
  constexpr u32 ND = DF_ROUND / 4;  // dwords compared per round
  u32 ca_[NW], sh[NW], am[NW];
  u32 anym = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    if (l0 == 0) len[k] = 0;  // (an alive candidate's len is l0 on entry: 0, or the cap its first pass ended on)
    ca_[k] = rb + (rc[k] & ~3u) + l0;
    AHIP_PIN(ca_[k]);
    sh[k] = rc[k] & 3u;
    am[k] = alive[k] ? ~0u : 0u;
    anym |= am[k];
  }
  u32 pa_ = rb + (rp & ~3u) + l0;
  AHIP_PIN(pa_);
  const u32 psh = rp & 3u;
  for (u32 l = l0, o = 0; anym; l += DF_ROUND, o += DF_ROUND) {  // the first round doubles as the check that the bucket-mate really shares 4 bytes
    const df_lds_u32p pp = (df_lds_u32p)(pa_ + o);
    u32 pw[ND + 1], p[ND];
#pragma unroll
    for (u32 d = 0; d <= ND; ++d) pw[d] = pp[d];
    u32 cw[NW][ND + 1];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const df_lds_u32p cp = (df_lds_u32p)(ca_[k] + o);
#pragma unroll
      for (u32 d = 0; d <= ND; ++d) cw[k][d] = cp[d];
    }
#pragma unroll
    for (u32 d = 0; d < ND; ++d) p[d] = __builtin_amdgcn_alignbyte(pw[d + 1], pw[d], psh);
    const u32 room = maxl - l;  // > 0 (maxl >= 4: the position has four bytes to hash)
    const u32 cap = (room < DF_ROUND ? room : DF_ROUND) * 8 + 7;  // first-difference bit index at which the bytes that count run out
    const u32 goon = room > DF_ROUND ? ~0u : 0u;
    anym = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      u32 t = cap;
#pragma unroll
      for (u32 d = 0; d < ND; d += 2) {  // (-1 | 32 stays -1: "these dwords are equal")
        const u32 f0 = df_ffbl(__builtin_amdgcn_alignbyte(cw[k][d + 1], cw[k][d], sh[k]) ^ p[d]) | (32u * d);
        const u32 f1 = df_ffbl(__builtin_amdgcn_alignbyte(cw[k][d + 2], cw[k][d + 1], sh[k]) ^ p[d + 1]) | (32u * (d + 1));
        const u32 m = f0 < f1 ? f0 : f1;
        t = t < m ? t : m;
      }
      const u32 n = (t >> 3) & am[k];  // bytes this round adds: min(equal bytes, DF_ROUND, room), nothing for a finished candidate
      len[k] += n;
      am[k] = n == DF_ROUND ? goon : 0u;
      anym |= am[k];
    }
  }
#pragma unroll
  for (int k = 0; k < NW; ++k) alive[k] = am[k] != 0;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding
// GLOBAL loads and stores (the match[] store of the previous step, the window prefetch), ~1-2 us each step.
AHIP_DEVINL void df_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// match[] holds len << 16 | dist per input position (0 = no match of >= 4 bytes)
// LA / LB: index bits of the one-way tables keyed by 8-byte / 16-byte strings (0 = no such table)
// SUB: threads of the workgroup = positions probed, then inserted, per step
template <u32 DF_HASH_BITS, u32 DF_WAYS, u32 LA = 0, u32 LB = 0, u32 SUB = 256>
__global__ __launch_bounds__(SUB) void deflate_match_kernel(const u8 *__restrict__ in, DeflateParams P,
                                                            u32 *__restrict__ match) {
  constexpr u32 DF_SUB = SUB;
  constexpr u32 AHEAD = SUB > 512 ? 2 * SUB : DF_AHEAD;  // bytes staged in front of the step
  static_assert(SUB % 64 == 0 && AHEAD % SUB == 0 && AHEAD >= 2 * SUB && DF_CHUNK + AHEAD + SUB <= DF_RING, "step geometry");
  __shared__ u16 tbl[(1u << DF_HASH_BITS) * DF_WAYS] __attribute__((aligned(4)));
  __shared__ u16 tblA[LA ? (1u << LA) : 2u] __attribute__((aligned(4)));
  __shared__ u16 tblB[LB ? (1u << LB) : 2u] __attribute__((aligned(4)));
  constexpr u32 KX = SUB - 1;  // position <-> key (df_insert)
  static_assert((SUB & (SUB - 1)) == 0, "positions per step: a power of two");
  constexpr u32 NX = (LA ? 1u : 0u) + (LB ? 1u : 0u);  // candidates from the long-string tables
  constexpr u32 MERGE = DF_WAYS < 4 ? DF_WAYS : 4;  // history steps inserted per barrier (distinct ways)
  __shared__ u32 ring[(DF_RING + DF_MIRROR) / 4 + 2];
  const u32 tid = threadIdx.x;
  if (P.store) return;
  // A workgroup takes a run of consecutive chunks.  The window of chunk c + 1 is the window of chunk c moved on by 32 KiB
  // (its history IS chunk c), so the tables chunk c leaves behind -- minus what has dropped out of the window, positions
  // counted from the new window's start, and the last fifteen positions of chunk c, which could not be hashed before the
  // bytes behind them were there -- are exactly the tables the history insertion below would build: only the first
  // chunk of a run inserts its history (it was ~15 % of the kernel); the output is the same whatever the runs are.
  const u32 per = (P.chunks + gridDim.x - 1) / gridDim.x;
  const u32 c_begin = blockIdx.x * per, c_end = c_begin + per < P.chunks ? c_begin + per : P.chunks;
  u32 ro = 0;        // ring offset of window position 0
  u64 wbase = 0;     // input offset of window position 0 (of the chunk before, when a new one starts)
  auto rc_of = [&](u32 x) -> u32 { x += ro; x = x >= 2 * DF_RING ? x - 2 * DF_RING : x; return x >= DF_RING ? x - DF_RING : x; };  // window position -> ring offset
  for (u32 chunk = c_begin; chunk < c_end; ++chunk) {
  const u64 cstart = (u64)chunk * DF_CHUNK;
  const u32 clen = (u32)((P.n - cstart) < DF_CHUNK ? (P.n - cstart) : DF_CHUNK);
  const u32 dict = cstart >= DF_CHUNK ? DF_CHUNK : (u32)cstart;  // raw bytes before the chunk that may be referenced
  const u8 *win = in + cstart - dict;                            // window base; positions are relative to it
  const u32 wlen = dict + clen;
  // bytes [q, q + 4) of the window (zero past the end) ...
  auto fetch = [&](u32 q) -> u32 {
    u32 v = 0;
    if (q + 4 <= wlen) v = load_u32_unaligned(win + q);
    else for (u32 k = 0; k < 4; ++k) v |= (q + k < wlen ? (u32)win[q + k] : 0u) << (8 * k);
    return v;
  };
  // ... into the ring
  auto put = [&](u32 q, u32 v) {
    const u32 r = rc_of(q);
    ring[r >> 2] = v;
    if (r < DF_MIRROR) ring[(DF_RING + r) >> 2] = v;
  };
  auto stage = [&](u32 q) { put(q, fetch(q)); };
  u32 base = 0;
  if (chunk == c_begin) {
    ro = 0;
    for (u32 i = tid; i < (1u << DF_HASH_BITS) * DF_WAYS; i += SUB) tbl[i] = (u16)DF_EMPTY;
    if (LA) for (u32 i = tid; i < (1u << LA); i += SUB) tblA[i] = (u16)DF_EMPTY;
    if (LB) for (u32 i = tid; i < (1u << LB); i += SUB) tblB[i] = (u16)DF_EMPTY;
    for (u32 q = 4 * tid; q < AHEAD; q += 4 * SUB) stage(q);  // [0, AHEAD)
    __syncthreads();
    // History before the chunk is only inserted.  Consecutive steps write different ways, so up to four are
    // done as one (same table as step by step, a quarter of the barriers).
    for (; base + MERGE * DF_SUB <= dict; base += MERGE * DF_SUB) {
      if (tid < MERGE * (SUB / 4)) stage(base + AHEAD + 4 * tid);
      __syncthreads();  // the last position's 4 bytes reach into what was just staged
      // the steps of one parity together (their slots are different dwords, or the same half of one), then the others
#pragma unroll
      for (u32 phase = 0; phase < 2; ++phase) {
#pragma unroll
        for (u32 k = phase; k < MERGE; k += 2) {
          const u32 p = base + k * DF_SUB + tid;  // p + 4 <= wlen: the chunk follows
          const u32 step = (base / DF_SUB) + k;
          const u32 e = df_hash4<DF_HASH_BITS>(df_rd4(ring, rc_of(p))) * DF_WAYS + (step & (DF_WAYS - 1));
          df_insert(tbl, e, p ^ KX, df_word(tbl, e));
          if (LA && p + 8 <= wlen) { const u32 eA = df_hash8<LA ? LA - 1 : 1>(df_rd8(ring, rc_of(p))) * 2 + (step & 1); df_insert(tblA, eA, p ^ KX, df_word(tblA, eA)); }
          if (LB && p + 16 <= wlen) { const u32 eB = df_hash16<LB ? LB - 1 : 1>(df_rd8(ring, rc_of(p)), df_rd8(ring, rc_of(p) + 8)) * 2 + (step & 1); df_insert(tblB, eB, p ^ KX, df_word(tblB, eB)); }
        }
        __syncthreads();
      }
    }
  } else {
    // carried over from the chunk before (a full one: it has a successor)
    __syncthreads();  // its last step's compares are done with the ring and the tables
    const u32 shift = (u32)((cstart - dict) - wbase);  // 0 behind the input's first chunk, else 32 KiB
    if (shift) {
      static_assert(SUB <= 32768 && DF_CHUNK == 32768, "a key's bit 15 is its position's: what drops out of the window has it clear");
      auto rebase = [&](u16 *t, u32 entries) {  // two keys a dword: positions below `shift` are forgotten, the others move down
        for (u32 i = tid; i < entries / 2; i += SUB) {
          const u32 x = ((const u32 *)t)[i];
          const u32 keep = ((x >> 15) & 0x00010001u) * 0xffffu;
          ((u32 *)t)[i] = x & keep & 0x7fff7fffu;
        }
      };
      rebase(tbl, (1u << DF_HASH_BITS) * DF_WAYS);
      if (LA) rebase(tblA, 1u << LA);
      if (LB) rebase(tblB, 1u << LB);
      ro = rc_of(shift);
    }
    __syncthreads();
    // the bytes behind the previous chunk's end were staged as zeros (there was nothing behind its window): the real ones,
    // from the last dword that reached over the end on
    for (u32 q = dict - 4 + 4 * tid; q < dict + AHEAD; q += 4 * SUB) stage(q);
    __syncthreads();
    if (tid < 16) {  // the positions the chunk before could not hash yet (4 / 8 / 16 bytes reached over its end), with their own step
      const u32 p = dict - 16 + tid, step = p / DF_SUB;
      if (p + 4 > dict) { const u32 e = df_hash4<DF_HASH_BITS>(df_rd4(ring, rc_of(p))) * DF_WAYS + (step & (DF_WAYS - 1)); df_insert(tbl, e, p ^ KX, df_word(tbl, e)); }
      if (LA && p + 8 > dict) { const u32 eA = df_hash8<LA ? LA - 1 : 1>(df_rd8(ring, rc_of(p))) * 2 + (step & 1); df_insert(tblA, eA, p ^ KX, df_word(tblA, eA)); }
      if (LB && p + 16 > dict) { const u32 eB = df_hash16<LB ? LB - 1 : 1>(df_rd8(ring, rc_of(p)), df_rd8(ring, rc_of(p) + 8)) * 2 + (step & 1); df_insert(tblB, eB, p ^ KX, df_word(tblB, eB)); }
    }
    __syncthreads();
    base = dict;
  }
  wbase = cstart - dict;
  // 256 positions per step.  A position is compared with the strings earlier steps left in its bucket
  // (read before this step's insertion) and with the string this step put into the bucket's current slot
  // if that one lies below it (distances under 256: runs and short periods) -- five candidates, one batch.
#ifdef AHIP_PROFILE
  u32 pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  AHIP_TICK(t_dict);
  // window prefetch: wave 0 loads the SUB bytes at base + AHEAD during one step and stores them into the
  // ring at the top of the next, so the load's latency is never waited for
  u32 pf = tid < SUB / 4 ? fetch(base + AHEAD + 4 * tid) : 0u;
  for (; base < wlen; base += DF_SUB) {
    AHIP_TICK(t0);
    if (tid < SUB / 4) {  // nobody reads these slots during this step
      put(base + AHEAD + 4 * tid, pf);
      pf = fetch(base + DF_SUB + AHEAD + 4 * tid);
    }
    const u32 p = base + tid;
    const bool has4 = p + 4 <= wlen;
    const bool search = has4 && p >= dict;
    const u32 rp = rc_of(p);
    u32 w = 0, h = 0, best_len = 0, best_dist = 0;
    // Compares stop at DF_CAP bytes: a wave waits for its longest compare, and neighbouring positions inside
    // one long match would each walk (nearly) all of it.  The parse extends the few matches it actually emits.
    u32 maxl = (wlen - p) < P.max_cmp ? (wlen - p) : P.max_cmp;
    maxl = maxl < DF_CAP ? maxl : DF_CAP;
    if (has4) { w = df_rd4(ring, rp); h = df_hash4<DF_HASH_BITS>(w); }
    u32 cand[DF_WAYS + 1 + NX];
    // the bucket's dwords (two ways each): the candidates, and the halves an insertion below has to carry along
    u32 bw[DF_WAYS / 2];
#pragma unroll
    for (u32 d = 0; d < DF_WAYS / 2; ++d) bw[d] = has4 ? ((const u32 *)tbl)[h * (DF_WAYS / 2) + d] : 0u;
#pragma unroll
    for (u32 way = 0; way < DF_WAYS; ++way) cand[way] = search ? ((bw[way >> 1] >> (16 * (way & 1))) & 0xffffu) : DF_EMPTY;  // (keys)
    const u32 step = base / DF_SUB;
    const u32 slot = h * DF_WAYS + (step & (DF_WAYS - 1));
    u32 eA = 0, eB = 0, wA = 0, wB = 0;
    const bool has8 = p + 8 <= wlen, has16 = p + 16 <= wlen;
    // the long-string tables: a bucket is one dword, its halves written in even / odd steps; the younger one is the candidate
    if (LA) {
      if (has8) { eA = df_hash8<LA ? LA - 1 : 1>(df_rd8(ring, rp)) * 2 + (step & 1); wA = df_word(tblA, eA); }
      cand[DF_WAYS + 1] = (search && has8) ? ((wA >> 16) > (wA & 0xffffu) ? (wA >> 16) : (wA & 0xffffu)) : DF_EMPTY;
    }
    if (LB) {
      if (has16) { eB = df_hash16<LB ? LB - 1 : 1>(df_rd8(ring, rp), df_rd8(ring, rp + 8)) * 2 + (step & 1); wB = df_word(tblB, eB); }
      cand[DF_WAYS + NX] = (search && has16) ? ((wB >> 16) > (wB & 0xffffu) ? (wB >> 16) : (wB & 0xffffu)) : DF_EMPTY;
    }
    AHIP_TICK(t1);
    df_lds_barrier();
    // (same-slot writers of one step: the largest position wins, df_insert -- not whoever is served last)
    if (has4) df_insert(tbl, slot, p ^ KX, bw[(step & (DF_WAYS - 1)) >> 1]);
    if (LA && has8) df_insert(tblA, eA, p ^ KX, wA);
    if (LB && has16) df_insert(tblB, eB, p ^ KX, wB);
    df_lds_barrier();
    AHIP_TICK(t2);
    AHIP_ACC(pc[0], t0, t1);
    AHIP_ACC(pc[1], t1, t2);
    if (search) {
      cand[DF_WAYS] = tbl[slot];
      constexpr u32 NC = DF_WAYS + 1 + NX;
      u32 rc[NC], dist[NC], len[NC];
      bool alive[NC];
      // (straight-line: a candidate `dist` back lies `dist` back in the ring too -- one wrap at most, dist < DF_RING -- and
      //  c < p && dist <= max_dist is one unsigned compare of dist - 1)
#pragma unroll
      for (u32 k = 0; k < NC; ++k) {
        const u32 c = cand[k] ^ KX;  // key -> position
        dist[k] = p - c;
        bool ok = cand[k] != DF_EMPTY && dist[k] - 1u < P.max_dist;
        if (k == DF_WAYS) ok = ok && c >= base;
        alive[k] = ok;
        const u32 back = rp - dist[k], wrapped = back + DF_RING;
        rc[k] = ok ? (back < wrapped ? back : wrapped) : rp;
      }
      AHIP_TICK(t3);
      AHIP_ACC(pc[2], t2, t3);
      df_match_lens<NC>(ring, rc, rp, maxl, alive, len);
      AHIP_TICK(t3b);
      AHIP_ACC(pc[6], t3, t3b);  // (the first pass alone; pc[3] has the tie-break pass as well)
      // Candidates that all reached the cap are told apart by comparing on (to `nice` bytes): the nearest one is not
      // the longest one on repetitive input.  Short periods (a tied candidate closer than the cap) are left alone --
      // the nearest candidate of a run is as good as any, and every position of the run would walk it again.
      const u32 room = (wlen - p) < P.max_cmp ? (wlen - p) : P.max_cmp;
      if (P.nice > DF_CAP && maxl == DF_CAP && room > DF_CAP) {
        u32 tied = 0, near = 0xffffffffu;
#pragma unroll
        for (u32 k = 0; k < NC; ++k) {
          const bool t = len[k] == DF_CAP;
          tied += t ? 1u : 0u;
          near = (t && dist[k] < near) ? dist[k] : near;
          alive[k] = t;
        }
        if (tied >= 2 && near >= DF_CAP) df_match_lens<NC>(ring, rc, rp, room < P.nice ? room : P.nice, alive, len, DF_CAP);
      }
      AHIP_TICK(t4);
      AHIP_ACC(pc[3], t3, t4);
#pragma unroll
      for (u32 k = 0; k < NC; ++k)
        if (k != DF_WAYS && (len[k] > best_len || (len[k] == best_len && len[k] && dist[k] < best_dist))) { best_len = len[k]; best_dist = dist[k]; }
      if (len[DF_WAYS] > best_len) { best_len = len[DF_WAYS]; best_dist = dist[DF_WAYS]; }
    }
    if (p >= dict && p < wlen) match[cstart + (p - dict)] = best_len >= DF_MINLEN ? ((best_len << 16) | best_dist) : 0u;
    AHIP_TICK(t5);
    AHIP_ACC(pc[4], t2, t5);
  }
#ifdef AHIP_PROFILE
  AHIP_TICK(t_end);
  AHIP_ACC(pc[5], t_dict, t_end);
  if (tid == 0) for (int k = 0; k < 8; ++k) match[P.n + 16 + (u64)chunk * 8 + k] = pc[k];
#endif
  }  // chunk
}

// ------------------------------------------------------------------------------------------
// D2: parse -- one wave per chunk
// ------------------------------------------------------------------------------------------
// The reference's decision at a position depends only on the matches at that position and the next
// (one-step lazy rule), so every position has a well-defined "next position"; the parse is the orbit of
// position 0.  64 positions at a time: the lanes compute step and token, a scalar loop hops through the
// block with v_readlane, and the visited lanes write their tokens compacted.
// true length (<= maxl) of a match known to hold for DF_CAP bytes: one wave-wide step, 4 bytes per lane.
// df_same4: how many of the nb (<= 4) bytes at a and at s are the same from the front (a dword each when all four count);
// df_extend_len: the length that follows from every lane's count.
AHIP_DEVINL u32 df_same4(const u8 *a, const u8 *s, u32 nb) {
#ifdef AHIP_ABL_NO_EXTEND  // dev ablation (wrong bytes): what the parse's trips to memory cost
  return nb;
#endif
  if (nb == 4) {
    const u32 x = load_u32_unaligned(a) ^ load_u32_unaligned(s);
    return x ? (u32)__builtin_ctz(x) >> 3 : 4u;
  }
  u32 same = nb;
  for (u32 k = nb; k-- > 0;)
    if (a[k] != s[k]) same = k;
  return same;
}
AHIP_DEVINL u32 df_extend_len(u32 same, bool in_range, u32 maxl) {
  const u64 stop = __ballot(in_range && same < 4);  // a mismatch, or the end of the allowed range, inside this lane
  if (!stop) return maxl;
  const int f = __ffsll((long long)stop) - 1;
  return DF_CAP + 4 * (u32)f + lane_bcast(same, f);
}
AHIP_DEVINL u32 df_extend(const u8 *a, u32 dist, u32 maxl, u32 lane) {
  const u32 o = DF_CAP + 4 * lane;
  u32 same = 4;
  if (o < maxl) same = df_same4(a + o, a + o - dist, maxl - o < 4 ? maxl - o : 4);  // (a - dist + o: a[x - dist] with unsigned x would wrap)
  return df_extend_len(same, o < maxl, maxl);
}
// the match at a (dist0 back) and the one at a + 1 (dist1 back) together: their loads are on the way at the same time --
// the parse waits for global memory here, and a position it lands on inside a long match has a long match behind it
// (Extending, block by block, the FIRST position of every run of capped matches with one distance and counting down from it
//  for the others -- no memory access left in the hop loop -- gave the same bytes and was SLOWER, 5.0 against 4.15 ms per GiB:
//  most such runs lie inside a match the parse takes and are never landed on.)
AHIP_DEVINL void df_extend2(const u8 *a, u32 dist0, u32 maxl0, u32 dist1, u32 maxl1, u32 lane, u32 &L0, u32 &L1) {
  const u32 o = DF_CAP + 4 * lane;
  u32 same0 = 4, same1 = 4;
  if (o < maxl0) same0 = df_same4(a + o, a + o - dist0, maxl0 - o < 4 ? maxl0 - o : 4);
  if (o < maxl1) same1 = df_same4(a + 1 + o, a + 1 + o - dist1, maxl1 - o < 4 ? maxl1 - o : 4);
  L0 = df_extend_len(same0, o < maxl0, maxl0);
  L1 = df_extend_len(same1, o < maxl1, maxl1);
}

__global__ __launch_bounds__(64) void deflate_parse_kernel(const u8 *__restrict__ in, DeflateParams P,
                                                           const u32 *__restrict__ match, u32 *__restrict__ tok,
                                                           u32 *__restrict__ ntok) {
  const u32 chunk = blockIdx.x, lane = threadIdx.x;
  if (chunk >= P.chunks) return;
  const u64 cstart = (u64)chunk * DF_CHUNK;
  const u32 clen = (u32)((P.n - cstart) < DF_CHUNK ? (P.n - cstart) : DF_CHUNK);
  u32 *t = tok + cstart;  // at most one token per byte
  u32 k = 0, pos = 0;
  if (!P.store) {
    const u64 below = (1ull << lane) - 1;
    // 63 positions a block: lane 63 only looks ahead (it is lane 0 of the next block), so the match at i + 1 is always the
    // neighbour lane's.  A block's matches and bytes are asked for one block ahead and not touched before the next round:
    // the wave does nothing but wait for them otherwise -- 512 blocks a chunk, a trip to memory each.  (Taking lane 63's
    // neighbour from the block ahead made every round wait for the loads it had just issued.)
    constexpr u32 STRIDE = 63;
    u32 m_next = lane < clen ? match[cstart + lane] : 0u;
    u32 b_next = lane < clen ? (u32)in[cstart + lane] : 0u;
    for (u32 base = 0; base < clen; base += STRIDE) {
      const u32 i = base + lane;
      const bool inb = i < clen && lane < STRIDE;
      const u32 m = m_next, byte = b_next;
      m_next = i + STRIDE < clen ? match[cstart + i + STRIDE] : 0u;
      b_next = i + STRIDE < clen ? (u32)in[cstart + i + STRIDE] : 0u;
      const u32 m1 = lane_gather(m, (lane + 1) & 63);  // (0 behind the chunk's end: m is; lane 63 is not used)
      const u32 l = m >> 16;  // <= min(DF_CAP, clen - i) by construction
      bool is_match = inb && l >= DF_MINLEN;
      if (is_match && P.lazy && (m1 >> 16) > l) is_match = false;  // a longer match starts at the next byte
      u32 step = is_match ? l : 1u;
      u32 token = is_match ? m : (0x80000000u | byte);
      const bool capped = inb && l >= DF_CAP;  // true length unknown (the tie-break pass may have taken l beyond the cap: still not final): settled when (if) the parse lands here
      u64 visited = 0;
      const u64 cap_mask = __ballot(capped);
      const u32 lim = base + STRIDE < clen ? base + STRIDE : clen;
      // Which positions the parse visits = the orbit of its entry position under "next position".  Followed hop by hop (a scalar
      // loop of ~ 12 instructions a hop, ~ 12 hops a block) that was the kernel's time: it is bound by instruction issue at its 8
      // waves per SIMD (no memory access left in the loop: an ablation without the extends costs the same).  Wave-wide instead:
      // with J = "the lane I hop to" (a capped match and the block's end hop to themselves), the set S of visited lanes grows
      // by S |= J^(2^k)(S) for k = 0 .. 5 -- the positions after t < 2^(k+1) hops are those after t' < 2^k hops and those 2^k
      // hops further -- one ds_permute (lanes of S send a 1 to their target) and one ds_bpermute (J doubled) a round.
      // (A run of literals as ONE hop of the scalar loop gave the same bytes and was slower, 5.5 against 4.2 ms per GiB.)
      const bool absorb = !inb || capped;  // (lanes behind the block's or the chunk's end are not positions of this block)
      while (pos < lim) {  // once; once more behind every capped match the parse lands on
        const u32 entry = pos - base;
        u32 J = absorb ? lane : (lane + step < 63u ? lane + step : 63u);
        u32 S = lane == entry ? 1u : 0u;
#pragma unroll
        for (u32 r = 0; r < 6; ++r) {
          const u32 got = (u32)__builtin_amdgcn_ds_permute((int)((S ? J : entry) << 2), 1);  // (everybody else re-marks the entry)
          S |= got;
          if (r < 5) J = lane_gather(J, J);
        }
        const u64 hit = __ballot(S != 0);
        const u64 vis = hit & ~__ballot(absorb);
        const u64 cap_hit = hit & cap_mask;  // the capped match the orbit ends on, if it does
        visited |= vis;
        if (!cap_hit) {  // left the block: behind the last position visited
          const int jl = 63 - __builtin_clzll(vis);
          pos = base + (u32)jl + lane_bcast(step, jl);
        } else {
          const u32 j = (u32)__builtin_ctzll(cap_hit);
          visited |= 1ull << j;
          pos = base + j;
          const u8 *a = in + cstart + pos;
          const u32 rem = clen - pos;
          const u32 d0 = lane_bcast(m, (int)j) & 0xffff;
          const u32 max0 = rem < P.max_cmp ? rem : P.max_cmp;
          u32 L0;
          bool take = true;
          if (P.lazy && pos + 1 < clen) {
            const u32 mm1 = lane_bcast(m1, (int)j);
            u32 L1 = mm1 >> 16;
            if (L1 >= DF_CAP) df_extend2(a, d0, max0, mm1 & 0xffff, (rem - 1) < P.max_cmp ? (rem - 1) : P.max_cmp, lane, L0, L1);
            else L0 = df_extend(a, d0, max0, lane);
            take = !(L1 > L0);
          } else L0 = df_extend(a, d0, max0, lane);
          if (lane == j) token = take ? ((L0 << 16) | d0) : (0x80000000u | byte);
          pos += take ? L0 : 1u;
        }
      }
      if ((visited >> lane) & 1) t[k + (u32)__popcll(visited & below)] = token;
      k += (u32)__popcll(visited);
    }
  }
  if (lane == 0) ntok[chunk] = k;
}

// ------------------------------------------------------------------------------------------
// D3: Huffman coding of one chunk
// ------------------------------------------------------------------------------------------
constexpr int DF_LCODES = 286, DF_DCODES = 30, DF_BLCODES = 19, DF_HEAP = 573;

struct EncLds {
  u32 fl[288], fd[32];
  u16 ltree[DF_HEAP * 2], dtree[(2 * DF_DCODES + 1) * 2], bltree[(2 * DF_BLCODES + 1) * 2];
  u32 keys[512], iw[512];  // sort keys freq << 9 | symbol; weights of the internal nodes
  u16 par[1024];           // parent of leaf r (sorted rank) / of internal node m + k
  u32 bl_count[16];
  u32 t_m, t_kraft;
  int t_maxcode;
  u32 opt_len;
  u32 wsum[4];
  u32 wsum2[2][8];  // the token rounds' bit counts per wave and half, two sets taking turns
  u32 run_bits;
  u16 cl_tok[DF_LCODES + DF_DCODES + 8];  // the code lengths of both trees as run-length tokens: symbol | extra value << 5
  u16 cl_pos[320];                          // df_cl_tokens: tokens in front of the run that starts at i
  u64 cl_start[5];                          //               bit i: a run of equal lengths starts at i (or i is the end)
  u32 cl_ntok;
  u32 cl_freq[32];                          // how often each of the 19 code-length symbols occurs among them
  u32 obuf[DF_SLAB / 4] __attribute__((aligned(16)));
};

__device__ const u8 k_extra_lbits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const u8 k_extra_dbits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const u8 k_extra_blbits[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
__device__ const u8 k_bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// length (3..258) -> (code 0..28, extra bit count, extra value); distance (1..32768) likewise
AHIP_DEVINL void df_len_code(u32 len, u32 &code, u32 &xb, u32 &xv) {
  const u32 lc = len - 3;
  if (lc < 8) { code = lc; xb = 0; xv = 0; return; }
  if (lc == 255) { code = 28; xb = 0; xv = 0; return; }
  const u32 k = 31 - (u32)__builtin_clz(lc);
  code = (k - 1) * 4 + ((lc >> (k - 2)) & 3);
  xb = k - 2;
  xv = lc & ((1u << xb) - 1);
}
AHIP_DEVINL void df_dist_code(u32 dist, u32 &code, u32 &xb, u32 &xv) {
  const u32 d = dist - 1;
  if (d < 4) { code = d; xb = 0; xv = 0; return; }
  const u32 k = 31 - (u32)__builtin_clz(d);
  code = 2 * k + ((d >> (k - 1)) & 1);
  xb = k - 1;
  xv = d & ((1u << xb) - 1);
}

AHIP_DEVINL u32 df_bi_reverse(u32 code, int len) { return __brev(code) >> (32 - len); }

// Code lengths and codes of one Huffman tree, built by the whole workgroup (256 threads).
//
// The reference (deflate.dart:2567-2784 = zlib's build_tree / gen_bitlen / gen_codes) is a heap algorithm run
// by one thread; on a single lane its ~10^4 dependent LDS accesses cost ~0.3 ms per block and were 83 % of this
// kernel.  Compressed BYTES are not pinned by the reference, only validity and size, so the
// same optimal lengths are reached a parallel way:
//   sort (freq, symbol) ascending (bitonic, in LDS)  ->  two-queue Huffman merge (one lane, ~2 LDS reads per
//   step, the only serial part)  ->  leaf depths by walking parents (a thread per leaf)  ->  clip to the length
//   limit and repair the Kraft sum with zlib's own move (one leaf from the deepest non-full level down, one
//   from the limit up next to it)  ->  hand the multiset of lengths out by frequency rank  ->  canonical codes
//   (zlib's gen_codes order: by symbol index within a length, bit-reversed).
// Like zlib, a tree with fewer than two used symbols gets dummy ones so that the code is complete.
template <int NP>
__device__ inline void df_build_tree_wg(EncLds &E, u16 *tree, int elems, int max_length, int &max_code_out, u32 tid) {
  u32 *keys = E.keys;
  // ---- keys + used count ----
  if (tid == 0) { E.t_m = 0; E.t_maxcode = -1; E.t_kraft = 0; }
  for (u32 i = tid; i < 16; i += 256) E.bl_count[i] = 0;
  __syncthreads();
  for (u32 i = tid; i < (u32)NP; i += 256) {
    const bool used = i < (u32)elems && tree[i * 2] != 0;
    keys[i] = used ? (((u32)tree[i * 2] << 9) | i) : 0xffffffffu;
    if (i < (u32)elems) tree[i * 2 + 1] = 0;
    if (used) { atomicAdd(&E.t_m, 1u); atomicMax(&E.t_maxcode, (int)i); }
  }
  __syncthreads();
  if (tid == 0) {
    while (E.t_m < 2) {  // zlib: force at least two codes of non-zero frequency
      const int node = E.t_maxcode < 2 ? ++E.t_maxcode : 0;
      tree[node * 2] = 1;
      keys[node] = (1u << 9) | (u32)node;
      E.t_m++;
    }
  }
  __syncthreads();
  const int m = (int)E.t_m;
  max_code_out = E.t_maxcode;
  // ---- bitonic sort, ascending (passes with j <= 64 stay inside the 128 keys a wave owns; giving only the others the
  //      workgroup's barrier changed nothing: the tree's time is the serial merge below) ----
  for (u32 k = 2; k <= (u32)NP; k <<= 1) {
    for (u32 j = k >> 1; j > 0; j >>= 1) {
      for (u32 t = tid; t < (u32)NP / 2; t += 256) {
        const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
        const u32 l = i | j;
        const u32 a = keys[i], b = keys[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[l] = a; }
      }
      __syncthreads();
    }
  }
  // ---- two-queue merge: leaves 0..m-1 (sorted), internal nodes m..2m-2 in creation (= weight) order ----
  // (one lane, a dependent LDS read per take.  Wave 0 running the loop as scalar code with 64 weights of either queue in a
  //  register across its lanes -- v_readlane for the heads, nothing read from LDS in the steady state -- was measured SLOWER:
  //  98 K cycles per chunk for the literal/length tree against 80 K; the loop turns into two dozen scalar branches a step)
  if (tid == 0) {
    u32 *iw = E.iw;
    u16 *par = E.par;
    int li = 0, ii = 0, ic = 0;
    u32 lw = keys[0] >> 9, nw = 0xffffffffu;  // heads of the two queues
    for (int step = 0; step < m - 1; ++step) {
      u32 w2 = 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bool leaf = li < m && (ii >= ic || lw <= nw);  // ties: the leaf first (smaller depth)
        if (leaf) { par[li] = (u16)(m + ic); w2 += lw; ++li; lw = li < m ? keys[li] >> 9 : 0xffffffffu; }
        else { par[m + ii] = (u16)(m + ic); w2 += nw; ++ii; nw = ii < ic ? iw[ii] : 0xffffffffu; }
      }
      iw[ic] = w2;
      if (ii == ic) nw = w2;  // the internal queue was empty: the new node is its head
      ++ic;
    }
  }
  __syncthreads();
  // ---- leaf depths (thread per leaf), clipped; level histogram and Kraft sum ----
  const int root = 2 * m - 2;
  for (int r = (int)tid; r < m; r += 256) {
    int d = 0, x = r;
    while (x != root) { x = E.par[x]; ++d; }
    if (d > max_length) d = max_length;
    if (d == 0) d = 1;  // (m >= 2, so this cannot happen; keeps a lone code at length 1)
    atomicAdd(&E.bl_count[d], 1u);
    atomicAdd(&E.t_kraft, 1u << (max_length - d));
  }
  __syncthreads();
  if (tid == 0) {
    int excess = (int)E.t_kraft - (1 << max_length);
    while (excess > 0 && E.bl_count[max_length] > 0) {
      int bits = max_length - 1;
      while (bits > 0 && E.bl_count[bits] == 0) bits--;
      if (bits == 0) break;
      E.bl_count[bits]--;
      E.bl_count[bits + 1] += 2;
      E.bl_count[max_length]--;
      excess--;
    }
  }
  __syncthreads();
  // ---- lengths by frequency rank: the rarest symbols take the longest codes ----
  for (int r = (int)tid; r < m; r += 256) {
    int run = 0, len = 1;
    for (int bits = max_length; bits >= 1; --bits) {
      const int c = (int)E.bl_count[bits];
      if (r < run + c) { len = bits; break; }
      run += c;
    }
    tree[(keys[r] & 511) * 2 + 1] = (u16)len;
  }
  __syncthreads();
  // ---- canonical codes ----
  for (int n = (int)tid; n < elems; n += 256) {
    const int len = tree[n * 2 + 1];
    if (len == 0) continue;
    u32 code = 0;
    for (int bits = 1; bits <= len; ++bits) code = (code + E.bl_count[bits - 1]) << 1;  // next_code[len]
    u32 rank = 0;
    for (int t = 0; t < n; ++t) rank += tree[t * 2 + 1] == len;
    tree[n * 2] = (u16)df_bi_reverse(code + rank, len);
  }
  __syncthreads();
}

// serial LSB-first bit writer into the LDS output image (one lane, before the parallel phase)
struct DfBits { u32 *buf; u32 pos; };
AHIP_DEVINL void df_put(DfBits &b, u32 v, u32 n) {
  if (!n) return;
  const u32 w = b.pos >> 5, s = b.pos & 31;
  b.buf[w] |= v << s;
  if (s + n > 32) b.buf[w + 1] |= v >> (32 - s);
  b.pos += n;
}
AHIP_DEVINL void df_put_code(DfBits &b, const u16 *tree, int c) { df_put(b, tree[c * 2], tree[c * 2 + 1]); }

// The code lengths of one tree as the run-length tokens of the block header (deflate.dart:2820-2920 = zlib's scan_tree /
// send_tree: symbols 0..15 a length, 16 = repeat the previous length 3..6 times, 17 / 18 = 3..10 / 11..138 zeros), by the
// whole workgroup.  The reference's loop is a small state machine (count, max_count, min_count, prevlen); what it does with
// a maximal run of L equal lengths v is a closed form of (v, L) alone -- a run starts with prevlen != v and the counts the
// length in front of it left: (7, 4), or (138, 3) for zeros --
//   v != 0:  the first 7 (or all L <= 7): fewer than 4 -> that many literals; else v, REP(count - 1).  Behind them groups
//            of 6 -> REP(6); a rest of 3..5 -> REP(rest), of 1..2 -> literals;
//   v == 0:  groups of 138 -> REPZ_11_138(138); a rest of 11.. -> REPZ_11_138, of 3..10 -> REPZ_3_10, of 1..2 -> literals --
// so a thread per run writes its tokens where a prefix sum of the token counts says, and one pass serves both the
// frequencies of the code-length tree (scan_tree) and, once that tree is built, the emission (send_tree).  One lane
// walking the 316 lengths twice was 153 K of the kernel's 425 K cycles per chunk.
// Appends the tokens of tree[0 .. max_code] to E.cl_tok at E.cl_ntok (all threads call; barriers inside).
__device__ inline void df_cl_tokens(EncLds &E, const u16 *tree, int max_code, u32 tid) {
  const int n = max_code + 1;
  const u32 base = E.cl_ntok;
  __syncthreads();
  // where the runs start, a bit each (and one at n): a run's length is the distance to the next bit -- a lane counting
  // along its run (up to 138 dependent LDS reads, twice per run) was 57 K of the kernel's 336 K cycles per chunk
  for (int i = (int)tid; i < 320; i += 256) {  // (whole waves: tid < 64 come round twice)
    const bool st = i == n || (i < n && (i == 0 || tree[i * 2 + 1] != tree[(i - 1) * 2 + 1]));
    const u64 bm = __ballot(st);
    if ((tid & 63) == 0) E.cl_start[i >> 6] = bm;
  }
  __syncthreads();
  // tokens of the run that starts at i (0 elsewhere)
  auto run_tokens = [&](int i, int &L) -> u32 {
    const u32 v = tree[i * 2 + 1];
    u32 w = (u32)i >> 6;
    u64 mk = E.cl_start[w] & ~((2ull << (i & 63)) - 1);  // starts behind i in its word
    while (!mk) mk = E.cl_start[++w];                     // (the bit at n <= 286 ends the search)
    L = (int)(w * 64 + (u32)__builtin_ctzll(mk)) - i;
    if (v != 0) {
      const int first = L < 7 ? L : 7;
      u32 t = first < 4 ? (u32)first : 2u;
      int rest = L - first;
      t += (u32)(rest / 6);
      rest %= 6;
      t += rest >= 3 ? 1u : (u32)rest;
      return t;
    }
    u32 t = (u32)(L / 138);
    const int rest = L % 138;
    t += rest >= 3 ? 1u : (u32)rest;
    return t;
  };
  for (int i = (int)tid; i < 320; i += 256) {
    u32 t = 0;
    if (i < n && (i == 0 || tree[i * 2 + 1] != tree[(i - 1) * 2 + 1])) { int L; t = run_tokens(i, L); }
    E.cl_pos[i] = (u16)t;
  }
  __syncthreads();
  if (tid < 64) {  // exclusive prefix sum over the 320 entries, five a lane
    u32 v[5], sum = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { v[k] = E.cl_pos[tid * 5 + k]; sum += v[k]; }
    u32 total;
    u32 ex = wave_excl_sum(sum, total);
#pragma unroll
    for (int k = 0; k < 5; ++k) { E.cl_pos[tid * 5 + k] = (u16)ex; ex += v[k]; }
    if (tid == 0) E.cl_ntok = base + total;
  }
  __syncthreads();
  for (int i = (int)tid; i < n; i += 256) {
    if (!(i == 0 || tree[i * 2 + 1] != tree[(i - 1) * 2 + 1])) continue;
    int L;
    (void)run_tokens(i, L);
    const u32 v = tree[i * 2 + 1];
    u16 *o = E.cl_tok + base + E.cl_pos[i];
    if (v != 0) {
      const int first = L < 7 ? L : 7;
      if (first < 4) { for (int k = 0; k < first; ++k) *o++ = (u16)v; }
      else { *o++ = (u16)v; *o++ = (u16)(16u | ((u32)(first - 1 - 3) << 5)); }
      int rest = L - first;
      for (; rest >= 6; rest -= 6) *o++ = (u16)(16u | (3u << 5));
      if (rest >= 3) *o++ = (u16)(16u | ((u32)(rest - 3) << 5));
      else for (int k = 0; k < rest; ++k) *o++ = (u16)v;
    } else {
      int rest = L;
      for (; rest >= 138; rest -= 138) *o++ = (u16)(18u | (127u << 5));
      if (rest >= 11) *o++ = (u16)(18u | ((u32)(rest - 11) << 5));
      else if (rest >= 3) *o++ = (u16)(17u | ((u32)(rest - 3) << 5));
      else for (int k = 0; k < rest; ++k) *o++ = 0;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void deflate_encode_kernel(const u8 *__restrict__ in, DeflateParams P,
                                                             const u32 *__restrict__ tok, const u32 *__restrict__ ntok,
                                                             u8 *__restrict__ slabs, u32 *__restrict__ csize) {
  __shared__ EncLds E;
  const u32 chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 cstart = (u64)chunk * DF_CHUNK;
  const u32 clen = (u32)((P.n - cstart) < DF_CHUNK ? (P.n - cstart) : DF_CHUNK);
  const bool last = chunk + 1 == P.chunks && !P.open;
  const u32 nt = ntok[chunk];
  const u32 *t = tok + cstart;
  u8 *slab = slabs + (u64)chunk * DF_SLAB;
  AHIP_TICK(e0);
  static_assert(DF_SLAB % 16 == 0, "the output image is cleared 16 bytes at a time");
  for (u32 i = tid; i < DF_SLAB / 16; i += 256) ((uint4 *)E.obuf)[i] = make_uint4(0u, 0u, 0u, 0u);
  for (u32 i = tid; i < 288; i += 256) E.fl[i] = 0;
  if (tid < 32) { E.fd[tid] = 0; E.cl_freq[tid] = 0; }
  __syncthreads();
  bool stored = P.store != 0;
  u32 total_bits = 0;
  if (!stored) {
    // ---- histogram (four tokens a thread on their way at a time) ----
    for (u32 i = tid; i < nt; i += 1024) {
      u32 v4[4];
#pragma unroll
      for (u32 u = 0; u < 4; ++u) v4[u] = i + 256 * u < nt ? t[i + 256 * u] : 0u;
#pragma unroll
      for (u32 u = 0; u < 4; ++u) {
        const u32 v = v4[u];
        if (i + 256 * u >= nt) continue;
        if (v >> 31) atomicAdd(&E.fl[v & 0xff], 1u);
        else {
          u32 c, xb, xv;
          df_len_code(v >> 16, c, xb, xv);
          atomicAdd(&E.fl[257 + c], 1u);
          df_dist_code(v & 0xffff, c, xb, xv);
          atomicAdd(&E.fd[c], 1u);
        }
      }
    }
    __syncthreads();
    for (u32 i = tid; i < DF_HEAP * 2; i += 256) E.ltree[i] = 0;
    for (u32 i = tid; i < (2 * DF_DCODES + 1) * 2; i += 256) E.dtree[i] = 0;
    for (u32 i = tid; i < (2 * DF_BLCODES + 1) * 2; i += 256) E.bltree[i] = 0;
    __syncthreads();
    for (u32 i = tid; i < 286; i += 256) E.ltree[i * 2] = (u16)E.fl[i];
    if (tid < 30) E.dtree[tid * 2] = (u16)E.fd[tid];
    __syncthreads();
    AHIP_TICK(e1);
    // ---- trees (whole workgroup), then run-length scan of the lengths and the header (one lane) ----
    if (tid == 0) E.ltree[256 * 2] = 1;  // end of block
    __syncthreads();
    int l_max_code, d_max_code, b_max_code;
    df_build_tree_wg<512>(E, E.ltree, DF_LCODES, 15, l_max_code, tid);
    AHIP_TICK(e1a);
    df_build_tree_wg<32>(E, E.dtree, DF_DCODES, 15, d_max_code, tid);
    AHIP_TICK(e1b);
    // the code lengths of both trees as run-length tokens (each tree a stream of its own), and their frequencies
    if (tid == 0) E.cl_ntok = 0;
    __syncthreads();
    df_cl_tokens(E, E.ltree, l_max_code, tid);
    df_cl_tokens(E, E.dtree, d_max_code, tid);
    const u32 n_cl = E.cl_ntok;
    for (u32 i = tid; i < n_cl; i += 256) atomicAdd(&E.cl_freq[E.cl_tok[i] & 31u], 1u);
    __syncthreads();
    if (tid < DF_BLCODES) E.bltree[tid * 2] = (u16)E.cl_freq[tid];
    __syncthreads();
    AHIP_TICK(e1c);
    df_build_tree_wg<32>(E, E.bltree, DF_BLCODES, 7, b_max_code, tid);
    AHIP_TICK(e1d);
    if (tid == 0) {
      int max_blindex;
      for (max_blindex = DF_BLCODES - 1; max_blindex >= 3; max_blindex--)
        if (E.bltree[k_bl_order[max_blindex] * 2 + 1] != 0) break;
      DfBits b{E.obuf, 0};
      df_put(b, (2u << 1) | (last ? 1u : 0u), 3);  // dynamic block, BFINAL on the last chunk
      df_put(b, (u32)(l_max_code + 1 - 257), 5);
      df_put(b, (u32)(d_max_code + 1 - 1), 5);
      df_put(b, (u32)(max_blindex + 1 - 4), 4);
      for (int r = 0; r <= max_blindex; r++) df_put(b, E.bltree[k_bl_order[r] * 2 + 1], 3);
      E.run_bits = b.pos;
    }
    __syncthreads();
    // the tokens' codes, 256 a round: bit offsets by a prefix sum, the bits by atomicOr (like the block's own tokens below)
    for (u32 base = 0; base < n_cl; base += 256) {
      const u32 i = base + tid;
      u32 bits = 0, nb = 0;
      if (i < n_cl) {
        const u32 tk = E.cl_tok[i], sym = tk & 31u;
        bits = E.bltree[sym * 2];
        nb = E.bltree[sym * 2 + 1];
        bits |= (tk >> 5) << nb;
        nb += k_extra_blbits[sym];
      }
      u32 wtotal;
      const u32 inc = wave_excl_sum(nb, wtotal);
      if (lane == 0) E.wsum[wave] = wtotal;
      __syncthreads();
      u32 off = E.run_bits + inc;
      for (u32 w = 0; w < wave; ++w) off += E.wsum[w];
      if (nb) {
        const u32 wi = off >> 5, sft = off & 31;
        const u64 lo = (u64)bits << sft;
        atomicOr(&E.obuf[wi], (u32)lo);
        if ((u32)(lo >> 32)) atomicOr(&E.obuf[wi + 1], (u32)(lo >> 32));
      }
      __syncthreads();
      if (tid == 0) E.run_bits += E.wsum[0] + E.wsum[1] + E.wsum[2] + E.wsum[3];
      __syncthreads();
    }
    AHIP_TICK(e2);
    // ---- exact size first: a chunk that does not shrink (or would not fit the LDS image) is stored ----
    // (from the histograms, like the reference's opt_len, deflate.dart:2567-2648: a symbol costs its code length plus its
    //  extra bits however often it occurs -- a second pass over the tokens was 37 K of the kernel's 336 K cycles per chunk)
    {
      u32 mybits = 0;
      for (u32 i = tid; i < 286; i += 256) mybits += E.fl[i] * ((u32)E.ltree[i * 2 + 1] + (i >= 257 ? (u32)k_extra_lbits[i - 257] : 0u));
      if (tid < 30) mybits += E.fd[tid] * ((u32)E.dtree[tid * 2 + 1] + (u32)k_extra_dbits[tid]);
      if (tid == 0) E.opt_len = 0;
      __syncthreads();
      atomicAdd(&E.opt_len, mybits);
      __syncthreads();
      const u32 need = (E.run_bits + E.opt_len + E.ltree[256 * 2 + 1] + 3 + 7 + 32 + 7) / 8;
      if (need > clen + 5 || need > DF_SLAB - 16) stored = true;
    }
    AHIP_TICK(e3);
#ifdef AHIP_PROFILE
    if (tid == 0) { u32 *pc = (u32 *)(slab + DF_SLAB - 32); pc[0] = (u32)((e1 - e0) >> 4); pc[1] = (u32)((e2 - e1) >> 4); pc[2] = (u32)((e3 - e2) >> 4);
                    pc[4] = (u32)((e1a - e1) >> 4); pc[5] = (u32)((e1c - e1b) >> 4); pc[6] = (u32)((e1d - e1c) >> 4); pc[7] = (u32)((e2 - e1d) >> 4); }
#endif
    if (!stored) {
    // ---- tokens, 512 per round (two halves of 256, a token of each per thread): the next round's tokens are asked for before
    //      this one's are worked on, and ONE barrier a round (the waves' bit counts alternate between two sets of slots;
    //      everybody keeps the running offset) ----
    auto token_bits = [&](u32 v, u64 &bits, u32 &nb) {
      if (v >> 31) {
        const u32 s = v & 0xff;
        bits = E.ltree[s * 2];
        nb = E.ltree[s * 2 + 1];
      } else {
        u32 c, xb, xv;
        df_len_code(v >> 16, c, xb, xv);
        bits = E.ltree[(257 + c) * 2];
        nb = E.ltree[(257 + c) * 2 + 1];
        bits |= (u64)xv << nb; nb += xb;
        df_dist_code(v & 0xffff, c, xb, xv);
        bits |= (u64)E.dtree[c * 2] << nb; nb += E.dtree[c * 2 + 1];
        bits |= (u64)xv << nb; nb += xb;
      }
    };
    auto deposit = [&](u32 off, u64 bits, u32 nb) {
      if (!nb) return;
      const u32 wi = off >> 5, s = off & 31;
      const u64 lo = bits << s;
      atomicOr(&E.obuf[wi], (u32)lo);
      if ((u32)(lo >> 32)) atomicOr(&E.obuf[wi + 1], (u32)(lo >> 32));
      if (s && (bits >> (64 - s))) atomicOr(&E.obuf[wi + 2], (u32)(bits >> (64 - s)));
    };
    u32 run = E.run_bits;
    u32 va_next = tid < nt ? t[tid] : 0u, vb_next = tid + 256 < nt ? t[tid + 256] : 0u;
    for (u32 base = 0, par = 0; base < nt; base += 512, par ^= 1) {
      const u32 ia = base + tid, ib = ia + 256;
      const u32 va = va_next, vb = vb_next;
      va_next = ia + 512 < nt ? t[ia + 512] : 0u;
      vb_next = ib + 512 < nt ? t[ib + 512] : 0u;
      u64 bits_a = 0, bits_b = 0;
      u32 nb_a = 0, nb_b = 0;
      if (ia < nt) token_bits(va, bits_a, nb_a);
      if (ib < nt) token_bits(vb, bits_b, nb_b);
      const u32 inc_a = wave_incl_sum(nb_a), inc_b = wave_incl_sum(nb_b);
      if (lane == 63) { E.wsum2[par][wave] = inc_a; E.wsum2[par][4 + wave] = inc_b; }
      __syncthreads();
      u32 ws[8];
#pragma unroll
      for (u32 w = 0; w < 8; ++w) ws[w] = E.wsum2[par][w];
      u32 off_a = run + inc_a - nb_a, off_b = run + ws[0] + ws[1] + ws[2] + ws[3] + inc_b - nb_b;
#pragma unroll
      for (u32 w = 0; w < 4; ++w) { off_a += w < wave ? ws[w] : 0u; off_b += w < wave ? ws[4 + w] : 0u; }
#pragma unroll
      for (u32 w = 0; w < 8; ++w) run += ws[w];
      deposit(off_a, bits_a, nb_a);
      deposit(off_b, bits_b, nb_b);
    }
    __syncthreads();  // (the last round's bits are in the image)
    if (tid == 0) E.run_bits = run;
    // ---- end of block, then the byte-aligning empty stored block (not after the last chunk) ----
    if (tid == 0) {
      DfBits b{E.obuf, E.run_bits};
      df_put_code(b, E.ltree, 256);
      if (!last) {
        df_put(b, 0, 3);                    // BFINAL=0, BTYPE=00
        b.pos = (b.pos + 7) & ~7u;          // _biWindup
        df_put(b, 0x0000, 16);
        df_put(b, 0xffff, 16);
      } else {
        b.pos = (b.pos + 7) & ~7u;
      }
      E.run_bits = b.pos;
    }
    __syncthreads();
    total_bits = E.run_bits;
#ifdef AHIP_PROFILE
    { AHIP_TICK(e4); if (tid == 0) { u32 *pc = (u32 *)(slab + DF_SLAB - 32); pc[3] = (u32)((e4 - e3) >> 4); } }
#endif
    }
  }
  if (stored) {
    // one stored block: header byte, LEN, NLEN, raw bytes (clen <= 32768 < 65536); byte aligned by itself
    if (tid == 0) {
      slab[0] = last ? 1 : 0;
      slab[1] = (u8)clen; slab[2] = (u8)(clen >> 8);
      slab[3] = (u8)~clen; slab[4] = (u8)(~clen >> 8);
      csize[chunk] = clen + 5;
    }
    for (u32 i = tid; i < clen; i += 256) slab[5 + i] = in[cstart + i];
    return;
  }
  const u32 nbytes = total_bits / 8;
  for (u32 i = tid; i < (nbytes + 3) / 4; i += 256) ((u32 *)slab)[i] = E.obuf[i];
  if (tid == 0) csize[chunk] = nbytes;
}

// D4a: where every chunk's bytes go -- the exclusive scan of the chunk sizes, on the device (one workgroup: a thread sums a
// run of consecutive chunks, the workgroup scans the 1 024 sums in LDS), so that the encode, the scan and the gather queue
// up behind one another without the host in between.  total[0] = the size of the whole stream.
__global__ __launch_bounds__(1024) void deflate_offsets_kernel(const u32 *__restrict__ csize, u32 chunks, u64 *__restrict__ coff,
                                                               u64 *__restrict__ total) {
  __shared__ u64 part[1024];
  const u32 tid = threadIdx.x, per = (chunks + 1023) / 1024;
  const u32 lo = min(chunks, tid * per), hi = min(chunks, lo + per);
  u64 sum = 0;
  for (u32 i = lo; i < hi; ++i) sum += csize[i];
  part[tid] = sum;
  __syncthreads();
  for (u32 d = 1; d < 1024; d <<= 1) {
    const u64 add = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  u64 at = part[tid] - sum;
  for (u32 i = lo; i < hi; ++i) { coff[i] = at; at += csize[i]; }
  if (tid == 1023) total[0] = part[1023];
}

// D4: gather the chunk slabs into the contiguous output (nothing is written behind cap: the host reports AHIP_E_CAP from
// the total)
__global__ __launch_bounds__(256) void deflate_concat_kernel(const u8 *__restrict__ slabs, const u32 *__restrict__ csize,
                                                             const u64 *__restrict__ coff, u8 *__restrict__ out, u64 cap) {
  const u32 chunk = blockIdx.x;
  const u8 *src = slabs + (u64)chunk * DF_SLAB;
  const u32 n = csize[chunk];
  if (coff[chunk] + n > cap) return;
  u8 *dst = out + coff[chunk];
  for (u32 i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

}  // namespace ahip
