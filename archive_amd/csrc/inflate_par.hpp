// inflate_par.hpp -- wave-parallel decode of ONE Huffman-coded DEFLATE block.
//
// The reference decodes a block strictly symbol by symbol (inflate.dart:300-343).  A wave that
// does the same keeps 63 of its 64 lanes idle and is bound by the LDS/HBM latency of every
// symbol.  Here all 64 lanes decode the SAME block at once:
//
//   window      the next WIN_BITS of the bitstream are staged into LDS with coalesced loads and
//               cut into 64 subsequences of SUB_BITS; lane i owns subsequence i.
//   pass A      lane 0 starts at the known token boundary; every other lane starts blind at the
//               first bit of its subsequence and decodes until it crosses into the next one.
//               Huffman streams self-synchronise, so most blind lanes end on a true boundary.
//   pass B..    lane i restarts from the end position lane i-1 reported, counts tokens and
//               output bytes and records its tokens (one coalesced row store per step into a
//               per-workgroup slab in device scratch).  Lanes up to the first one whose end
//               position changed are final (their start was true); the rest repeat.
//   gather      wave prefix sums of the per-lane counts give every lane its slot in the LDS token
//               queue; whole lanes are taken in batches that fit the queue / output window.
//   resolve     pass 1: wave scans of token lengths give each token its output offset (the
//               literal/match boundary scan).  pass 2: 64 output BYTES at a time, one per lane:
//               a start-slot scatter + wave prefix-max finds the token covering each byte; the
//               byte is a literal, an earlier byte of the LDS window, a byte of already flushed
//               output (L2-resident), or -- runs and very short distances -- the byte of a lower
//               lane, resolved by pointer doubling.  The window is flushed to HBM with coalesced
//               16-byte stores.
//
// Any anomaly (bad symbol on the true path, back-reference before the member start, output
// window exhausted, input too close to its end for unchecked reads) drops to the serial
// decoder of inflate_wave.hpp at the start of the current window; that path reproduces the
// reference's behaviour for malformed data exactly, so the parallel path only ever commits
// tokens of well-formed data.  Results are bit-identical by construction: both paths produce
// the same token sequence and writeBackReference semantics.
#pragma once
#include "inflate_wave.hpp"

namespace ahip {

#ifndef AHIP_SUB_BITS
#define AHIP_SUB_BITS 512
#endif
#ifndef AHIP_TOK_CAP
#define AHIP_TOK_CAP 1024
#endif
#ifndef AHIP_OB_CAP
#define AHIP_OB_CAP 4608
#endif
#ifndef AHIP_SLAB_ROWS
#define AHIP_SLAB_ROWS 192
#endif
constexpr int SUB_BITS = AHIP_SUB_BITS;          // bits per lane subsequence
constexpr int WIN_BITS = 64 * SUB_BITS;          // compressed bits per window (2 KiB at 256)
constexpr int IN_DWORDS = WIN_BITS / 32 + 8;     // + slack: a token may run 48 bits past the window (multiple of 4)
constexpr int TOK_CAP = AHIP_TOK_CAP;            // token queue entries per resolve batch
constexpr int OB_CAP = AHIP_OB_CAP;              // output bytes assembled in LDS per resolve batch
constexpr int SLAB_ROWS = AHIP_SLAB_ROWS;        // tokens one lane may record per window
constexpr int SLAB_WORDS = SLAB_ROWS * 64;       // per-workgroup token slab in device scratch (L2-resident)

constexpr u32 TK_LIT = 0x80000000u;  // | byte
constexpr u32 TK_EOB = 0x40000000u;
constexpr u32 TK_ERR = 0x20000000u;
// match: len << 16 | dist   (len <= 258, dist <= 32768)

// LDS of the tokenizer (next to WaveLds): the staged bitstream window
struct TokLds {
  u32 inbuf[IN_DWORDS] __attribute__((aligned(16)));
};
// LDS of the resolver: token queue, output window, start-slot rows
// E = u8: bytes.  E = u16: symbols of the chunked single-stream decode -- a byte value, or 0x8000 + j for "byte j
// of the 32 KiB of output in front of this chunk" (not known yet when the chunk is resolved).
template <typename E>
struct ParLdsT {
  u32 tok[TOK_CAP];
  E obuf[OB_CAP + 32] __attribute__((aligned(16)));
  u32 slot[3 * 64];  // two alternating 64-entry start-slot rows + one dump row
  u32 misc[4];       // [0] index of the first stored-run record in the current batch
};
using ParLds = ParLdsT<u8>;
constexpr u32 SYM_MARK = 0x8000;
// history element at window-relative offset si < 0; opos = elements of this chunk in front of the window
template <typename E>
AHIP_DEVINL u32 hist_get(const E *hist, i32 si, u64 opos) {
  if (sizeof(E) == 1) return hist[si];
  const i64 a = (i64)opos + si;
  return a >= 0 ? (u32)hist[si] : (u32)(SYM_MARK + 32768 + a);
}

constexpr u32 TK_STORED = 0x60000000u;  // | len (3..65535), followed by two words: absolute input byte offset lo, hi

// Append-only token stream of one member in device memory (tokenizer -> resolver hand-off).
// A member never needs more words than it has output bytes: every literal/match token covers
// >= 1 byte and a stored run of >= 3 bytes takes 3 words (shorter runs are emitted as literals).
struct TokSink {
  u32 *base;  // nullptr: sizing run, nothing is stored
  u64 w;      // words written
};

struct ParStats { u32 windows, rounds, fallbacks, partial; u32 cyc[8]; u32 dbg; };

// Per-lane LSB-first bit reader over the LDS window.  The stream continues at bit `sh` of the
// 64-bit pair (hi:lo); `nextw` is the dword after `hi`, always already requested from LDS, so a
// refill never waits on the critical path.  v_alignbit_b32 extracts 32 stream bits in one op.
struct LaneBits { u32 lo, hi, nextw, sh, ptr; };
AHIP_DEVINL void lb_init(LaneBits &d, const u32 *inbuf, u32 p) {
  u32 w = p >> 5;
  d.lo = inbuf[w];
  d.hi = inbuf[w + 1];
  d.nextw = inbuf[w + 2];
  d.ptr = w + 2;
  d.sh = p & 31;
}
// bring sh below 32 (branch-free) and re-request the look-ahead dword
AHIP_DEVINL void lb_normalize(LaneBits &d, const u32 *inbuf) {
  const bool adv = d.sh >= 32;
  d.lo = adv ? d.hi : d.lo;
  d.hi = adv ? d.nextw : d.hi;
  d.ptr += adv ? 1u : 0u;
  d.sh &= 31;
  d.nextw = inbuf[d.ptr];
}
AHIP_DEVINL u32 lb_peek32(const LaneBits &d) { return __builtin_amdgcn_alignbit(d.hi, d.lo, d.sh); }
AHIP_DEVINL u32 lb_pos(const LaneBits &d) { return (d.ptr - 2) * 32 + d.sh; }

// Wave-uniform description of the codes LONGER than a primary table, held in scalar registers
// for the whole block: first canonical code, symbol count and sorted-symbol offset of each
// length root+1 .. root+N.  Resolving a long code then needs no dependent LDS chain: N
// compare/select steps on the bit-reversed window and one read of the sorted symbol list.
template <int N>
struct LongMeta { u32 first[N], count[N], offset[N]; };
template <int N>
AHIP_DEVINL void load_long_meta(LongMeta<N> &m, const CodeDesc &cd, int root) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int Lk = root + 1 + k;
    m.first[k] = Lk < 16 ? uniform(cd.first[Lk < 16 ? Lk : 15]) : 0u;
    m.count[k] = Lk < 16 ? uniform(cd.count[Lk < 16 ? Lk : 15]) : 0u;
    m.offset[k] = Lk < 16 ? uniform(cd.offset[Lk < 16 ? Lk : 15]) : 0u;
  }
}
template <bool IS_DIST, int N>
AHIP_DEVINL u32 long_resolve(const LongMeta<N> &m, const u32 *sorted, u32 bits, int root) {
  const u32 rev = __brev(bits);
  u32 pos = 0, len = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const u32 Lk = root + 1 + k;
    const u32 idx = (rev >> (32 - Lk)) - m.first[k];
    const bool hit = idx < m.count[k];  // prefix-free: at most one length hits
    pos = hit ? m.offset[k] + idx : pos;
    len = hit ? Lk : len;
  }
  const u32 e = sorted[pos];
  const u32 hole = IS_DIST ? dist_entry(0, 0) : (u32)E_HOLE;  // unfilled entry: symbol 0, length 0
  return len ? e : hole;
}
constexpr int LL_LONG_N = 15 - LL_ROOT;
constexpr int D_LONG_N = 15 - D_ROOT;
struct BlockMeta {
  LongMeta<LL_LONG_N> ll;
  LongMeta<D_LONG_N> d;
};

// One token at the lane's cursor, straight-line: every lane runs the litlen AND the distance
// half (a 64-lane step almost always contains a match anyway); selects pick the result.  The
// only branches skip the long-code resolution when no lane needs it.
AHIP_DEVINL u32 decode_token(LaneBits &d, const WaveLds &L, const BlockMeta &M, const u32 *inbuf) {
  lb_normalize(d, inbuf);
  const u32 w = lb_peek32(d);  // 32 valid bits; litlen code + extra <= 20
  u32 e = L.ll[w & ((1u << LL_ROOT) - 1)];
  if (__any(e & E_LONG)) {
    asm volatile("; long litlen code" ::: "memory");  // keep this a real branch: if-converted, its LDS read would sit on every step's critical path
    const u32 e2 = long_resolve<false>(M.ll, L.ll_sorted, w, LL_ROOT);
    e = (e & E_LONG) ? e2 : e;
  }
  const u32 cl = e & 15;
  const u32 xb = (e >> 4) & 15;
  const u32 lenv = (e >> 16) + ((w >> cl) & ((1u << xb) - 1));
  const bool is_lit = e & E_LIT;
  const bool is_special = e & (E_EOB | E_BAD | E_HOLE);
  const bool is_match = !is_lit && !is_special;
  d.sh += cl + (is_match ? xb : 0u);
  lb_normalize(d, inbuf);
  const u32 w2 = lb_peek32(d);  // distance code + extra <= 28
  u32 t = L.dt[w2 & ((1u << D_ROOT) - 1)];
  if (__any(is_match && (t & E_LONG))) {
    asm volatile("; long distance code" ::: "memory");
    const u32 t2 = long_resolve<true>(M.d, L.d_sorted, w2, D_ROOT);
    t = (t & E_LONG) ? t2 : t;
  }
  const u32 dl = t & 15;
  const u32 dxb = (t >> 4) & 15;
  const u32 dist = (t >> 16) + ((w2 >> dl) & ((1u << dxb) - 1));
  d.sh += is_match ? dl + dxb : 0u;
  u32 tok = (lenv << 16) | dist;
  tok = (is_match && (t & E_BAD)) ? TK_ERR : tok;
  tok = is_special ? ((e & E_EOB) ? TK_EOB : TK_ERR) : tok;
  tok = is_lit ? (TK_LIT | (e >> 16)) : tok;
  return tok;
}

constexpr u32 LR_EOB = 1, LR_ERR = 2, LR_OVF = 4;
struct LaneRun { u32 end, flags, ntok, nbytes; i32 need; };  // need = max over matches of (dist - bytes before it in this lane)

// Decode from `start` until the cursor reaches `boundary` (or EOB / error).
//  RECORD: token j of this lane goes to slab[j * 64 + lane] -- every active lane is at the same
//          j, so each step is one coalesced 256-byte row store into the L2-resident slab.
template <bool RECORD>
AHIP_DEVINL LaneRun run_lane(bool active, bool rec, u32 start, u32 boundary, const WaveLds &L, const BlockMeta &M,
                             const u32 *inbuf, u32 *slab, int lane) {
  LaneRun r{start, 0, 0, 0, 0};
  LaneBits d{0, 0, 0, 0, 2};
  if (active) lb_init(d, inbuf, start);
  u32 lguard = 0;
  for (;;) {
    bool go = active && r.end < boundary && r.flags == 0;
    if (!__any(go)) break;
    if (++lguard > 2048) { r.flags = LR_ERR; break; }
    if (go) {
      u32 t = decode_token(d, L, M, inbuf);
      if (t & (TK_EOB | TK_ERR)) {
        r.flags = (t & TK_EOB) ? LR_EOB : LR_ERR;
        r.end = lb_pos(d);
      } else if (RECORD && r.ntok >= (u32)SLAB_ROWS) {
        r.flags = LR_OVF;
      } else {
        if (RECORD && rec) slab[r.ntok * 64 + lane] = t;
        const bool lit = t >> 31;
        const i32 req = lit ? 0 : (i32)(t & 0xffff) - (i32)r.nbytes;
        r.need = req > r.need ? req : r.need;
        r.ntok += 1;
        r.nbytes += lit ? 1u : (t >> 16);
        r.end = lb_pos(d);
      }
    }
  }
  return r;
}

// Execute `ntok` queued tokens (nbytes of output) in stream order into the LDS output window.
//  hist: global address of the window's first output byte (earlier output lies below it).
//
// Byte-per-lane formulation -- no per-token loops, no divergence on match length:
//   pass 1  wave scans turn every token into a key {start offset, literal flag, byte | dist-1}
//           (the literal/match boundary scan); keys replace the tokens in the queue.
//   pass 2  64 output bytes at a time.  The (at most 64) tokens that START inside the group drop
//           their key into a 64-entry slot array at their start offset; a wave prefix-max then
//           hands every byte lane the key of the token that covers it.  A lane's byte is the
//           literal, an earlier byte of the window (LDS), a byte of flushed history (HBM/L2), or
//           the byte of a lower lane of the same group (resolved by pointer doubling over the
//           LDS crossbar; only runs and very short distances get there).
//           History loads of group g+1 are issued before group g is finished, so their latency
//           hides behind LDS work.
struct GroupFront {
  u32 val;    // literal byte, or history byte (once the load lands)
  i32 si;     // source offset inside the window (negative: flushed history)
  bool act, lit;
  bool pre;   // byte of a far match already deposited into the window by pass 1
};
template <typename E>
AHIP_DEVINL GroupFront resolve_front(ParLdsT<E> &P, u32 ntok, u32 nbytes, const E *hist, u64 opos, u32 g0, u32 &tcur,
                                     u32 &carry, int lane) {
  u32 *slot = P.slot + ((g0 >> 6) & 1) * 64;  // two slot arrays alternate: no write-after-read stall
  wave_sync();
  slot[lane] = 0;
  const u32 kidx = tcur + lane;
  const u32 k = kidx < ntok ? P.tok[kidx] : 0u;
  const u32 offk = (k >> 17) & 0x1fff;
  const bool ing = k != 0 && offk < g0 + 64;  // tokens are sorted by offset, so these form lanes 0..cnt-1
  (ing ? slot + (offk - g0) : P.slot + 128 + lane)[0] = k;  // lanes without a start write to the dump row (branch-free)
  const u32 cnt = (u32)__popcll(__ballot(ing));
  wave_sync();  // other lanes wrote slot[]: without this hipcc forwards this lane's own 0
  u32 key = slot[lane];
  key = wave_incl_umax(key);
  key = key > carry ? key : carry;  // the token that covers the start of the group
  const u32 last = lane_bcast(k, (int)((cnt - 1) & 63));
  carry = cnt ? last : carry;
  tcur += cnt;
  GroupFront f;
  const u32 x = g0 + lane;
  f.act = x < nbytes;
  f.lit = (key >> 16) & 1;
  f.pre = !f.lit && ((key >> 15) & 1);
  f.val = key & 0xff;
  f.si = f.pre ? (i32)x : (i32)x - (i32)((key & 0x7fff) + 1);
#ifndef AHIP_ABLATE_FAR
  if (f.act && !f.lit && f.si < 0) f.val = hist_get(hist, f.si, opos);  // far sources pass 1 did not take (long, or too close to the window)
#endif
  return f;
}
template <typename E>
AHIP_DEVINL void resolve_back(ParLdsT<E> &P, const GroupFront &f, u32 g0, E *ob, int lane) {
  u32 val = f.val;
  const bool copy = f.act && !f.lit;
  wave_sync();  // bytes of earlier groups were stored by other lanes
  if (copy && f.si >= 0 && (f.si < (i32)g0 || f.pre)) val = ob[f.si];  // pre: its own deposited byte (lower lanes may copy from it)
  const bool dep = copy && !f.pre && f.si >= (i32)g0;
  if (__any(dep)) {
    u32 srcl = dep ? (u32)(f.si - (i32)g0) : (u32)lane;
    bool res = !dep;
    int guard = 0;  // a chain of lower-lane pointers halves every step: 6 steps always suffice
    do {
      const u32 sv = lane_gather(val, srcl);
      const u32 sr = lane_gather(res ? 1u : 0u, srcl);
      const u32 ss = lane_gather(srcl, srcl);
      if (!res) {
        if (sr) { val = sv; res = true; }
        else srcl = ss;
      }
    } while (__any(!res) && ++guard < 8);
  }
  if (f.act && !f.pre) ob[g0 + lane] = (E)val;
}
// pass 2 of the resolver: keys are already in the queue (resolve_member builds them)
template <typename E>
AHIP_DEVINL void resolve_bytes(ParLdsT<E> &P, u32 ntok, u32 nbytes, const E *hist, u64 opos, u32 A, int lane) {
  E *ob = P.obuf + A;
  u32 tcur = 0, carry = 0;
  if (nbytes == 0 || nbytes > (u32)OB_CAP || ntok > (u32)TOK_CAP) return;  // never spin on corrupt bookkeeping
  GroupFront cur = resolve_front(P, ntok, nbytes, hist, opos, 0, tcur, carry, lane);
  u32 g0 = 0;
  for (; g0 + 64 < nbytes; g0 += 64) {  // front of the next group and back of this one: one straight-line body
    GroupFront nxt = resolve_front(P, ntok, nbytes, hist, opos, g0 + 64, tcur, carry, lane);
    resolve_back(P, cur, g0, ob, lane);
    cur = nxt;
  }
  resolve_back(P, cur, g0, ob, lane);
}

// Flush the assembled window to HBM: byte head up to 16-byte alignment, 16-byte body, byte tail.
// obuf index A + i holds output byte i, with A = (address of byte 0) & 15, so LDS and global
// addresses are congruent mod 16.
AHIP_DEVINL void flush_window(const ParLdsT<u16> &P, u16 *g, u32 A, u32 n, int lane) {  // symbols: plain copy
  for (u32 i = lane; i < n; i += 64) g[i] = P.obuf[A + i];
}
AHIP_DEVINL void flush_window(const ParLds &P, u8 *g, u32 A, u32 nbytes, int lane) {
  u32 head = (16 - A) & 15;
  if (head > nbytes) head = nbytes;
  if ((u32)lane < head) g[lane] = P.obuf[A + lane];
  u32 body = (nbytes - head) & ~15u;
  const uint4 *src = (const uint4 *)(P.obuf + A + head);
  uint4 *dst = (uint4 *)(g + head);
  for (u32 i = lane; i < body / 16; i += 64) dst[i] = src[i];
  u32 tail0 = head + body;
  if (tail0 + lane < nbytes) g[tail0 + lane] = P.obuf[A + tail0 + lane];
}

// ------------------------------------------------------------------------------------------
// Tokenizer side
// ------------------------------------------------------------------------------------------

// Serial decode of one Huffman block that EMITS tokens instead of writing bytes: the checked path
// for everything irregular (same decisions, in the same order, as huffman_token<WRITE, CAREFUL>).
template <bool CAREFUL>
AHIP_DEVINL u32 huffman_token_emit(WaveLds &L, BitCursor &b, OutCursor &o, TokSink &sink, u32 ll_max, u32 d_max, int lane) {
  if (CAREFUL && b.pos + ll_max > b.total_bits) return 100 + MS_FALSE_EOS;
  u64 w = peek_bits(b);
  u32 e = uniform(L.ll[(u32)w & ((1u << LL_ROOT) - 1)]);
  if (e & E_LONG) e = uniform(long_lookup<false>(L.lld, L.ll_sorted, (u32)w, LL_ROOT));
  u32 cl = e & 15;
  if (e & (E_LIT | E_EOB | E_BAD | E_HOLE)) {
    if (e & E_LIT) {
      if (o.pos >= o.limit) return 100 + MS_CAP;
      if (sink.base && lane == 0) sink.base[sink.w] = TK_LIT | (e >> 16);
      sink.w += 1;
      o.pos += 1;
      b.pos += cl;
      return 0;
    }
    if (e & E_EOB) { b.pos += cl; return 1; }
    if (e & E_BAD) return 100 + MS_FALSE;
    return 100 + MS_HANG;
  }
  u32 used = cl;
  w >>= cl;
  u32 xb = (e >> 4) & 15;
  i32 len = (i32)(e >> 16);
  if (CAREFUL && xb && b.pos + used + xb > b.total_bits) len -= 1;
  else { len += (i32)((u32)w & ((1u << xb) - 1)); w >>= xb; used += xb; }
  if (CAREFUL && b.pos + used + d_max > b.total_bits) return 100 + MS_FALSE_EOS;
  u32 d = uniform(L.dt[(u32)w & ((1u << D_ROOT) - 1)]);
  if (d & E_LONG) d = uniform(long_lookup<true>(L.dd, L.d_sorted, (u32)w, D_ROOT));
  if (d & E_BAD) return 100 + MS_FALSE;
  u32 dl = d & 15;
  w >>= dl;
  used += dl;
  u32 dxb = (d >> 4) & 15;
  i32 dist = (i32)(d >> 16);
  if (CAREFUL && dxb && b.pos + used + dxb > b.total_bits) dist -= 1;
  else { dist += (i32)((u32)w & ((1u << dxb) - 1)); used += dxb; }
  b.pos += used;
  if ((u64)dist > o.pos) return 100 + MS_FARREF;
  if (o.pos + (u64)len > o.limit) return 100 + MS_CAP;
  if (sink.base && lane == 0) sink.base[sink.w] = ((u32)len << 16) | (u32)dist;
  sink.w += 1;
  o.pos += (u64)len;
  return 0;
}
AHIP_DEVINL u32 huffman_block_emit(WaveLds &L, BitCursor &b, OutCursor &o, TokSink &sink, int lane) {
  const u32 ll_max = L.lld.maxlen, d_max = L.dd.maxlen;
  for (;;) {
    u32 r;
    if ((b.pos >> 3) + 16 <= b.in_len) r = huffman_token_emit<false>(L, b, o, sink, ll_max, d_max, lane);
    else r = huffman_token_emit<true>(L, b, o, sink, ll_max, d_max, lane);
    if (r == 0) continue;
    if (r == 1) return MS_OK;
    return r - 100;
  }
}
// _parseUncompressedBlock as tokens: runs of >= 3 bytes become one TK_STORED record (3 words)
AHIP_DEVINL u32 stored_block_emit(BitCursor &b, OutCursor &o, TokSink &sink, int lane) {
  b.pos = (b.pos + 7) & ~7ull;
  int len = read_bits(b, 16);
  int nlen_raw = read_bits(b, 16);
  int nlen = nlen_raw ^ 0xffff;
  if (len != 0 && len != nlen) return (len < 0 || nlen_raw < 0) ? MS_FALSE_EOS : MS_FALSE;
  u64 byte = b.pos >> 3;
  if ((u64)len > b.in_len - byte) return MS_FALSE;
  if (o.pos + (u64)len > o.limit) return MS_CAP;
  if (len >= 3) {
    if (sink.base && lane == 0) {
      sink.base[sink.w] = TK_STORED | (u32)len;
      sink.base[sink.w + 1] = (u32)byte;
      sink.base[sink.w + 2] = (u32)(byte >> 32);
    }
    sink.w += 3;
  } else {
    for (int i = 0; i < len; ++i) {
      if (sink.base && lane == 0) sink.base[sink.w] = TK_LIT | b.in[byte + i];
      sink.w += 1;
    }
  }
  o.pos += (u64)len;
  b.pos += 8ull * (u64)len;
  return MS_OK;
}

// Decode one Huffman block (tables already built in L) starting at b.pos into tokens.
// Returns MS_* exactly like huffman_block().
AHIP_DEVINL u32 huffman_block_tokenize(WaveLds &L, TokLds &P, u32 *slab, BitCursor &b, OutCursor &o, TokSink &sink,
                                       int lane, ParStats &st) {
  BlockMeta M;
  load_long_meta(M.ll, L.lld, LL_ROOT);
  load_long_meta(M.d, L.dd, D_ROOT);
  const bool emit = sink.base != nullptr;
  u32 wguard = 0;
  for (;;) {
    if (++wguard > (1u << 20)) { st.dbg |= 1; break; }
    const u64 gbyte = (b.pos >> 3) & ~3ull;
    if (gbyte + (u64)IN_DWORDS * 4 > b.in_len) break;  // too close to the end: checked serial path
    // ---- stage the window ----
    AHIP_TICK(t_a);
    {
      const u8 *g = b.in + gbyte;
      for (int k = lane * 4; k < IN_DWORDS; k += 256) {
        const uint4 v = load_u128_unaligned(g + 4 * k);  // IN_DWORDS is a multiple of 4
        *(uint4 *)(P.inbuf + k) = v;
      }
    }
    wave_sync();
    AHIP_TICK(t_b);
    AHIP_ACC(st.cyc[1], t_a, t_b);
    st.windows++;
    const u32 s0 = (u32)(b.pos - gbyte * 8);
    const u32 boundary = (u32)(lane + 1) * SUB_BITS;
    // ---- pass A: lane 0 from the true boundary (recording), the others blind.  The SAME loop body as
    //      the pass-B rounds on purpose: a separate "ends only" variant was measured 8 % slower
    //      (code size / instruction cache) ----
    LaneRun R = run_lane<true>(true, emit && lane == 0, lane == 0 ? s0 : (u32)lane * SUB_BITS, boundary, L, M, P.inbuf,
                               slab, lane);
    AHIP_TICK(t_c);
    AHIP_ACC(st.cyc[2], t_b, t_c);
    // ---- pass B rounds: restart from the predecessor's end until the chain is consistent ----
    // A lane decodes again only when the start it was given last time is no longer its predecessor's end, so
    // after the first round (every lane once, recording) a round usually keeps one or two lanes busy and the
    // lock-step loop is as long as THEIR token count, not the longest of 64.  At the fixpoint every lane
    // started where its predecessor ended and lane 0 started at the true boundary: the chain is the true path
    // up to the first flagged lane.
    u32 used_start = lane == 0 ? s0 : ~0u;
    u32 rguard = 0;
    for (;;) {
      if (++rguard > 80) { st.dbg |= 2; break; }
      // DPP reads need the SOURCE lane active: take lane-1's values with every lane enabled
      const u32 prev_end = lane_prev(R.end), prev_flags = lane_prev(R.flags);
      const bool act = lane > 0 && prev_flags == 0 && prev_end != used_start;
      if (!__any(act)) break;
      LaneRun R2 = run_lane<true>(act, emit, prev_end, boundary, L, M, P.inbuf, slab, lane);
      if (act) { R = R2; used_start = prev_end; }
      st.rounds++;
    }
    const int final_upto = 63;
    AHIP_TICK(t_d);
    AHIP_ACC(st.cyc[3], t_c, t_d);
    // ---- who is on the true path ----
    u64 final_mask = (final_upto >= 63) ? ~0ull : ((2ull << final_upto) - 1);
    u64 flagged = __ballot(R.flags != 0) & final_mask;
    int kstop = flagged ? (__ffsll((long long)flagged) - 1) : 64;
    u32 stop_flags = flagged ? lane_bcast(R.flags, kstop) : 0u;
    if (stop_flags & (LR_ERR | LR_OVF)) { st.fallbacks++; break; }  // the serial decoder decides
    const int nlanes = kstop < 64 ? kstop + 1 : 64;  // kstop's tokens before its EOB count too
    const bool valid = lane < nlanes;
    u32 tot_tok, tot_bytes;
    const u32 T = wave_excl_sum(valid ? R.ntok : 0u, tot_tok);
    const u32 B = wave_excl_sum(valid ? R.nbytes : 0u, tot_bytes);
    if ((u64)tot_bytes > o.limit - o.pos) { st.fallbacks++; break; }  // output window exhausted: serial path reports it
    // a back-reference reaching before the member start: dist > bytes before the token
    if (__any(valid && (i64)R.need > (i64)(o.pos + B))) { st.fallbacks++; break; }
    const u32 next_pos = lane_bcast(R.end, kstop < 64 ? kstop : 63);  // past the EOB code, or lane 63's end
    // ---- slab (lane-major rows) -> the member's token stream (stream order) ----
    AHIP_TICK(t_e0);
    if (emit) {
      // Transposed through LDS (the idle bitstream window) so that the stream is written with
      // coalesced stores: per-lane 4-byte stores to 64 different lines ran at one line per cycle in
      // the address coalescer and cost more than either decode pass.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      constexpr u32 QCAP = 1024;
      static_assert(IN_DWORDS >= (int)QCAP, "token transpose buffer lives in the window buffer");
      const u32 cnt = valid ? R.ntok : 0u;
      u32 *q = sink.base + sink.w;
      for (u32 lo = 0; lo < tot_tok; lo += QCAP) {
        // rows of this lane whose stream index T + r falls into [lo, lo + QCAP)
        const u32 rb = lo > T ? lo - T : 0u;
        const u32 re = lo + QCAP > T ? (cnt < lo + QCAP - T ? cnt : lo + QCAP - T) : 0u;
        const bool any_rows = rb < re;
        const u32 rmin = ~wave_umax(any_rows ? ~rb : 0u), rmax = wave_umax(any_rows ? re : 0u);
        const u32 tb = T - lo;  // wraps for lanes that start before the batch; only used when in range
        for (u32 r = rmin; r < rmax; r += 16) {  // 16 row loads in flight: this loop is latency-bound
          u32 t[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) t[u] = (r + u >= rb && r + u < re) ? slab[(r + u) * 64 + lane] : 0u;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (r + u >= rb && r + u < re) P.inbuf[tb + r + u] = t[u];
        }
        wave_sync();
        const u32 nq = tot_tok - lo < QCAP ? tot_tok - lo : QCAP;
        for (u32 k = lane; k < nq; k += 64) q[lo + k] = P.inbuf[k];
        wave_sync();
      }
    }
    AHIP_TICK(t_f);
    AHIP_ACC(st.cyc[4], t_e0, t_f);
    sink.w += tot_tok;
    o.pos += tot_bytes;
    b.pos = gbyte * 8 + next_pos;
    if (kstop < 64) return MS_OK;
  }
  AHIP_TICK(t_s0);
  u32 rs = huffman_block_emit(L, b, o, sink, lane);
  AHIP_TICK(t_s1);
  AHIP_ACC(st.cyc[7], t_s0, t_s1);
  return rs;
}

// Inflate one stream.  Mirrors Inflate._inflate(): loop blocks until BFINAL, an error, or EOS.
//  PAR = false: the serial byte-writing decoder (A/B baseline, single kernel).
//  PAR = true : tokenizer -- no bytes are written; tokens go to `tokens` (nullptr = sizing run).
//  CHUNK = true: one chunk of a long stream (ChunkCtx): bit-granular start, `hist` bytes of earlier output count
//  as already produced (so the back-reference range check holds across the chunk boundary), and the loop stops
//  in front of a block header that sits on a candidate position; end_pos is then reported in BITS.
template <bool WRITE, bool PAR, bool CHUNK = false>
AHIP_DEVINL void inflate_member(WaveLds &L, HeaderLds &H, TokLds *P, u32 *slab, const u8 *in, u64 in_len,
                                const MemberDesc &m, u8 *out, u32 *tokens, MemberResult &res, int lane,
                                const ChunkCtx *cx = nullptr) {
  ParStats st{};
  BitCursor b{in, in_len, in_len * 8, m.in_off * 8 + (CHUNK ? cx->start_bit : 0u), nullptr, 0, 0};
  const u64 hist = CHUNK ? cx->hist : 0u;
  OutCursor o{out + m.out_off, hist, m.out_limit + hist};
  TokSink sink{tokens, 0};
  u32 status = MS_EOS, blocks = 0;
  for (;;) {
    if (((b.pos + 7) >> 3) >= in_len) { status = MS_EOS; break; }
    if (CHUNK && blocks) {  // a later chunk takes over at this block?
      u32 lo = 0, hi = cx->n_cand;
      while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (cx->cand_bits[mid] < b.pos) lo = mid + 1; else hi = mid; }
      if (lo < cx->n_cand && cx->cand_bits[lo] == b.pos) { status = MS_CHUNK_END; break; }
    }
    int hdr = read_bits(b, 3);
    ++blocks;
    const bool final_block = hdr & 1;
    const int btype = hdr >> 1;
    u32 r;
    if (btype == 0) {
      r = PAR ? stored_block_emit(b, o, sink, lane) : stored_block<WRITE>(b, o, lane);
    } else if (btype == 3) {
      r = MS_FALSE;
    } else {
      int hlit = 288, hdist = 30;
      r = MS_OK;
      AHIP_TICK(t_h0);
      if (btype == 1) fixed_lengths(H.lens, lane);
      else {
        if (PAR) {  // stage the (at most 569-byte) dynamic header into the idle window buffer
          const u64 sb = (b.pos >> 3) & ~3ull;
          constexpr u32 HDR_STAGE = 640;
          if (sb + HDR_STAGE <= in_len) {
            for (u32 k = lane * 4; k < HDR_STAGE / 4; k += 256)
              *(uint4 *)(P->inbuf + k) = load_u128_unaligned(in + sb + 4 * k);
            wave_sync();
            b.stage = P->inbuf; b.stage_byte = sb; b.stage_len = HDR_STAGE;
          }
        }
        r = dynamic_header(H, b, lane, hlit, hdist);
        b.stage = nullptr;
      }
      if (r == MS_OK) {
        bool ok = build_decode_table<false>(H.lens, hlit, L.ll, LL_ROOT, L.lld, L.ll_sorted, lane);
        ok &= build_decode_table<true>(H.lens + hlit, hdist, L.dt, D_ROOT, L.dd, L.d_sorted, lane);
        AHIP_TICK(t_h1);
        AHIP_ACC(st.cyc[0], t_h0, t_h1);
        if (!ok) r = MS_OVERSUB;
        else if (PAR) r = huffman_block_tokenize(L, *P, slab, b, o, sink, lane, st);
        else r = huffman_block<WRITE>(L, b, o, lane);
      }
    }
    if (r != MS_OK) { status = r; break; }
    if (final_block) { status = MS_OK; break; }
  }
  if (lane == 0) {
    // Position the reference's InputStream is left at.  Exact after a complete block
    // (whole bytes are un-read) and after an end-of-input failure; after a bad-symbol failure
    // in the middle of the input the reference has over-read by up to two bytes -- see
    // DESIGN.md "deviations".
    u64 end = (b.pos + 7) >> 3;
    if (status == MS_FALSE_EOS) { end = in_len; status = MS_FALSE; }  // every byte was pulled into the accumulator
    res.end_pos = end > in_len ? in_len : end;
    if (CHUNK && status == MS_CHUNK_END) res.end_pos = b.pos;
    res.out_len = o.pos - hist;
    res.status = status;
    res.blocks = blocks;
    res.windows = st.windows;
    res.rounds = st.rounds;
    res.fallbacks = st.fallbacks;
    res.partial = st.partial;
    for (int k = 0; k < 8; ++k) res.cyc[k] = st.cyc[k];
    if (st.dbg) res.cyc[7] = 0xdead0000u | st.dbg;
    res.tok_words = sink.w;
  }
}

// ------------------------------------------------------------------------------------------
// Resolver side: replay one member's token stream into its output window
// ------------------------------------------------------------------------------------------
template <typename E>
AHIP_DEVINL void resolve_member(ParLdsT<E> &P, const u8 *in, const u32 *tokens, u64 nwords, E *out_base, u32 *cyc, int lane) {
  constexpr bool MARK = sizeof(E) == 2;
  u64 cur = 0, opos = 0;
  if (lane == 0) P.misc[0] = 0xffffffffu;
  wave_sync();
  while (cur < nwords) {
    AHIP_TICK(t_0);
    // ---- fetch up to TOK_CAP words; a stored-run record ends the batch (it is handled at a batch head) ----
    const u32 want = (nwords - cur) < (u64)TOK_CAP ? (u32)(nwords - cur) : (u32)TOK_CAP;
    u32 first_stored = want;
    for (u32 k = lane; k < want; k += 64) {
      const u32 t = tokens[cur + k];
      P.tok[k] = t;
      if ((t & 0xe0000000u) == TK_STORED) atomicMin(&P.misc[0], k);  // lowest index wins
    }
    wave_sync();
    first_stored = uniform(P.misc[0]) < want ? uniform(P.misc[0]) : want;
    // NOTE: the two words after a TK_STORED header are raw offsets and may alias the pattern; only the
    // FIRST hit is trusted, and the scan restarts after it.
    if (first_stored == 0) {
      // stored run at the head: input -> output copy
      const u32 len = P.tok[0] & 0xffff;
      const u64 src = (u64)P.tok[1] | ((u64)P.tok[2] << 32);
      for (u32 i = lane; i < len; i += 64) out_base[opos + i] = (E)in[src + i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      opos += len;
      cur += 3;
      if (lane == 0) P.misc[0] = 0xffffffffu;
      wave_sync();
      continue;
    }
    u32 ntok = first_stored;
    wave_sync();
    if (lane == 0) P.misc[0] = 0xffffffffu;
    // ---- pass 1 (keys) with the byte cut at OB_CAP; matches whose whole source is flushed history (and
    //      at most 16 bytes long) are fetched per TOKEN with two 8-byte loads and deposited into the window
    //      right here -- one vector-memory instruction per 64 tokens instead of one byte gather per 64 bytes ----
    E *g = out_base + opos;
    const u32 A = MARK ? 0u : (u32)((uintptr_t)g & 15);
    E *obw = P.obuf + A;
    // software-pipelined by one chunk of 64 tokens: the history loads of chunk c+1 are in flight while
    // chunk c is deposited
    struct Chunk { u32 idx, len, off, key, total, nf, nin; bool fits, pre, cut; u64 w0, w1; };
    auto prep = [&](u32 c, u32 run) -> Chunk {
      Chunk q;
      q.idx = c + lane;
      const bool inb = q.idx < ntok;
      const u32 t = inb ? P.tok[q.idx] : 0u;
      const bool lit = t >> 31;
      q.len = inb ? (lit ? 1u : (t >> 16)) : 0u;
      q.off = run + wave_excl_sum(q.len, q.total);
      q.fits = inb && q.off + q.len <= (u32)OB_CAP;
      q.key = (1u << 30) | (q.off << 17) | (lit ? (0x10000u | (t & 0xff)) : ((t & 0xffff) - 1));
      const i32 srel = (i32)q.off - (i32)(t & 0xffff);  // source start relative to the window
      q.pre = !MARK && q.fits && !lit && q.len <= 16 && srel + 16 <= 0;  // (symbols take the per-element path)
      q.w0 = q.w1 = 0;
      if (q.pre) { const u8 *sp = (const u8 *)g + srel; q.w0 = load_u64_unaligned(sp); q.w1 = load_u64_unaligned(sp + 8); }
      const u64 fm = __ballot(q.fits), im = __ballot(inb);
      q.nf = (u32)__popcll(fm);
      q.nin = (u32)__popcll(im);
      q.cut = fm != im;  // the window is full: cut after the last fitting token
      return q;
    };
    u32 run = 0, kept = 0, c = 0;
    Chunk ck = prep(0, 0);
    for (;;) {
      const bool more = !ck.cut && c + 64 < ntok;
      Chunk nxt = ck;
      if (more) nxt = prep(c + 64, run + ck.total);
      if (ck.pre) {
        E *dp = obw + ck.off;
#pragma unroll
        for (u32 k = 0; k < 16; ++k)
          if (k < ck.len) dp[k] = (E)(u8)((k < 8 ? ck.w0 : ck.w1) >> (8 * (k & 7)));
      }
      if (ck.fits) P.tok[ck.idx] = ck.key | (ck.pre ? 0x8000u : 0u);
      if (ck.cut) {
        kept = c + ck.nf;
        run = ck.nf ? lane_bcast(ck.off + ck.len, (int)ck.nf - 1) : run;
        break;
      }
      run += ck.total;
      kept = c + ck.nin;
      if (!more) break;
      ck = nxt;
      c += 64;
    }
    ntok = kept;
    wave_sync();
    AHIP_TICK(t_1);
    AHIP_ACC(cyc[6], t_0, t_1);
    const u32 nbytes = run;
    resolve_bytes(P, ntok, nbytes, (const E *)g, opos, A, lane);
    wave_sync();
    AHIP_TICK(t_2);
    AHIP_ACC(cyc[5], t_1, t_2);
    flush_window(P, g, A, nbytes, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // later batches read this output back
    AHIP_TICK(t_3);
    AHIP_ACC(cyc[5], t_2, t_3);
    opos += nbytes;
    cur += ntok;
  }
}

}  // namespace ahip
