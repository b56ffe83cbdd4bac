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

struct ParLds {
  u32 inbuf[IN_DWORDS] __attribute__((aligned(16)));
  u32 tok[TOK_CAP];
  u8 obuf[OB_CAP + 32] __attribute__((aligned(16)));
  u32 slot[3 * 64];  // two alternating 64-entry start-slot rows + one dump row
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
AHIP_DEVINL u32 long_resolve(const LongMeta<N> &m, const u16 *sorted, u32 bits, int root) {
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
  const u32 sym = sorted[pos];
  const u32 e = IS_DIST ? dist_entry(sym, len) : litlen_entry(sym, len);
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
struct LaneRun { u32 end, flags, ntok, nbytes; };

// Decode from `start` until the cursor reaches `boundary` (or EOB / error).
//  RECORD: token j of this lane goes to slab[j * 64 + lane] -- every active lane is at the same
//          j, so each step is one coalesced 256-byte row store into the L2-resident slab.
template <bool RECORD>
AHIP_DEVINL LaneRun run_lane(bool active, bool rec, u32 start, u32 boundary, const WaveLds &L, const BlockMeta &M,
                             const u32 *inbuf, u32 *slab, int lane) {
  LaneRun r{start, 0, 0, 0};
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
        r.ntok += 1;
        r.nbytes += (t >> 31) ? 1u : (t >> 16);
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
};
AHIP_DEVINL GroupFront resolve_front(ParLds &P, u32 ntok, u32 nbytes, const u8 *hist, u32 g0, u32 &tcur, u32 &carry,
                                     int lane) {
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
  f.val = key & 0xff;
  f.si = (i32)x - (i32)((key & 0x7fff) + 1);
#ifndef AHIP_ABLATE_FAR
  if (f.act && !f.lit && f.si < 0) f.val = hist[f.si];
#endif
  return f;
}
AHIP_DEVINL void resolve_back(ParLds &P, const GroupFront &f, u32 g0, u8 *ob, int lane) {
  u32 val = f.val;
  const bool copy = f.act && !f.lit;
  wave_sync();  // bytes of earlier groups were stored by other lanes
  if (copy && f.si >= 0 && f.si < (i32)g0) val = ob[f.si];
  const bool dep = copy && f.si >= (i32)g0;
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
  if (f.act) ob[g0 + lane] = (u8)val;
}
AHIP_DEVINL void resolve_window(ParLds &P, u32 ntok, u32 nbytes, const u8 *hist, u32 A, int lane) {
  // pass 1: tokens -> keys
  u32 run = 0;
  for (u32 c = 0; c < ntok; c += 64) {
    const u32 idx = c + lane;
    const bool in = idx < ntok;
    const u32 t = in ? P.tok[idx] : 0u;
    const bool lit = t >> 31;
    const u32 len = in ? (lit ? 1u : (t >> 16)) : 0u;
    u32 total;
    const u32 off = run + wave_excl_sum(len, total);
    const u32 key = (1u << 30) | (off << 17) | (lit ? (0x10000u | (t & 0xff)) : ((t & 0xffff) - 1));
    if (in) P.tok[idx] = key;
    run += total;
  }
  wave_sync();
  // pass 2: bytes
  u8 *ob = P.obuf + A;
  u32 tcur = 0, carry = 0;
  if (nbytes == 0 || nbytes > (u32)OB_CAP || ntok > (u32)TOK_CAP) return;  // never spin on corrupt bookkeeping
  GroupFront cur = resolve_front(P, ntok, nbytes, hist, 0, tcur, carry, lane);
  u32 g0 = 0;
  for (; g0 + 64 < nbytes; g0 += 64) {  // front of the next group and back of this one: one straight-line body
    GroupFront nxt = resolve_front(P, ntok, nbytes, hist, g0 + 64, tcur, carry, lane);
    resolve_back(P, cur, g0, ob, lane);
    cur = nxt;
  }
  resolve_back(P, cur, g0, ob, lane);
}

// Flush the assembled window to HBM: byte head up to 16-byte alignment, 16-byte body, byte tail.
// obuf index A + i holds output byte i, with A = (address of byte 0) & 15, so LDS and global
// addresses are congruent mod 16.
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

// Move the tokens of lanes [la, lb) from the slab into the LDS queue in stream order, checking
// every distance against the bytes that exist before the token.  Returns false when a
// back-reference reaches before the member start (the serial decoder then decides).
//  T/B: exclusive prefix sums of tokens / bytes over the window's lanes; Ta/Ba: their values at la.
AHIP_DEVINL bool gather_batch(ParLds &P, const u32 *slab, int la, int lb, u32 n, u32 T, u32 B, u32 Ta, u32 Ba,
                              u64 hist0, int lane) {
  const bool mine = lane >= la && lane < lb;
  const u32 cnt = mine ? n : 0u;
  const u32 steps = wave_umax(cnt);
  u32 *q = P.tok + (T - Ta);
  u64 avail = hist0 + (B - Ba);  // bytes of this member that precede the lane's first token
  bool far = false;
  for (u32 k = 0; k < steps; k += 8) {
    u32 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = (k + u < cnt) ? slab[(k + u) * 64 + lane] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k + u < cnt) {
        const bool lit = t[u] >> 31;
        if (!lit && (u64)(t[u] & 0xffff) > avail) far = true;
        avail += lit ? 1u : (t[u] >> 16);
        q[k + u] = t[u];
      }
    }
  }
  return !__any(far);
}

// Decode one Huffman block (tables already built in L) starting at b.pos.
// Returns MS_* exactly like huffman_block().
template <bool WRITE>
AHIP_DEVINL u32 huffman_block_parallel(WaveLds &L, ParLds &P, u32 *slab, BitCursor &b, OutCursor &o, int lane,
                                       ParStats &st) {
  BlockMeta M;
  load_long_meta(M.ll, L.lld, LL_ROOT);
  load_long_meta(M.d, L.dd, D_ROOT);
  u32 wguard = 0;
  for (;;) {
    if (++wguard > 4096) { st.dbg |= 1; break; }
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
    // ---- pass A: lane 0 from the true boundary (recording), the others blind ----
    LaneRun R = run_lane<true>(true, lane == 0, lane == 0 ? s0 : (u32)lane * SUB_BITS, boundary, L, M, P.inbuf, slab, lane);
    AHIP_TICK(t_c);
    AHIP_ACC(st.cyc[2], t_b, t_c);
    // ---- pass B rounds: restart from the predecessor's end until the chain is consistent ----
    int final_upto = 0;
    u32 rguard = 0;
    for (;;) {
      if (++rguard > 80) { st.dbg |= 2; break; }
      u64 flagged = __ballot(R.flags != 0);
      u64 final_mask = (final_upto >= 63) ? ~0ull : ((2ull << final_upto) - 1);
      if (flagged & final_mask) break;  // a final lane ended the block (or hit an error)
      if (final_upto >= 63) break;
      u32 prev_end = lane_prev(R.end);
      u32 prev_flags = lane_prev(R.flags);
      bool act = lane > final_upto && prev_flags == 0;
      LaneRun R2 = run_lane<true>(act, true, prev_end, boundary, L, M, P.inbuf, slab, lane);
      bool mism = act && (R2.end != R.end || R2.flags != R.flags);
      if (act) R = R2;
      u64 mm = __ballot(mism);
      int f = mm ? (__ffsll((long long)mm) - 1) : 63;
      final_upto = f > final_upto ? f : final_upto + 1;
      st.rounds++;
    }
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
    const u32 next_pos = lane_bcast(R.end, kstop < 64 ? kstop : 63);  // past the EOB code, or lane 63's end
    const u32 lane_start = lane == 0 ? s0 : lane_prev(R.end);
    // ---- resolve in batches of whole lanes that fit the LDS token queue / output window ----
    bool bail = false;
    int la = 0;
    u32 bguard = 0;
    while (la < nlanes) {
      if (++bguard > 80) { st.dbg |= 4; bail = true; break; }
      const u32 Ta = lane_bcast(T, la), Ba = lane_bcast(B, la);
      const bool fits = valid && lane >= la && (T + R.ntok - Ta <= (u32)TOK_CAP) && (B + R.nbytes - Ba <= (u32)OB_CAP);
      const u64 fm = __ballot(fits) >> la;            // lanes la, la+1, ... as bits 0, 1, ...
      const int take = fm == ~0ull ? 64 : (__ffsll((long long)~fm) - 1);  // leading run of fitting lanes
      if (take == 0) {  // one lane alone exceeds the queue (very dense or very long tokens)
        b.pos = gbyte * 8 + lane_bcast(lane_start, la);
        bail = true;
        break;
      }
      const int lb = la + take;
      const u32 Tb = lb < 64 ? lane_bcast(T, lb < 64 ? lb : 63) : tot_tok;
      const u32 Bb = lb < 64 ? lane_bcast(B, lb < 64 ? lb : 63) : tot_bytes;
      const u32 ntok = (lb < nlanes ? Tb : tot_tok) - Ta, nbytes = (lb < nlanes ? Bb : tot_bytes) - Ba;
      AHIP_TICK(t_e0);
      if (!gather_batch(P, slab, la, lb, R.ntok, T, B, Ta, Ba, o.pos, lane)) {
        b.pos = gbyte * 8 + lane_bcast(lane_start, la);
        bail = true;
        break;
      }
      wave_sync();
      AHIP_TICK(t_f);
      AHIP_ACC(st.cyc[4], t_e0, t_f);
      if (WRITE && nbytes) {
        u8 *g = o.base + o.pos;
        const u32 A = (u32)((uintptr_t)g & 15);
        resolve_window(P, ntok, nbytes, g, A, lane);
        wave_sync();
        AHIP_TICK(t_g);
        AHIP_ACC(st.cyc[5], t_f, t_g);
        flush_window(P, g, A, nbytes, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // later batches read this output back
        AHIP_TICK(t_h);
        AHIP_ACC(st.cyc[6], t_g, t_h);
      }
      o.pos += nbytes;
      la = lb;
      if (la < nlanes) st.partial++;
    }
    if (bail) { st.fallbacks++; break; }
    b.pos = gbyte * 8 + next_pos;
    if (kstop < 64) return MS_OK;
  }
  AHIP_TICK(t_s0);
  u32 rs = huffman_block<WRITE>(L, b, o, lane);
  AHIP_TICK(t_s1);
  AHIP_ACC(st.cyc[7], t_s0, t_s1);
  return rs;
}

// Inflate one stream.  Mirrors Inflate._inflate(): loop blocks until BFINAL, an error, or EOS.
//  PAR: Huffman blocks go through huffman_block_parallel (P must be valid), else the serial decoder.
template <bool WRITE, bool PAR>
AHIP_DEVINL void inflate_member(WaveLds &L, ParLds *P, u32 *slab, const u8 *in, u64 in_len, const MemberDesc &m,
                                u8 *out, MemberResult &res, int lane) {
  ParStats st{};
  BitCursor b{in, in_len, in_len * 8, m.in_off * 8, nullptr, 0, 0};
  OutCursor o{out + m.out_off, 0, m.out_limit};
  u32 status = MS_EOS, blocks = 0;
  for (;;) {
    if (((b.pos + 7) >> 3) >= in_len) { status = MS_EOS; break; }
    int hdr = read_bits(b, 3);
    ++blocks;
    const bool final_block = hdr & 1;
    const int btype = hdr >> 1;
    u32 r;
    if (btype == 0) {
      r = stored_block<WRITE>(b, o, lane);
    } else if (btype == 3) {
      r = MS_FALSE;
    } else {
      int hlit = 288, hdist = 30;
      r = MS_OK;
      AHIP_TICK(t_h0);
      if (btype == 1) fixed_lengths(L.lens, lane);
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
        r = dynamic_header(L, b, lane, hlit, hdist);
        b.stage = nullptr;
      }
      if (r == MS_OK) {
        bool ok = build_decode_table<false>(L.lens, hlit, L.ll, LL_ROOT, L.lld, L.ll_sorted, lane);
        ok &= build_decode_table<true>(L.lens + hlit, hdist, L.dt, D_ROOT, L.dd, L.d_sorted, lane);
        AHIP_TICK(t_h1);
        AHIP_ACC(st.cyc[0], t_h0, t_h1);
        if (!ok) r = MS_OVERSUB;
        else if (PAR) r = huffman_block_parallel<WRITE>(L, *P, slab, b, o, lane, st);
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
    res.out_len = o.pos;
    res.status = status;
    res.blocks = blocks;
    res.windows = st.windows;
    res.rounds = st.rounds;
    res.fallbacks = st.fallbacks;
    res.partial = st.partial;
    for (int k = 0; k < 8; ++k) res.cyc[k] = st.cyc[k];
    if (st.dbg) res.cyc[7] = 0xdead0000u | st.dbg;
  }
}

}  // namespace ahip
