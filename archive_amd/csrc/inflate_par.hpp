// inflate_par.hpp -- wave-parallel decode of one DEFLATE stream ("member"): tokenizer and resolver.
//
// The reference decodes strictly symbol by symbol (inflate.dart:300-343) and copies every back-reference as it
// meets it (output_memory_stream.dart:79-98).  Here one wave64 works on one member in two stages that run as two
// kernels (archive_hip.hip):
//
//   tokenizer   huffman_block_tokenize(): the block's bitstream is cut into ITEMS of SUB_BITS; the 64 lanes are
//               independent workers in a continuous flow -- SPEC(s) decodes blind over the tail of item s to predict
//               where item s+1 starts (Huffman streams self-synchronise), RUN(s) decodes item s from a start and
//               records its tokens into the lane's own column of the member's token area.  A scoreboard in LDS
//               validates the runs in order, repairs mispredicted items at once and retires final items into the
//               member's RUN DIRECTORY {where the run's tokens are, how many, at which output offset the run starts}.
//               Every token already carries the output offset behind it, relative to its run (the lane counts the
//               bytes of its run anyway), so nobody ever needs a prefix sum over token lengths again.
//   resolver    resolve_member(): 64 TOKENS per step.  A lane takes one token: literals store their byte, matches
//               whose source lies in flushed output are fetched per token (16 / 32 bytes, one chunk ahead) and
//               deposited into an LDS output window with exact-length unaligned stores, matches whose source is
//               still in the window go in rounds (the first pending match is always ready, a later one when its
//               source ends in front of the first pending destination).  The window is flushed to HBM with 16-byte
//               stores.  resolve_member_sym() is the older byte-per-lane formulation, kept for the 16-bit symbols of
//               the chunked single-stream decode (sm_inflate.hpp).
//
// Anything irregular (bad symbol on the true path, back-reference before the start of the output, output window
// exhausted, input too close to its end for unchecked reads) drops to the serial decoder of inflate_wave.hpp at the
// last retired item; that path restates the reference symbol by symbol and emits tokens into the same store, so the
// parallel path only ever commits tokens of well-formed data and malformed data takes exactly the reference's path.
#pragma once
#include "inflate_wave.hpp"

namespace ahip {

#ifndef AHIP_SUB_BITS
#define AHIP_SUB_BITS 512
#endif
#ifndef AHIP_RING_DW
#define AHIP_RING_DW 1536
#endif
#ifndef AHIP_SPEC_BITS
#define AHIP_SPEC_BITS 256
#endif
#ifndef AHIP_STEPS
#define AHIP_STEPS 12
#endif
#ifndef AHIP_EMIT_MIN
#define AHIP_EMIT_MIN 16
#endif
constexpr int SUB_BITS = AHIP_SUB_BITS;          // bits per work item ("subsequence") of the tokenizer
constexpr int SUB_DW = SUB_BITS / 32;
// The staged bitstream is a ring of RING_DW dwords in LDS -- NOT a power of two: the tokenizer's time is inversely
// proportional to the waves resident on a CU (measured: 13.9 / 10.6 / 9.6 ms with 6 / 8 / 9 of them) and LDS is what
// limits them, so the ring is as small as the flow allows (96 items; 64 starve the lanes).  Its first RING_MIRROR dwords
// are kept a second time behind its end: a work unit reads at most that far past its first dword, so the lane readers
// never wrap an index (no mask in the decode step at all) -- only the scheduler does, once per unit.
constexpr int RING_DW = AHIP_RING_DW;
constexpr int ITEMS = RING_DW / SUB_DW;          // items the ring (and the scoreboard) holds
constexpr int RING_MIRROR = SUB_DW + 8;          // dwords 0 .. RING_MIRROR-1 again at RING_DW ..
constexpr u32 SPEC_BITS = AHIP_SPEC_BITS;        // a speculative run covers the last SPEC_BITS of its item
constexpr int STEPS = AHIP_STEPS;                // decode steps between two scheduling points
constexpr u32 EMIT_MIN = AHIP_EMIT_MIN;          // items retired per emit (<= 64)
constexpr u32 REPAIRS = 16;                      // repairs handed out per scheduling point (the one validation waits for comes first)
constexpr u32 EPOCH_ITEMS = (1u << 27) / SUB_BITS;  // positions inside an epoch stay below 2^27 + slack
static_assert(RING_DW % SUB_DW == 0 && RING_DW % 4 == 0 && RING_MIRROR % 4 == 0 && ITEMS >= 64, "ring geometry");
static_assert(SUB_BITS % 128 == 0 && SPEC_BITS <= (u32)SUB_BITS && SUB_BITS + 64 < 4096, "item geometry");
static_assert(EMIT_MIN >= 1 && EMIT_MIN <= 64 && (u32)ITEMS >= 2 * EMIT_MIN && ITEMS >= 32, "scheduler geometry");
// x in [0, 2 N) -> x mod N
AHIP_DEVINL u32 wrap_ring(u32 x) { return x >= (u32)RING_DW ? x - (u32)RING_DW : x; }
AHIP_DEVINL u32 wrap_item(u32 x) { return x >= (u32)ITEMS ? x - (u32)ITEMS : x; }

// ---- the token store: what the tokenizer hands to the resolver ----
// The decode step works on a STEP WORD: literal (0x8000 | byte) << 16 (negative), match len << 16 | dist
// (len <= 258, dist <= 32768); anything else is not negative and has a zero distance field.
constexpr u32 TK_LIT = 0x80000000u;  // | byte << 16
// What is RECORDED per token is  end << 16 | payload:
//   end      the low 16 bits of the output bytes the token's RUN has produced up to and including this token -- the
//            recording lane counts them anyway; a token's length is end - (end of the token before it), its output
//            offset is the run's offset (directory) + the end before it;
//   payload  0x8000 | byte for a literal, dist - 1 (< 0x8000) for a match.
constexpr u32 REC_LIT = 0x8000u;
AHIP_DEVINL u32 rec_word(u32 end_bytes, u32 step_word) {
  const bool lit = (i32)step_word < 0;
  return (end_bytes << 16) | (lit ? (step_word >> 16) : (step_word & 0xffffu) - 1u);
}
// Run directory entry (16 bytes, stream order):
//   x  word offset of the run's first token in the member's token area
//   y  tokens of the run (DF_CNT) | flags
//   z,w  output offset of the run's first byte, relative to the member's own first byte (64 bits)
// DF_BIG: `end` may wrap inside the run (it produced more than 65535 bytes, or the serial decoder wrote it): lengths
//         are still exact modulo 2^16, offsets come from a prefix sum over them.
// DF_STORED: not tokens at all but a stored block of (y & DF_CNT) >= 3 bytes; area[x], area[x + 1] = the absolute
//         input byte offset of its first byte (lo, hi).
typedef uint4 DirEnt;
constexpr u32 DF_BIG = 0x80000000u, DF_STORED = 0x40000000u, DF_CNT = 0x00ffffffu;
constexpr u32 DIR_BYTES = 16;

// LDS of the tokenizer (next to WaveLds): the staged bitstream ring and the per-item scoreboard
struct TokLds {
  u32 inbuf[RING_DW + RING_MIRROR] __attribute__((aligned(16)));
  u32 fa[ITEMS];    // decode run of item s: state<<30 | flags<<28 | lane<<22 | (start - s*SUB)<<12 | (end - s*SUB)
                    //   state 0 nothing yet, 1 a run is in flight, 2 a run has finished,
                    //   3 no run yet but a PREDICTED START: the speculative run over the tail of item s-1 ended (usably) at
                    //     s*SUB + (fa & 0xfff)
  u32 fb[ITEMS];    //   where its first token sits in the recording lane's column (words)
  u32 fc[ITEMS];    //   bytes | tokens<<20
  u16 need[ITEMS];  //   max over its matches of (distance - bytes of the item in front of the match): the "source
                    //   before the start of the output" check
  u16 q[REPAIRS];   // repair queue of one scheduling point: the starts to decode from, counted from the window's first item
};
// (The words used in every column of the member's token area are per-lane data: they live in a register of the owning
//  lane -- `colreg`, handed through the emitters by reference -- and the serial writer reads / writes another lane's with
//  v_readlane / a select.)
AHIP_DEVINL u32 col_get(u32 colreg, u32 c) { return lane_bcast(colreg, (int)c); }  // c wave-uniform
AHIP_DEVINL void col_set(u32 &colreg, u32 c, u32 v, int lane) { colreg = (u32)lane == c ? v : colreg; }
constexpr u32 SYM_MARK = 0x8000;  // 16-bit symbols of the chunked single-stream decode: SYM_MARK + j = "byte j of the 32 KiB in front of this chunk"

// Token store of one member in device memory (tokenizer -> resolver hand-off).
//   area  64 columns of col_cap words.  Lane l of the flow decoder records the tokens of its runs into column l, one
//         behind the other, and they stay there: nothing is transposed or copied.  The serial decoder (irregular
//         blocks, stored blocks) fills whatever the columns have left, column after column.
//   dir   the run directory (DirEnt), in stream order.  The resolver walks it; a lane of the resolver finds its token
//         by (run, index in the run).
// Sizes (tok_layout): 1.5 words of area and 1/16 directory entry per output byte -- a token covers >= 1 byte, a
// stored run of >= 3 bytes takes 2 words, repeated runs waste some; a member that still runs out (MS_TOKFULL) is
// decoded by the byte-writing serial kernel afterwards.
struct TokSink {
  u32 *area;   // nullptr: sizing run, nothing is stored
  u32 col_cap;
  DirEnt *dir;
  u32 dir_cap, ndir;
  u32 scol, spos, srun;  // serial writer: its column (~0u: none yet), next word, first word of the open run
  bool full;
  bool sizing;  // a sizing run that keeps its tokens: a full sink is reported as MR_FAR next to the true status
  u32 sbytes;   // serial writer: bytes the open run has produced (its first byte's output offset is not kept: the
                //   callers know where the output stands, out_rel, and the run starts sbytes in front of that)
};
// member k of a launch group whose output starts out_rel bytes into the group's output
AHIP_DEVINL void tok_layout(u64 out_rel, u64 out_limit, u32 k, u64 &tok_off, u32 &col_cap, u64 &dir_off, u32 &dir_cap) {
  tok_off = (out_rel * 3) / 2 + (u64)k * 1024;
  col_cap = (u32)(((out_limit * 3) / 2 + 1023) / 64);
  dir_off = out_rel / 16 + (u64)k * 64;
  dir_cap = (u32)(out_limit / 16 + 63);
}

// The same areas laid out along the INPUT, for a sizing run that keeps its tokens (output offsets are what it is
// about to find out): candidate k starts at byte `pos` of the stream and owns it up to the next candidate, `span`
// bytes on.  3 words of area per compressed byte (a token takes >= 2 bits, usually >= 8), one directory entry per
// 32 bytes (a run covers one 64-byte item).  A member that does not fit is tokenized again along the output.
#ifndef AHIP_IN_R
#define AHIP_IN_R 3
#endif
#ifndef AHIP_IN_ALIGN
#define AHIP_IN_ALIGN 16
#endif
#ifndef AHIP_IN_PAD
#define AHIP_IN_PAD 1024
#endif
constexpr u64 IN_R = AHIP_IN_R, IN_ALIGN = AHIP_IN_ALIGN, IN_PAD = AHIP_IN_PAD;  // words per input byte, area alignment and per-candidate allowance (words)
AHIP_DEVINL void tok_layout_in(u64 pos, u64 span, u32 k, u64 &tok_off, u32 &col_cap, u64 &dir_off, u32 &dir_cap) {
  // areas start on IN_ALIGN words, columns on 64-byte lines an odd number of lines apart; the allowance pays for the rounding
  tok_off = (pos * IN_R + (u64)k * IN_PAD) & ~(IN_ALIGN - 1);
  u64 c16 = (span * IN_R) / 1024;  // whole lines per column
  if (c16 > 0x003fffffu) c16 = 0x003fffffu;
  if (!(c16 & 1)) c16 = c16 ? c16 - 1 : 0;
  col_cap = c16 ? (u32)c16 * 16 : 8u;
  dir_off = pos / 32 + (u64)k * 64;
  const u64 dc = span / 32 + 63;
  dir_cap = dc > 0xffffffffu ? 0xffffffffu : (u32)dc;
}

struct ParStats { u32 windows, rounds, fallbacks, partial; u32 cyc[8]; u32 dbg; };

// Per-lane LSB-first bit reader over the staged ring.  The stream continues at bit `sh` of the 64-bit pair (hi:lo);
// `nextw` is the dword after `hi`, always already requested from LDS, so a refill never waits on the critical path.
// v_alignbit_b32 extracts 32 stream bits in one op.  `ptr` is the RING index of nextw and just counts up: a unit starts
// below RING_DW and ends inside the mirror at the latest.  `off` turns ring coordinates back into the stream position.
struct LaneBits { u32 lo, hi, nextw, sh, ptr, off; };
// p: bit position (epoch coordinates); rw: ring index of the dword that holds it (< RING_DW)
AHIP_DEVINL void lb_init(LaneBits &d, const u32 *inbuf, u32 p, u32 rw) {
  d.lo = inbuf[rw];
  d.hi = inbuf[rw + 1];
  d.nextw = inbuf[rw + 2];
  d.ptr = rw + 2;
  d.sh = p & 31;
  d.off = (p & ~31u) - (rw + 2) * 32;
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
AHIP_DEVINL u32 lb_pos(const LaneBits &d) { return d.ptr * 32 + d.sh + d.off; }

// One token at the lane's cursor, straight-line: every lane runs the litlen AND the distance half (a 64-lane step
// almost always contains a match anyway); selects pick the result.  The only branches skip the long-code
// resolution when no lane needs it.  Returns the token -- literal (0x8000 | byte) << 16, match len << 16 | dist --
// or, for anything else (end of block, bad litlen / distance symbol, unfilled entry), a word that is not negative
// and has a zero distance field; `e` then says which.
AHIP_DEVINL u32 decode_token(LaneBits &d, const WaveLds &L, const u32 *inbuf, u32 &e) {
  lb_normalize(d, inbuf);
  const u32 w = lb_peek32(d);  // 32 valid bits; litlen code + extra <= 20
  e = L.ll[w & ((1u << LL_ROOT) - 1)];
  if (AHIP_ANY_HINT(e & E_LONG)) {
    AHIP_ASM_NOTE("long litlen code");  // keep this a real branch: if-converted, its LDS read would sit on every step's critical path
    if (e & E_LONG) e = long_lookup(L.ll_sub, e, w, LL_ROOT);
  }
  const u32 cl = e & 15;
  const u32 xb = (e >> 4) & 15;  // 0 unless a length symbol
  const u32 lenv = (e >> 16) + __builtin_amdgcn_ubfe(w, cl, xb);
  const bool is_match = (e & (E_LIT | E_EOB | E_BAD | E_HOLE)) == 0;
  d.sh += cl + xb;
  lb_normalize(d, inbuf);
  const u32 w2 = lb_peek32(d);  // distance code + extra <= 28
  u32 t = L.dt[w2 & ((1u << D_ROOT) - 1)];
  if (AHIP_ANY_HINT(t & E_LONG)) {  // (whatever the lane's litlen symbol was: a spurious second-level read is harmless and costs less than asking)
    AHIP_ASM_NOTE("long distance code");
    if (t & E_LONG) t = long_lookup(L.d_sub, t, w2, D_ROOT);
  }
  const u32 dl = t & 15;
  const u32 dxb = (t >> 4) & 15;
  const u32 dist = (t >> 16) + __builtin_amdgcn_ubfe(w2, dl, dxb);  // 0 for the symbols 30 / 31
  d.sh += is_match ? dl + dxb : 0u;
  return (lenv << 16) | (is_match ? dist : 0u);
}

constexpr u32 LR_EOB = 1, LR_ERR = 2;  // how a run ended before its boundary

// ------------------------------------------------------------------------------------------
// Tokenizer side
// ------------------------------------------------------------------------------------------

// ---- the serial writer: one lane-uniform token at a time into the free space of the columns ----
// Its runs are always flagged DF_BIG (they may be of any length; the resolver takes their offsets from a prefix sum).
AHIP_DEVINL void sink_close(TokSink &k, u32 &colreg, int lane, u64 out_rel) {  // close the open run (out_rel: output produced so far), note how far its column is used
  if (!k.area || k.scol >= 64) return;
  if (k.spos > k.srun) {
    if (k.ndir < k.dir_cap) {
      const u64 base = out_rel - k.sbytes;
      if (lane == 0) k.dir[k.ndir] = make_uint4(k.srun, (k.spos - k.srun) | DF_BIG, (u32)base, (u32)(base >> 32));
      k.ndir++;
    } else k.full = true;
  }
  col_set(colreg, k.scol, k.spos - k.scol * k.col_cap, lane);
  k.srun = k.spos;
  k.sbytes = 0;
}
// (re)start: the flow decoder may have used this column meanwhile
AHIP_DEVINL void sink_open(TokSink &k, u32 colreg) {
  k.sbytes = 0;
  if (!k.area || k.scol >= 64) return;
  k.spos = k.scol * k.col_cap + col_get(colreg, k.scol);
  k.srun = k.spos;
}
AHIP_DEVINL bool sink_room(TokSink &k, u32 &colreg, u32 words, int lane, u64 out_rel) {  // `words` contiguous words
  for (;;) {
    if (k.scol < 64 && k.spos + words <= (k.scol + 1) * k.col_cap && k.spos - k.srun < (1u << 22)) return true;
    const bool split = k.scol < 64 && k.spos + words <= (k.scol + 1) * k.col_cap;  // only the run got too long
    sink_close(k, colreg, lane, out_rel);
    if (split) continue;
    k.scol += 1;  // ~0u -> 0
    if (k.scol >= 64) { k.scol = 64; k.full = true; return false; }
    k.spos = k.scol * k.col_cap + col_get(colreg, k.scol);
    k.srun = k.spos;
  }
}
// one token: `step_word` as the decode step makes it (literal / match), `adv` the bytes it produces; out_rel = the
// output offset in FRONT of it
AHIP_DEVINL void sink_put(TokSink &k, u32 &colreg, u32 step_word, u32 adv, int lane, u64 out_rel) {
  if (!k.area || k.full) return;
  if (!sink_room(k, colreg, 1, lane, out_rel)) return;
  k.sbytes += adv;
  if (lane == 0) k.area[k.spos] = rec_word(k.sbytes, step_word);
  k.spos += 1;
}

// Serial decode of one Huffman block that EMITS tokens instead of writing bytes: the checked path
// for everything irregular (same decisions, in the same order, as huffman_token<WRITE, CAREFUL>).
template <bool CAREFUL>
AHIP_DEVINL u32 huffman_token_emit(WaveLds &L, BitCursor &b, OutCursor &o, TokSink &sink, u32 &colreg, u32 ll_max, u32 d_max, int lane) {
  if (CAREFUL && b.pos + ll_max > b.total_bits) return 100 + MS_FALSE_EOS;
  u64 w = peek_bits(b);
  u32 e = uniform(L.ll[(u32)w & ((1u << LL_ROOT) - 1)]);
  if (e & E_LONG) e = uniform(long_lookup(L.ll_sub, e, (u32)w, LL_ROOT));
  u32 cl = e & 15;
  if (e & (E_LIT | E_EOB | E_BAD | E_HOLE)) {
    if (e & E_LIT) {
      if (o.pos >= o.limit) return 100 + MS_CAP;
      sink_put(sink, colreg, e & 0xffff0000u, 1u, lane, o.pos - o.org);  // (0x8000 | byte) << 16
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
  if (d & E_LONG) d = uniform(long_lookup(L.d_sub, d, (u32)w, D_ROOT));
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
  if ((u64)dist > o.pos - o.hist) o.far = 1;
  sink_put(sink, colreg, ((u32)len << 16) | (u32)dist, (u32)len, lane, o.pos - o.org);
  o.pos += (u64)len;
  return 0;
}
AHIP_DEVINL u32 huffman_block_emit(WaveLds &L, BitCursor &b, OutCursor &o, TokSink &sink, u32 &colreg, int lane) {
  const u32 ll_max = L.lld.maxlen, d_max = L.dd.maxlen;
  sink_open(sink, colreg);
  u32 rs;
  for (;;) {
    u32 r;
    if ((b.pos >> 3) + 16 <= b.in_len) r = huffman_token_emit<false>(L, b, o, sink, colreg, ll_max, d_max, lane);
    else r = huffman_token_emit<true>(L, b, o, sink, colreg, ll_max, d_max, lane);
    if (r == 0) continue;
    rs = r == 1 ? (u32)MS_OK : r - 100;
    break;
  }
  sink_close(sink, colreg, lane, o.pos - o.org);
  return rs;
}
// _parseUncompressedBlock as a directory entry: a block of >= 3 bytes becomes one DF_STORED entry (two area words hold
// its input offset); shorter ones are literals
AHIP_DEVINL u32 stored_block_emit(BitCursor &b, OutCursor &o, TokSink &sink, u32 &colreg, int lane) {
  b.pos = (b.pos + 7) & ~7ull;
  b.blen = 0;
  int len = read_bits(b, 16);
  int nlen_raw = read_bits(b, 16);
  int nlen = nlen_raw ^ 0xffff;
  if (len != 0 && len != nlen) return (len < 0 || nlen_raw < 0) ? MS_FALSE_EOS : MS_FALSE;
  if (nlen_raw < 0) b.pos = b.total_bits;  // an empty block whose NLEN the input's end cuts short: see stored_block()
  u64 byte = b.pos >> 3;
  if ((u64)len > b.in_len - byte) return MS_FALSE;
  if (o.pos + (u64)len > o.limit) return MS_CAP;
  sink_open(sink, colreg);
  const u64 out_rel = o.pos - o.org;
  if (len >= 3) {
    if (sink.area && !sink.full && sink_room(sink, colreg, 2, lane, out_rel)) {
      if (sink.ndir < sink.dir_cap) {
        if (lane == 0) {
          sink.area[sink.spos] = (u32)byte;
          sink.area[sink.spos + 1] = (u32)(byte >> 32);
          sink.dir[sink.ndir] = make_uint4(sink.spos, (u32)len | DF_STORED, (u32)out_rel, (u32)(out_rel >> 32));
        }
        sink.ndir++;
      } else sink.full = true;
      sink.spos += 2;
      sink.srun = sink.spos;  // the two words belong to the entry above, not to a token run
    }
    sink_close(sink, colreg, lane, out_rel);
  } else {
    for (int i = 0; i < len; ++i) sink_put(sink, colreg, TK_LIT | ((u32)b.in[byte + i] << 16), 1u, lane, out_rel + (u64)i);
    sink_close(sink, colreg, lane, out_rel + (u64)len);
  }
  o.pos += (u64)len;
  b.pos += 8ull * (u64)len;
  return MS_OK;
}

// Decode one Huffman block (tables already built in L) starting at b.pos into tokens.
// Returns MS_* exactly like huffman_block().
//
// Continuous-flow decode.  The block's bitstream is cut into ITEMS of SUB_BITS; the 64 lanes are independent
// workers that pick up work units at scheduling points (every STEPS decode steps) and never wait for each other
// inside an item:
//   SPEC(s)  blind run over the last SPEC_BITS of item s (Huffman streams self-synchronise): its end is the
//            PREDICTED start of item s+1.  Nothing is recorded.
//   RUN(s)   decode from a start (predicted, or true) to the end of item s, recording tokens into the lane's column of
//            the member's token area (TokSink), where they stay.
// A scoreboard in LDS (TokLds::fa/fb/fc/need/spec, indexed by item mod ITEMS) holds what the runs found.  At every
// scheduling point the wave
//   publishes  the runs that ended,
//   validates  in order: item V is final when its run started exactly where item V-1's final run ended (64 items
//              per step: one LDS read per lane, a DPP shift, a ballot).  A mispredicted item is simply put back
//              with its true start; only items that used its (wrong) end repeat,
//   retires    final items EMIT_MIN at a time: their runs are entered into the member's run directory (stream order);
//              the freed ring slots are refilled with the next compressed bytes,
//   assigns    idle lanes: the true-start run of item V first, then runs whose predicted start is known, then
//              speculation ahead (not beyond `hint_end_bits`, the end of the member when the index knows it).
// Lanes therefore stay busy across what used to be window boundaries: a token is decoded about 1.6 times
// (speculation over half an item + the run + a few repeats) at near-full lane occupancy.
// Anything irregular on the true path -- bad symbol, back-reference before the start of the output, output
// window exhausted, a full column, input too close to its end -- stops the flow at the last retired item and
// hands the rest of the block to the serial decoder, which restates the reference symbol by symbol.
AHIP_DEVINL u32 huffman_block_tokenize(WaveLds &L, TokLds &P, BitCursor &b, OutCursor &o, TokSink &sink, u32 &colreg, int lane,
                                       ParStats &st, u64 hint_end_bits) {
  constexpr u32 SLACK_DW = 4;  // a token may run 48 bits past its item and the reader looks two dwords ahead
  constexpr u32 SUB = SUB_BITS;
  u32 eguard = 0;
  for (;;) {  // epochs: positions are 32-bit offsets from the epoch origin
    if (++eguard > (1u << 16)) { st.dbg |= 1; break; }
    const bool emit = sink.area != nullptr && !sink.full;  // (fixed for the epoch: the decode loop below is compiled for either case)
    bool give_up_tokens = false;
    const u64 gbyte = (b.pos >> 3) & ~3ull;
    if (gbyte + (u64)(2 * SUB_DW + SLACK_DW) * 4 > b.in_len) break;  // too close to the end: checked serial path
    const u64 avail_dw = (b.in_len - gbyte) >> 2;
    const u64 ni64 = (avail_dw - SLACK_DW) / SUB_DW;
    const u32 n_items = ni64 > EPOCH_ITEMS ? EPOCH_ITEMS : (u32)ni64;  // items this epoch may decode (>= 2)
    u32 n_spec = n_items;                                               // items the member is expected to reach into
    if (hint_end_bits > gbyte * 8) {
      const u64 h = (hint_end_bits - gbyte * 8 + SUB - 1) / SUB;
      if (h < n_spec) n_spec = (u32)h;
    }
    const u32 spec_cap = (n_spec ? n_spec - 1 : 0u) < n_items - 1 ? (n_spec ? n_spec - 1 : 0u) : n_items - 1;  // SPEC(s) serves item s+1
    AHIP_TICK(t_a);
    for (u32 i = lane; i < (u32)ITEMS; i += 64) P.fa[i] = 0;
    const u32 total_dw = n_items * SUB_DW + SLACK_DW;
    // (Measured and not kept, profiles/r04_experiments.md: filling the ring in units whose data was requested one retire
    //  step ahead -- the retire step's wait for its loads is not what the phase costs; smaller retire batches cost more
    //  in per-batch work than their fewer decode steps save.)
    u32 stage_hi = 0;  // dwords staged so far (multiple of 4)
    auto stage_to = [&](u32 target) {
      for (u32 w = stage_hi + (u32)lane * 4; w < target; w += 256) {
        const u32 rw = w % (u32)RING_DW;
        const uint4 v = load_u128_unaligned(b.in + gbyte + 4 * (u64)w);
        *(uint4 *)(P.inbuf + rw) = v;
        if (rw < (u32)RING_MIRROR) *(uint4 *)(P.inbuf + RING_DW + rw) = v;  // the head of the ring, again behind its end
      }
      stage_hi = target;
    };
    stage_to(total_dw < (u32)RING_DW ? total_dw : (u32)RING_DW);
    wave_sync();
    AHIP_TICK(t_b);
    AHIP_ACC(st.cyc[1], t_a, t_b);
    st.windows++;
    const u32 t0 = (u32)(b.pos - gbyte * 8);  // true start of item 0
    // wave-uniform scheduler state
    u32 next_spec = 0, next_fix = 1, V = 0, tV = t0, retired = 0, t_ret = t0, g = 0;
    bool block_done = false, stop_serial = false;
    // lane state: ms = mode<<28 | item (mode 0 idle, 1 SPEC, 2 RUN)
    u32 ms = 0, bound = 0, start = 0, endp = 0, fl = 0, nbytes = 0, row0 = 0, myslot = 0;  // myslot: scoreboard slot of the lane's item
    u32 rowctr = colreg;  // words this lane has recorded into its column of the token area
    i32 need = 0;
    LaneBits d{0, 0, 0, 0, 2, 0};
    u32 *const col = sink.area + (u32)lane * sink.col_cap;  // this lane's column of the member's token area
    u32 rot = 0;  // rotates which idle lanes take the recording runs, so that the columns fill evenly
    u32 guard = 0;
    for (;;) {
      if (++guard > 4 * n_items + 4096) {  // never spin: the serial decoder takes over at the last retired item
        st.dbg |= 2;
#ifdef AHIP_FLOW_DEBUG
        st.cyc[0] = V; st.cyc[1] = retired; st.cyc[2] = next_fix; st.cyc[3] = next_spec; st.cyc[4] = g; st.cyc[5] = n_items;
        st.cyc[6] = uniform(P.fa[V % (u32)ITEMS]); st.cyc[7] = (block_done ? 1u : 0u) | (u32)__popcll(__ballot((ms >> 28) != 0)) << 8 | stage_hi << 16;
#endif
        stop_serial = true;
        break;
      }
      AHIP_TICK(t_s0);
      // ===== scan the window [V, V + 64): validate in order, find the runs that have to be repeated =====
      // pend = where this item's run has to start: the end of its predecessor's finished run (tV, the truth, for
      // item V).  An item is final when its run started there and its predecessor is final.  A run that started
      // somewhere else -- or an item that lost its run -- is decoded again from pend as soon as pend is known,
      // not only when validation gets there.
      u64 rm = 0;     // lanes whose item needs a (new) run
      u32 pend = 0;   //   and where it starts
      if (!block_done && !stop_serial) {
        const u32 wi = V + (u32)lane;
        const u32 wslot = wrap_item(V % (u32)ITEMS + (u32)lane);  // (V is wave-uniform: the division is scalar)
        const bool win = wi < n_items && wi < retired + (u32)ITEMS;
        const u32 wa = win ? P.fa[wslot] : (1u << 30);
        const u32 wstate = wa >> 30, wfl = (wa >> 28) & 3u;
        const u32 wend = wi * SUB + (wa & 0xfffu), wstart = wi * SUB + ((wa >> 12) & 0x3ffu);
        const u32 wkey = (wstate == 2 && wfl == 0) ? wend : ~0u;  // usable end of a finished run
        pend = lane_prev(wkey);  // DPP: every lane enabled here
        pend = lane == 0 ? tV : pend;
        const bool ok = wstate == 2 && wstart == pend;
        const u64 okm = __ballot(ok);
        u32 n = okm == ~0ull ? 64u : (u32)__builtin_ctzll(~okm);
        const u64 flm = __ballot(ok && wfl != 0) & (n == 64 ? ~0ull : ((1ull << n) - 1));
        u32 stop_flags = 0;
        if (flm) {
          const int j = __builtin_ctzll(flm);
          stop_flags = lane_bcast(wfl, j);
          n = (stop_flags & LR_EOB) ? (u32)j + 1 : (u32)j;  // an item that ends in a bad symbol is left to the serial decoder
        }
        if (n) { tV = lane_bcast(wend, (int)n - 1); V += n; }
        if (stop_flags & LR_EOB) block_done = true;
        else if (stop_flags) stop_serial = true;
        if ((ms >> 28) == 1 && (ms & 0x0fffffffu) < V) { ms = 0; bound = 0; endp = 0; }  // speculation nobody needs any more
        if (next_fix <= V) next_fix = V + 1;
        const bool rerun = win && (u32)lane >= n && pend != ~0u && wi < next_fix && (wstate == 0 || wstate == 3 || (wstate == 2 && wstart != pend));
        rm = (block_done || stop_serial) ? 0ull : __ballot(rerun);
      }
      AHIP_TICK(t_s1);
      AHIP_ACC(st.cyc[3], t_s0, t_s1);
      // ===== retire: the runs of final items are entered into the run directory, the ring is refilled =====
      const bool finishing = block_done || stop_serial || V >= n_items;
      while (V - retired >= (finishing ? 1u : EMIT_MIN)) {
        const u32 nb = V - retired < 64 ? V - retired : 64u;
        const bool mine = (u32)lane < nb;
        const u32 s = retired + (u32)lane;
        const u32 sslot = wrap_item(retired % (u32)ITEMS + (u32)lane);
        const u32 a = mine ? P.fa[sslot] : 0u;
        const u32 r0 = mine ? P.fb[sslot] : 0u;
        const u32 c = mine ? P.fc[sslot] : 0u;
        const u32 nd = mine ? (u32)P.need[sslot] : 0u;
        const u32 cnt = c >> 20, nby = c & 0xfffffu, cl = (a >> 22) & 63u;
        u32 tot_bytes;
        const u32 B = wave_excl_sum(nby, tot_bytes);
        const u64 hm = __ballot(mine && cnt != 0);
        const u32 nh = (u32)__popcll(hm);
        const bool bad_cap = (u64)tot_bytes > o.limit - o.pos;         // output window exhausted
        const bool bad_far = __any(mine && (u64)nd > o.pos + B) != 0;  // back-reference before the start of the output
        // (a sizing run that keeps its tokens only has the room the COMPRESSED size suggests: when that runs out the
        //  tokens are given up -- the member is tokenized again by the decode proper -- and the flow starts a new
        //  epoch that only counts)
        const bool bad_dir = emit && sink.ndir + nh > sink.dir_cap;    // directory full
        if (bad_dir && sink.sizing) { sink.full = true; give_up_tokens = true; stop_serial = true; break; }
        if (bad_cap || bad_far || bad_dir) {
          st.fallbacks++;
          st.dbg |= bad_cap ? 4u : 0u; st.dbg |= bad_far ? 8u : 0u; st.dbg |= bad_dir ? 16u : 0u;
          if (bad_dir) sink.full = true;
          stop_serial = true;
          break;
        }
        if (__any(mine && (u64)nd > o.pos - o.hist + B)) o.far = 1;  // reaches into earlier output (q8): resolved late
        if (emit) {
          if (mine && cnt != 0) {
            const u64 ob = o.pos - o.org + B;  // where the run's first byte goes, counted from the member's own first byte
            sink.dir[sink.ndir + wave_rank(hm)] =
                make_uint4(cl * sink.col_cap + r0, cnt | (nby > 0xffffu ? DF_BIG : 0u), (u32)ob, (u32)(ob >> 32));
          }
          sink.ndir += nh;
        }
        o.pos += tot_bytes;
        t_ret = lane_bcast(s * SUB + (a & 0xfffu), (int)nb - 1);
        if (mine) P.fa[sslot] = 0;
        retired += nb;
        const u32 room = retired * SUB_DW + (u32)RING_DW;
        wave_sync();
        stage_to(total_dw < room ? total_dw : room);
        wave_sync();
      }
      // a recording lane must not run out of column: the serial decoder takes over with what all columns have left
      if (emit && !finishing && __any(rowctr + (u32)STEPS > sink.col_cap)) {
        if (sink.sizing) { sink.full = true; give_up_tokens = true; stop_serial = true; }
        else { st.fallbacks++; st.dbg |= 32; stop_serial = true; }
      }
      AHIP_TICK(t_s2);
      AHIP_ACC(st.cyc[4], t_s1, t_s2);
      if (stop_serial || (finishing && retired == V)) break;
      // ===== assign work to idle lanes: repairs first, then runs with a predicted start, then speculation =====
      {
        const bool idle = (ms >> 28) == 0;
        const u64 im = __ballot(idle);
        if (im) {
          const u32 nidle = (u32)__popcll(im);
          if (rot >= nidle) rot = 0;
          u32 rank = wave_rank(im) + rot;
          rank = rank >= nidle ? rank - nidle : rank;
          rot += 7;
          const u32 stage_items = (stage_hi - SLACK_DW) / SUB_DW;  // items whose bits (+ slack) are in the ring
          const u32 lim = n_items < stage_items ? n_items : stage_items;
          // (a) repairs (ordered: the one validation waits for comes first)
          // (a repair's item lies in [V, V + 64) -- V as the scan above has just advanced it --, its start less than 48 bits
          //  behind the item's boundary: the queue holds it as an offset from the window's first item)
          const u32 vslot = V % (u32)ITEMS;
          const bool rr = ((rm >> lane) & 1) && pend / SUB < lim;
          const u64 rm2 = __ballot(rr);
          const u32 nrr = (u32)__popcll(rm2) < REPAIRS ? (u32)__popcll(rm2) : REPAIRS;
          const u32 na = nrr < nidle ? nrr : nidle;
          const u32 rr_rank = wave_rank(rm2);
          if (rr && rr_rank < na) { P.q[rr_rank] = (u16)(pend - V * SUB); P.fa[wrap_item(vslot + (pend / SUB - V))] = 1u << 30; }
#ifdef AHIP_PROFILE
          st.rounds += na;  // (repairs: a diagnostic)
#endif
#ifdef AHIP_FLOW_STATS
          st.cyc[4] += na;
#endif
          // (b) runs whose predicted start is known: consecutive items from next_fix
          const u32 sb = next_fix + (u32)lane;
          const u32 fslot = next_fix % (u32)ITEMS;
          const u32 bfa = sb < lim && sb < retired + (u32)ITEMS ? P.fa[wrap_item(fslot + (u32)lane)] : 0u;
          const bool okb = (bfa >> 30) == 3u;  // a predicted start is known
          const u64 okm = __ballot(okb);
          const u32 nready = okm == ~0ull ? 64u : (u32)__builtin_ctzll(~okm);
          // (c) speculation ahead
          if (next_spec < V) next_spec = V;
          const u32 slim = spec_cap < lim ? spec_cap : lim;
          const u32 nsp = slim > next_spec ? slim - next_spec : 0u;
          const u32 nb_ = nready < nidle - na ? nready : nidle - na;
          const u32 nc_ = nsp < nidle - na - nb_ ? nsp : nidle - na - nb_;
          const u32 u2 = rank - na;  // wraps for the lanes that take (a); unused there
          const bool take_a = idle && rank < na;
          const bool take_b = idle && rank >= na && u2 < nb_;
          const bool take_c = idle && rank >= na && !take_b && u2 - nb_ < nc_;
          wave_sync();  // P.q
          if (take_a || take_b || take_c) {
            u32 s_new, st_new, slot_new;
            if (take_a) {
              const u32 rel = P.q[rank];  // bits behind the window's first item
              st_new = V * SUB + rel; s_new = V + rel / SUB; slot_new = wrap_item(vslot + rel / SUB);
            } else if (take_b) {
              s_new = next_fix + u2; slot_new = wrap_item(fslot + u2);
              st_new = s_new * SUB + (P.fa[slot_new] & 0xfffu);
            } else {
              s_new = next_spec + (u2 - nb_); slot_new = wrap_item(next_spec % (u32)ITEMS + (u2 - nb_));
              st_new = (s_new + 1) * SUB - SPEC_BITS;
              if (st_new < t0) st_new = t0;  // item 0 of the epoch: nothing in front of the true start belongs to this block
            }
            ms = (take_c ? 1u << 28 : 2u << 28) | s_new;
            myslot = slot_new;
            bound = (s_new + 1) * SUB;
            start = st_new; endp = st_new; fl = 0; nbytes = 0; need = 0; row0 = rowctr;
            // (the unit starts inside its item or, a repair / predicted start, less than 64 bits behind its end)
            lb_init(d, P.inbuf, st_new, wrap_ring(slot_new * (u32)SUB_DW + ((st_new - s_new * SUB) >> 5)));
            if (take_b) P.fa[slot_new] = 1u << 30;  // in flight
          }
          next_fix += nb_;
          next_spec += nc_;
          wave_sync();
        }
      }
      AHIP_TICK(t_s3);
      AHIP_ACC(st.cyc[3], t_s2, t_s3);
      // ===== decode steps =====  (idle and finished lanes have bound == 0)
#ifdef AHIP_FLOW_STATS  // dev (CPU emulation): what the lanes did -- cyc[0] lane-steps, [1] of them speculative, [2] scheduling points,
                        // [3] lanes left idle by the assignment, [4] repairs
      st.cyc[2] += 1;
      st.cyc[3] += (u32)__popcll(__ballot((ms >> 28) == 0));
#endif
      for (int k = 0; k < STEPS; ++k) {
        const bool go = endp < bound;
        if (!__any(go)) break;
#ifdef AHIP_FLOW_STATS
        st.cyc[0] += (u32)__popcll(__ballot(go));
        st.cyc[1] += (u32)__popcll(__ballot(go && (ms >> 28) == 1));
#endif
        if (go) {
          u32 e;
          const u32 t = decode_token(d, L, P.inbuf, e);
          endp = lb_pos(d);
          const bool spc = (i32)t >= 0 && (t & 0xffffu) == 0;
          if (AHIP_ANY_HINT(spc)) {
            AHIP_ASM_NOTE("special symbol");
            if (spc) { fl = (e & E_EOB) ? LR_EOB : LR_ERR; bound = 0; }
          }
          if (!spc) {
            const bool lit = (i32)t < 0;
            const i32 req = (i32)(t & 0xffff) - (i32)nbytes;  // (a literal has no distance: never positive)
            need = req > need ? req : need;
            nbytes += lit ? 1u : (t >> 16);
#ifdef AHIP_ABL_NO_TOKSTORE  // dev ablation: what the token stores cost
            if (ms >> 29) rowctr += 1;
#else
            if (ms >> 29) { if (emit) col[rowctr] = rec_word(nbytes, t); rowctr += 1; }
#endif
          }
        }
        ++g;
      }
      AHIP_TICK(t_s4);
      AHIP_ACC(st.cyc[2], t_s3, t_s4);
      // ===== publish the runs that ended =====
      {
        const u32 mode = ms >> 28, s = ms & 0x0fffffffu;
        const bool fin = mode != 0 && !(endp < bound);
        if (__any(fin)) {
          if (fin) {
            const u32 bnd = (s + 1) * SUB;
            if (mode == 1) {
              // the predicted start of the NEXT item goes into that item's slot -- if it lies inside the ring's window and
              // nobody has given it a run meanwhile (an unusable speculation leaves nothing: the item is decoded when its
              // predecessor's run has ended)
              const bool usable = fl == 0 && endp - bnd < 64 && s + 1 < retired + (u32)ITEMS && s + 1 < n_items;
              const u32 nslot = wrap_item(myslot + 1);
              if (usable && (P.fa[nslot] >> 30) == 0u) P.fa[nslot] = (3u << 30) | (endp - bnd);
            } else {
              P.fb[myslot] = row0;
              P.fc[myslot] = nbytes | (((rowctr - row0) & 0xfffu) << 20);
              P.need[myslot] = (u16)need;
              P.fa[myslot] = (2u << 30) | (fl << 28) | ((u32)lane << 22) | ((start - s * SUB) << 12) | (endp - s * SUB);
            }
            ms = 0; bound = 0; endp = 0;
          }
          wave_sync();
        }
      }
    }
    colreg = rowctr;
    b.pos = gbyte * 8 + t_ret;
#ifdef AHIP_PROFILE
    st.partial += g;  // decode steps of the wave
#endif
    if (give_up_tokens) continue;  // a new epoch from the last retired item, counting only
    if (stop_serial) { if (!(st.dbg & 2)) { /* counted at the site */ } break; }
    if (block_done) return MS_OK;
    // the epoch is used up (V == n_items): go on from the new origin
  }
  AHIP_TICK(t_x0);
  u32 rs = huffman_block_emit(L, b, o, sink, colreg, lane);
  AHIP_TICK(t_x1);
  AHIP_ACC(st.cyc[7], t_x0, t_x1);
  return rs;
}

// Inflate one stream.  Mirrors Inflate._inflate(): loop blocks until BFINAL, an error, or EOS.
//  PAR = false: the serial byte-writing decoder (A/B baseline, single kernel).
//  PAR = true : tokenizer -- no bytes are written; tokens go to `sink` (sink.area == nullptr = sizing run).
//  CHUNK = true: one chunk of a long stream (ChunkCtx): bit-granular start, `hist` bytes of earlier output count
//  as already produced (so the back-reference range check holds across the chunk boundary), and the loop stops
//  in front of a block header that sits on a candidate position; end_pos is then reported in BITS.
//  exact != nullptr: over-subscribed code lengths are decoded with the reference's own table (serial decoder only)
template <bool WRITE, bool PAR, bool CHUNK = false>
AHIP_DEVINL void inflate_member(WaveLds &L, HeaderLds &H, TokLds *P, const u8 *in, u64 in_len, const MemberDesc &m, u8 *out,
                                TokSink sink, MemberResult &res, int lane, const ChunkCtx *cx = nullptr,
                                u32 *exact = nullptr) {
  ParStats st{};
  BitCursor b{in, in_len, in_len * 8, m.in_off * 8 + (CHUNK ? cx->start_bit : 0u), nullptr, 0, 0, 0};
  const u64 hist = CHUNK ? cx->hist : m.hist;
  OutCursor o{out + m.out_off - hist, hist, m.out_limit > ~0ull - hist ? ~0ull : m.out_limit + hist, CHUNK ? (cx->watch ? hist : 0u) : hist, 0, hist};
  u32 colreg = 0;  // words this lane's column of the member's token area holds (tokenizer only)
  // where the index expects the deflate data to end (a hint for speculation only: the decode itself never trusts it)
  const u64 hint_end_bits = (!CHUNK && m.expect_end != ~0ull) ? m.expect_end * 8 : 0ull;
  u32 status = MS_EOS, blocks = 0;
  for (;;) {
    if (((b.pos + 7) >> 3) >= in_len) { status = MS_EOS; break; }
    if (CHUNK && blocks) {  // a later chunk takes over at this block?
      u32 lo = 0, hi = cx->n_cand;
      while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (cx->cand_bits[mid] < b.pos) lo = mid + 1; else hi = mid; }
      if (lo < cx->n_cand && cx->cand_bits[lo] == b.pos) { status = MS_CHUNK_END; break; }
    }
    b.blen = (8 - ((u32)b.pos & 7)) & 7;  // what the accumulator holds between blocks: the rest of the current byte
    int hdr = read_bits(b, 3);
    ++blocks;
    const bool final_block = hdr & 1;
    const int btype = hdr >> 1;
    u32 r;
    if (btype == 0) {
      r = PAR ? stored_block_emit(b, o, sink, colreg, lane) : stored_block<WRITE>(b, o, lane);
    } else if (btype == 3) {
      r = MS_FALSE;
    } else {
      int hlit = 288, hdist = 30;
      r = MS_OK;
      AHIP_TICK(t_h0);
      if (btype == 1) fixed_lengths(H.lens, lane);
      else {
        const u64 sb = (b.pos >> 3) & ~3ull;
        constexpr u32 HDR_STAGE = 640;  // a dynamic header is at most 562 bytes
#ifndef AHIP_HEADER_LDS
        bool fast_ok = false;
        if (sb + HDR_STAGE <= in_len) {
          const BitCursor b0 = b;
          r = dynamic_header_fast(H, b, in + sb, sb, lane, hlit, hdist);
          fast_ok = r == MS_OK;
          if (!fast_ok) b = b0;  // a damaged header: once more through the tracked path below
        }
        if (!fast_ok)
#endif
        {
          if (PAR && sb + HDR_STAGE <= in_len) {  // stage the header into the idle ring buffer
            for (u32 k = lane * 4; k < HDR_STAGE / 4; k += 256)
              *(uint4 *)(P->inbuf + k) = load_u128_unaligned(in + sb + 4 * k);
            wave_sync();
            b.stage = P->inbuf; b.stage_byte = sb; b.stage_len = HDR_STAGE;
          }
          r = dynamic_header(H, b, lane, hlit, hdist);
          b.stage = nullptr;
        }
      }
      const u64 data_pos = b.pos;
      const u32 data_blen = b.blen;
      bool replayable = false;
      AHIP_TICK(t_hm);
      AHIP_ACC(st.cyc[5], t_h0, t_hm);
      if (r == MS_OK) {
        bool ok = build_decode_table<false>(H.lens, hlit, L.ll, LL_ROOT, L.lld, L.ll_sub, LL_SUB, lane, (u16 *)H.cl);
        AHIP_TICK(t_hn);
        AHIP_ACC(st.cyc[6], t_hm, t_hn);
        ok &= build_decode_table<true>(H.lens + hlit, hdist, L.dt, D_ROOT, L.dd, L.d_sub, D_SUB, lane, (u16 *)H.cl);
        replayable = ok;
        AHIP_TICK(t_h1);
        AHIP_ACC(st.cyc[0], t_h0, t_h1);
        if (!ok) {
          if (!PAR && exact) {
            ExactTabs X{exact, exact + 32768, 0, 0};
            X.ll_max = build_exact_table(H.lens, hlit, X.ll, lane);
            X.d_max = build_exact_table(H.lens + hlit, hdist, X.dt, lane);
            r = huffman_block_exact<WRITE>(X, b, o, lane);
          } else {
            r = MS_OVERSUB;
          }
        }
        else if (PAR) r = huffman_block_tokenize(L, *P, b, o, sink, colreg, lane, st, hint_end_bits);
        else r = huffman_block<WRITE>(L, b, o, lane);
        if (r == MS_FALSE && replayable) {  // a bad symbol: where exactly did the reference's reader stop?
          b.pos = data_pos;
          b.blen = data_blen;
          replay_to_failure(L, b);
        }
      }
    }
    if (r != MS_OK) { status = r; break; }
    if (final_block) { status = MS_OK; break; }
  }
  if (lane == 0) {
    // Position the reference's InputStream is left at: after a complete block whole bytes are un-read
    // (inflate.dart:337-340); after a failure inside a block nothing is (the accumulator keeps what it pulled).
    u64 end = (b.pos + 7) >> 3;
    if (status == MS_FALSE) end = (b.pos + b.blen) >> 3;
    if (status == MS_FALSE_EOS) { end = in_len; status = MS_FALSE; }  // every byte was pulled into the accumulator
    res.end_pos = end > in_len ? in_len : end;
    if (CHUNK && status == MS_CHUNK_END) res.end_pos = b.pos;
    res.out_len = o.pos - hist;
    res.status = (PAR && sink.full && !sink.sizing) ? (u32)MS_TOKFULL : status;
    res.blocks = blocks | (((!CHUNK && o.far) || (PAR && sink.full && sink.sizing)) ? MR_FAR : 0u) | ((CHUNK && o.far) ? MR_REACH : 0u);
    res.windows = st.windows;
    res.rounds = st.rounds;
    res.fallbacks = st.fallbacks;
    res.partial = st.partial;
    for (int k = 0; k < 8; ++k) res.cyc[k] = st.cyc[k];
#ifndef AHIP_FLOW_DEBUG
    if (st.dbg) res.cyc[7] = 0xdead0000u | st.dbg;
#else
    if (st.dbg) res.blocks |= st.dbg << 16;
#endif
    res.tok_words = sink.ndir;  // runs in the directory
  }
}

// ------------------------------------------------------------------------------------------
// Resolver side: replay one member's token stream into its output window
// ------------------------------------------------------------------------------------------
// exactly `len` (1..16) bytes of the 16 in (w0, w1) to dp, any alignment: two overlapping unaligned stores [0, w) and
// [len - w, len), w = 8 / 4; gfx950's LDS takes unaligned 2/4/8-byte accesses (tools/micro/lds_unaligned.hip)
AHIP_DEVINL void deposit16(u8 *dp, u32 len, u64 w0, u64 w1) {
  if (len >= 8) {
    const u32 sh = 8 * (len - 8);  // 0..64
    const u64 tail = sh == 0 ? w0 : (sh == 64 ? w1 : ((w0 >> sh) | (w1 << (64 - sh))));
    ((unaligned_u64 *)dp)->v = w0;
    ((unaligned_u64 *)(dp + len - 8))->v = tail;
  } else if (len >= 4) {
    ((unaligned_u32 *)dp)->v = (u32)w0;
    ((unaligned_u32 *)(dp + len - 4))->v = (u32)(w0 >> (8 * (len - 4)));
  } else {  // a match is at least 3 bytes (RFC 1951; the reference's EOS quirk only shortens lengths >= 11): exactly 3 here
    ((unaligned_u16 *)dp)->v = (u16)w0;
    dp[2] = (u8)(w0 >> 16);
  }
}

// -DAHIP_PROFILE_RES: shader-clock cycles / 16 of the resolver's phases into cyc[0..7] (tools/kstats.py)
//   0 look setup  1 gather  2 prep  3 classify + literals + late fetch  4 rounds  5 hard matches  6 flush  7 whole member
#ifdef AHIP_PROFILE_RES
#define RTICK(var) const u64 var = __builtin_amdgcn_s_memtime()
#define RACC(slot, t0, t1) cyc[slot] += (u32)(((t1) - (t0)) >> 4)
#else
#define RTICK(var) do { } while (0)
#define RACC(slot, t0, t1) do { } while (0)
#endif

// -DAHIP_RES_STATS (CPU emulation only): event counts of the resolver, summed over members
#ifdef AHIP_RES_STATS
static unsigned long long res_stats[16];
#define RSTAT(i, n) do { if (lane == 0) res_stats[i] += (unsigned long long)(n); } while (0)
#else
#define RSTAT(i, n) do { } while (0)
#endif
#ifndef AHIP_WIN_CAP
#define AHIP_WIN_CAP 2560
#endif
#ifndef AHIP_WIN_KEEP
#define AHIP_WIN_KEEP 640
#endif
#ifndef AHIP_PEND_CAP
#define AHIP_PEND_CAP 384
#endif
constexpr u32 WIN_CAP = AHIP_WIN_CAP;              // bytes of output the LDS window holds
constexpr u32 WIN_FLUSH = WIN_CAP - AHIP_WIN_KEEP;  // ... and it is flushed once a chunk leaves it fuller than this
constexpr u32 PEND_CAP = AHIP_PEND_CAP;            // deferred matches per window
constexpr u32 LOOK_TOK = 4096;                     // tokens per look at the directory (a flow run has at most 4095)
static_assert(WIN_CAP % 32 == 0 && WIN_CAP <= 32768 && WIN_FLUSH >= 512 && AHIP_WIN_KEEP >= 264 && PEND_CAP >= 128, "window geometry");
// E = u8: bytes.  E = u16: symbols of the chunked single-stream decode (sm_inflate.hpp) -- a byte value, or
// SYM_MARK + j for "byte j of the 32 KiB of output in front of this chunk", not known yet when the chunk is resolved.
template <typename E>
struct ResLdsT {
  E obuf[WIN_CAP + 64] __attribute__((aligned(16)));  // + alignment offset (< 16 bytes) + a 16-element read past the last source element
  u32 pmap[WIN_CAP / 32 + 4];  // one bit per window byte: a deferred match has yet to write it
  uint2 plist[PEND_CAP];       // the window's deferred matches in stream order: {window index | len << 16, distance}
  u32 rbits[LOOK_TOK / 32 + 4];  // the current look at the directory: bit i = token i of the look is the first of its run
  uint2 rtab[64];              //   run r of the look: {area offset of its token 0 - index of that token in the look, output offset rel. borg}
};
using ResLds = ResLdsT<u8>;

// exactly `len` (3..16) SYMBOLS of the 16 in (w0..w3) to dp: two overlapping unaligned stores like deposit16(), in
// 2-byte units ([0, w) and [2 len - w, 2 len), w = 16 / 8 bytes; three symbols = 4 + 2 bytes)
AHIP_DEVINL void deposit16_sym(u16 *dp, u32 len, u64 w0, u64 w1, u64 w2, u64 w3) {
  u8 *bp = (u8 *)dp;
  const u32 b = 2 * len;
  if (len >= 8) {
    const u32 o = b - 16;  // 0, 2, .. 16: first byte of the last 16
    const bool up = o >= 8;
    const u64 x0 = up ? w1 : w0, x1 = up ? w2 : w1, x2 = up ? w3 : w2;
    const u32 sh = 8 * (o & 7);
    const u64 lo = o == 16 ? w2 : (sh ? (x0 >> sh) | (x1 << (64 - sh)) : x0);
    const u64 hi = o == 16 ? w3 : (sh ? (x1 >> sh) | (x2 << (64 - sh)) : x1);
    ((unaligned_u64 *)bp)->v = w0;
    ((unaligned_u64 *)(bp + 8))->v = w1;
    ((unaligned_u64 *)(bp + o))->v = lo;
    ((unaligned_u64 *)(bp + o + 8))->v = hi;
  } else if (len >= 4) {
    const u32 sh = 8 * (b - 8);  // 0, 16, 32, 48
    const u64 tail = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
    ((unaligned_u64 *)bp)->v = w0;
    ((unaligned_u64 *)(bp + b - 8))->v = tail;
  } else {  // three symbols (a match is at least 3 long)
    ((unaligned_u32 *)bp)->v = (u32)w0;
    ((unaligned_u16 *)(bp + 4))->v = (u16)(w0 >> 32);
  }
}

// One member: token runs (area, dir) -> bytes at out_base.  64 tokens per step, one per lane.
//   offsets  every lane knows where its token goes without a scan: the run's output offset (directory) + the `end` of
//            the token before it.  All lane offsets are relative to `borg`, the window position when the current look at
//            the directory began; `wrel` (scalar) = window position - borg, so window index = offset - wrel.
//   chunk    straight-line, nothing waits on LDS: literals store their byte; a match whose 16 (32) source bytes are
//            flushed output was fetched per token one chunk ahead (prep) and is deposited with exact-length unaligned
//            stores; a match whose source touches the window is DEFERRED: appended to the window's pending list, its
//            destination bytes marked in a bitmap.
//   flush    the deferred matches, 64 per step, in rounds: one is ready when no byte of its source is still marked
//            (exact dependences: about as many rounds as the longest chain of matches copying from matches); ready
//            ones read 16 (32) window bytes, deposit, and clear their marks.  Matches longer than 32 bytes, overlapping
//            their own source or straddling the window start are copied by the whole wave when they are the first
//            pending one (everything in front of their destination is final then).  Then the window goes to HBM with
//            16-byte stores.
// Returns false when a chunk could not be finished within its loop bound (cannot happen -- see process() -- but a bound that
// fires must not drop tokens silently: the caller turns it into MS_INTERNAL).
template <typename E>
AHIP_DEVINL bool resolve_member(ResLdsT<E> &P, const u8 *in, const u32 *area, const DirEnt *dir, u32 ndir, E *out_base, u32 *cyc,
                                int lane) {
  constexpr bool MARK = sizeof(E) == 2;
  constexpr u32 EPV = 16 / sizeof(E);        // elements per 16-byte vector
  constexpr u32 SIMPLE_MAX = MARK ? 16 : 32;  // longest match the two-piece (bytes) / one-piece (symbols) deposit takes
  u64 wpos = 0;                                   // output offset (member-relative) of the window's first byte
  u32 wfill = 0;                                  // bytes assembled in the window
  u32 npend = 0;                                  // deferred matches of the window
  u32 A = (u32)(((uintptr_t)out_base / sizeof(E)) & (EPV - 1));  // obuf[A + i] <-> out_base[wpos + i]: LDS and global congruent mod 16 bytes
  for (u32 i = lane; i < WIN_CAP / 32 + 4; i += 64) P.pmap[i] = 0;
  wave_sync();
  RTICK(r_begin);
  // marks of window bytes [b, b + len): set or cleared by the owning lane -- two dwords cover 32 bytes, longer ones loop
  auto mark = [&](u32 b, u32 len, bool set) {
    const u32 w = b >> 5;
    const u64 m = (len >= 32 ? 0xffffffffull : (1ull << len) - 1) << (b & 31);
    if (set) { atomicOr(&P.pmap[w], (u32)m); atomicOr(&P.pmap[w + 1], (u32)(m >> 32)); }
    else { atomicAnd(&P.pmap[w], ~(u32)m); atomicAnd(&P.pmap[w + 1], ~(u32)(m >> 32)); }
    if (len > 32) {
      const u32 e = b + len;
      for (u32 q = b + 32; q < e; q += 32) {  // (b + 32 .. e) in pieces of <= 32 bits, each inside two dwords
        const u32 n = e - q < 32 ? e - q : 32u;
        const u64 mq = (n >= 32 ? 0xffffffffull : (1ull << n) - 1) << (q & 31);
        if (set) { atomicOr(&P.pmap[q >> 5], (u32)mq); atomicOr(&P.pmap[(q >> 5) + 1], (u32)(mq >> 32)); }
        else { atomicAnd(&P.pmap[q >> 5], ~(u32)mq); atomicAnd(&P.pmap[(q >> 5) + 1], ~(u32)(mq >> 32)); }
      }
    }
  };
  // the first pending match, copied by the whole wave: every byte in front of its destination is final
  auto wave_copy = [&](u32 fd, u32 L_, u32 D_) {
    E *const ob = P.obuf + A;
    const i32 S_ = (i32)fd - (i32)D_;
    const E *gsrc = out_base + wpos;  // window index i < 0 <-> gsrc[i] ...
    const u32 n0 = D_ < L_ ? D_ : L_;  // the part that does not read its own output
#pragma nounroll
    for (u32 k = (u32)lane; k < n0; k += 64) {
      const i32 si = S_ + (i32)k;
      // (values are selected, not pointers: an LDS / global pointer select trips the gfx950 backend)
      const u32 lv = ob[si >= 0 ? si : 0];
      u32 gv = 0;
      const i64 ab = (i64)wpos + si;   // ... unless it lies in front of the chunk (symbols only): a marker
      if (si < 0 && (!MARK || ab >= 0)) gv = gsrc[si];
      if (MARK && ab < 0) gv = (u32)(SYM_MARK + 32768 + ab);
      ob[fd + k] = (E)(si >= 0 ? lv : gv);
    }
    wave_sync();
    u32 filled = n0;  // a multiple of D_ from here on: the destination repeats with period D_
    while (filled < L_) {
      const u32 n = filled < L_ - filled ? filled : L_ - filled;
#pragma nounroll
      for (u32 k = (u32)lane; k < n; k += 64) ob[fd + filled + k] = ob[fd + k];
      wave_sync();
      filled += n;
    }
  };
  auto store_match = [&](E *dp, u32 len, u64 w0, u64 w1, u64 w2, u64 w3) {
    if constexpr (MARK) {
      deposit16_sym(dp, len, w0, w1, w2, w3);  // len <= 16 symbols: one 32-byte piece
    } else {  // len <= 32: two 16-byte pieces, the second ends at len
      if (len > 16) {
        ((unaligned_u64 *)dp)->v = w0;
        ((unaligned_u64 *)(dp + 8))->v = w1;
        ((unaligned_u64 *)(dp + len - 16))->v = w2;
        ((unaligned_u64 *)(dp + len - 8))->v = w3;
      }
      if (len <= 16) deposit16(dp, len, w0, w1);
    }
  };
  // 16 (bytes: 16 + the last 16) source elements of a match whose source is flushed output at window index so (< 0);
  // symbols in front of the chunk are made up: markers count upwards like the bytes they stand for
  auto fetch = [&](i32 so, u32 len, u64 &w0, u64 &w1, u64 &w2, u64 &w3) {
#ifdef AHIP_ABL_NO_FETCH  // dev ablation (wrong bytes): what the global source fetches cost
    w0 = w1 = w2 = w3 = (u64)so + len; return;
#endif
    if constexpr (MARK) {
      const i64 ab = (i64)wpos + so;
      if (ab < 0) {
        const u64 m = (u64)(SYM_MARK + 32768 + ab) * 0x0001000100010001ull + 0x0003000200010000ull;  // (fields behind `len` may carry: never stored)
        w0 = m; w1 = m + 0x0004000400040004ull; w2 = m + 0x0008000800080008ull; w3 = m + 0x000c000c000c000cull;
      } else {
        const u8 *sp = (const u8 *)(out_base + wpos + so);
        w0 = load_u64_unaligned(sp); w1 = load_u64_unaligned(sp + 8); w2 = load_u64_unaligned(sp + 16); w3 = load_u64_unaligned(sp + 24);
      }
    } else {
#ifdef AHIP_ABL_FETCH_NEAR  // dev ablation (wrong bytes): every source fetch from the member's first 4 KiB -- L2 hits only
      const u8 *sp = (const u8 *)(out_base + ((wpos + so) & 4095));
#else
      const u8 *sp = (const u8 *)(out_base + wpos + so);
#endif
      w0 = load_u64_unaligned(sp);
      w1 = load_u64_unaligned(sp + 8);
      if (len > 16) { w2 = load_u64_unaligned(sp + len - 16); w3 = load_u64_unaligned(sp + len - 8); }
    }
  };
  // "deposit now": the source is flushed output and the pieces do it.  The first piece may read past the source (the
  // deposit ignores what lies behind `len`): with dist >= 16 those elements are still this member's own output --
  // allocated, just not written yet.  Symbols: the source lies either wholly in the chunk or wholly in front of it.
  auto deposit_now = [&](bool isM, u32 len, u32 dist, i32 so) -> bool {
    const i32 span = (!MARK && len > 16) ? (i32)len : 16;
    bool g = isM && len <= SIMPLE_MAX && dist >= len && so + (i32)len <= 0 && (so + span <= 0 || dist >= 16);
    if constexpr (MARK) {
      const i64 ab = (i64)wpos + so;
      g = g && (ab >= 0 || ab + (i64)len <= 0);
    }
    return g;
  };
  auto resolve_pending = [&]() {
#ifdef AHIP_ABL_NO_ROUNDS  // dev ablation (wrong bytes; tools/ablate.py): what the rounds cost
    npend = 0; return;
#endif
    RTICK(r_r0);
    E *const ob = P.obuf + A;
    wave_sync();
    for (u32 b0 = 0; b0 < npend; b0 += 64) {
      RSTAT(7, 1);
      const bool have = b0 + (u32)lane < npend;
      const uint2 e = P.plist[have ? b0 + lane : 0];
      const u32 wo = e.x & 0xffffu, len = e.x >> 16, dist = e.y;
      const i32 so = (i32)wo - (i32)dist;
      const bool simple = have && so >= 0 && len <= SIMPLE_MAX && dist >= len;
      const u32 sa = simple ? (u32)so : 0u;             // (every lane reads somewhere harmless)
      const u32 sa2 = (!MARK && simple && len > 16) ? sa + len - 16 : sa;  // bytes: the piece that ends at len
      const u64 lmask = len >= 32 ? 0xffffffffull : ((1ull << len) - 1);
      bool pl = have;
      for (;;) {  // (every round finishes at least the first pending match -- by the simple path, which nothing can block, or by
                  //  the whole-wave copy -- so 64 rounds at most: no guard whose firing would drop matches silently)
        const u64 pend = __ballot(pl);
        if (!pend) break;
        RSTAT(8, 1);
        // marks over the source, and the source bytes themselves, in one LDS round trip -- by the lanes that still wait
        // only: the resolver's deferred side is bound by LDS cycles (SQ_LDS_IDX_ACTIVE 78 % of the kernel, three fifths of
        // them in these rounds, profiles/r05_experiments.md), a round serves nine lanes on average, and an unaligned
        // 8-byte access costs its cycles per ACTIVE lane
        u32 m0 = ~0u, m1 = ~0u;
        u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#ifndef AHIP_ROUNDS_ALL_LANES
        if (pl && simple)
#endif
        {
          m0 = P.pmap[sa >> 5]; m1 = P.pmap[(sa >> 5) + 1];
          const u8 *sb = (const u8 *)(ob + sa), *sb2 = MARK ? sb + 16 : (const u8 *)(ob + sa2);  // symbols: one piece of 32 bytes
          w0 = ((const unaligned_u64 *)sb)->v; w1 = ((const unaligned_u64 *)(sb + 8))->v;
          w2 = ((const unaligned_u64 *)sb2)->v; w3 = ((const unaligned_u64 *)(sb2 + 8))->v;
        }
        const u64 marks = ((((u64)m1 << 32) | m0) >> (sa & 31)) & lmask;
        const bool act = pl && simple && marks == 0;
        const int f = __builtin_ctzll(pend);
        const u64 am = __ballot(act);
        wave_sync();
        if (!((am >> f) & 1)) {  // the first pending one is not a simple one (a simple first one is never blocked)
          RTICK(r_h0);
          const u32 fd = lane_bcast(wo, f), L_ = lane_bcast(len, f), D_ = lane_bcast(dist, f);
          RSTAT(9, 1);
          wave_copy(fd, L_, D_);
          if (lane == f) mark(wo, len, false);
          pl = pl && lane != f;
          RTICK(r_h1);
          RACC(5, r_h0, r_h1);
        }
        if (act) {
          store_match(ob + wo, len, w0, w1, w2, w3);
          mark(wo, len, false);
        }
        pl = pl && !act;
        wave_sync();  // this round's bytes and marks are final for the next one
      }
    }
    npend = 0;
    RTICK(r_r1);
    RACC(4, r_r0, r_r1);
  };
  auto flush = [&]() {  // deferred matches, then window -> HBM: byte head up to 16-byte alignment, 16-byte body, byte tail
    RSTAT(6, 1);
    if (npend) resolve_pending();
    RTICK(r_f0);
    wave_sync();
    E *g = out_base + wpos;
    u32 head = (EPV - A) & (EPV - 1);
    if (head > wfill) head = wfill;
#ifndef AHIP_ABL_NO_FLUSH  // dev ablation (wrong bytes): what writing the window out costs
    if ((u32)lane < head) g[lane] = P.obuf[A + lane];
    const u32 body = (wfill - head) & ~(EPV - 1);
    const uint4 *src = (const uint4 *)(P.obuf + A + head);
    uint4 *dst = (uint4 *)(g + head);
    for (u32 i = lane; i < body / EPV; i += 64) dst[i] = src[i];
    const u32 tail0 = head + body;
    if (tail0 + lane < wfill) g[tail0 + lane] = P.obuf[A + tail0 + lane];
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // later matches read this output back
    wave_sync();
    wpos += wfill;
    A = (A + wfill) & (EPV - 1);
    wfill = 0;
    RTICK(r_f1);
    RACC(6, r_f0, r_f1);
  };
  struct Tok { u32 t; i32 base; bool first, inb; u32 nin; };  // a chunk's tokens as loaded: word, run offset (rel. borg); nin = lanes in the look
  struct Ck {  // ... decoded
    u32 t, len; i32 ob;  // ob = output offset rel. borg
    bool inb, pre, gc;   // pre: the source bytes are in w0..w3; gc: classified "deposit now" when prepared
    u64 w0, w1, w2, w3;
    u32 wrel0; i32 cend; u64 inbm;  // (uniform) the window position it was classified against; end of the chunk's output rel. borg; ballot(inb)
  };
  u32 de = 0;
  bool all_done = true;
  while (de < ndir) {
    RTICK(r_l0);
    const u32 ei = de + (u32)lane;
    const bool have = ei < ndir;
    const DirEnt dv = have ? dir[ei] : make_uint4(0u, 0u, 0u, 0u);
    const u64 special = __ballot(have && (dv.y & (DF_BIG | DF_STORED)) != 0);
    const u32 nleft = ndir - de < 64u ? ndir - de : 64u;
    const u32 nplain = special ? (u32)__builtin_ctzll(special) : nleft;  // ordinary runs in front of the first special entry
    const u64 borg = wpos;
    u32 wrel = 0;
    RSTAT(0, 1);
    // ---- one pass over a chunk, straight-line: the lanes of `rem` that fit the window (and its pending list), a prefix;
    //      returns the rest (the caller flushes and comes back) ----
    auto pass = [&](Ck &c, u64 rem) -> u64 {
      RTICK(r_p0);
      const bool lit = (c.t & REC_LIT) != 0;
      const u32 dist = (c.t & 0x7fffu) + 1u;
      const bool isM = c.inb && !lit;
      const i32 wo = c.ob - (i32)wrel;  // window index of the destination
      const i32 so = wo - (i32)dist;    // window index of the source (negative: flushed output)
      // A match is deposited right here when deposit_now() says so; everything else is deferred to the flush.
      bool Gc = c.gc;  // as classified when the chunk was prepared ...
      RSTAT(2, 1);
      if (wrel != c.wrel0) { RSTAT(10, 1); Gc = deposit_now(isM, c.len, dist, so); }  // ... unless the window has moved since
      const bool defer_ = isM && !Gc;
      bool fit = c.inb;
      const bool whole = rem == c.inbm && (u32)(c.cend - (i32)wrel) <= WIN_CAP && npend + 64 <= PEND_CAP;
      if (!whole) {  // the window or its pending list ends inside the chunk: what fits has to be a prefix of what remains
        RSTAT(3, 1);
        fit = ((rem >> lane) & 1) && (u32)wo + c.len <= WIN_CAP;
        const u64 dm = __ballot(fit && defer_);
        const u32 before = (u32)__popcll(dm & ((1ull << lane) - 1));
        fit = fit && npend + before + (defer_ ? 1u : 0u) <= PEND_CAP;
        const u64 fm0 = __ballot(fit), low = rem & (0 - rem);
        const u64 run = (fm0 & low) ? fm0 & ~(fm0 + low) : 0ull;  // the lanes from the first remaining one up to the first that does not fit
        fit = (run >> lane) & 1;
      }
      const u64 fm = __ballot(fit);
      if (!fm) return rem;
      E *const ob = P.obuf + A;
      if (fit && lit) ob[wo] = (E)(u8)c.t;
      const bool G = fit && Gc;
      const bool D = fit && defer_;
      // flushed sources that were not fetched ahead (the window was flushed after the chunk was prepared): fetched
      // now, and waited for inside this branch, so that nothing else waits for the loads of the NEXT chunk
      const bool late = G && !c.pre;  // (only after a flush: otherwise `pre` is exactly the classification)
      if (wrel != c.wrel0 && __any(late)) {
        RSTAT(4, 1);
        if (late) {
          fetch(so, c.len, c.w0, c.w1, c.w2, c.w3);
          c.pre = true;
        }
        AHIP_PIN(c.w0); AHIP_PIN(c.w1); AHIP_PIN(c.w2); AHIP_PIN(c.w3);
      }
#ifndef AHIP_ABL_NO_DEPOSIT  // dev ablation (wrong bytes): what the deposits of fetched matches cost
      if (G) store_match(ob + wo, c.len, c.w0, c.w1, c.w2, c.w3);
#endif
      const u64 dm = __ballot(D);
#ifdef AHIP_RES_STATS
      { const u64 gm_ = __ballot(G), lm_ = __ballot(fit && lit); RSTAT(5, __popcll(dm)); RSTAT(11, __popcll(gm_)); RSTAT(12, __popcll(lm_)); }
#endif
#ifdef AHIP_ABL_NO_DEFER  // dev ablation (wrong bytes): what the deferral bookkeeping (list + marks) and the rounds cost together
      if (false) {
#else
      if (dm) {
#endif
        if (D) {
          P.plist[npend + (u32)__popcll(dm & ((1ull << lane) - 1))] = make_uint2((u32)wo | (c.len << 16), dist);
          mark((u32)wo, c.len, true);
        }
        npend += (u32)__popcll(dm);
      }
      const int last = 63 - __builtin_clzll(fm);
      wfill = lane_bcast((u32)wo + c.len, last);
      RTICK(r_p1);
      RACC(3, r_p0, r_p1);
      return rem & ~fm;
    };
    // a whole chunk: passes and flushes (ONE flush site per loop: the flush is a lot of code and registers)
    auto process = [&](Ck &c) {
      RSTAT(1, 1);
      u64 rem = c.inbm;
      // (a pass after a flush always takes at least one token -- the window starts at its first byte then and a token is at
      //  most 258 bytes --, so 64 passes and as many flushes are the most a chunk can need; the bound only keeps a broken
      //  invariant from spinning, and reaching it is reported, not swallowed)
      for (u32 guard = 0; guard < 300; ++guard) {
        if (rem) rem = pass(c, rem);
        if (!rem && wfill < WIN_FLUSH) break;
        flush();
        wrel = (u32)(wpos - borg);
        // sources that were fetched ahead stay valid (flushed output never changes); everything else is looked at again
      }
      if (rem) all_done = false;
    };
    if (nplain == 0) {
      // ---- a special entry at the head of the look ----
      const u32 y = lane_bcast(dv.y, 0), x = lane_bcast(dv.x, 0);
      const u32 cnt = y & DF_CNT;
      if (y & DF_STORED) {  // stored block: input -> output copy, past the window
        if (wfill) flush();
        const u64 src = (u64)uniform(area[x]) | ((u64)uniform(area[x + 1]) << 32);
        E *g = out_base + wpos;
        for (u32 i = lane; i < cnt; i += 64) g[i] = (E)in[src + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        wpos += cnt;
        A = (A + cnt) & (EPV - 1);
      } else {
        // a run whose `end` fields may wrap: lengths are exact modulo 2^16, offsets from a prefix sum
        u32 carry_end = 0;
        u32 run = wfill;  // offset (rel. borg) of the next chunk's first byte
        for (u32 c0 = 0; c0 < cnt; c0 += 64) {
          Ck c;
          c.inb = c0 + (u32)lane < cnt;
          c.t = c.inb ? area[x + c0 + lane] : 0u;
          const u32 end = c.t >> 16;
          u32 pe = lane_prev(end);
          pe = lane == 0 ? carry_end : pe;
          c.len = c.inb ? ((end - pe) & 0xffffu) : 0u;
          u32 tot;
          c.ob = (i32)(run + wave_excl_sum(c.len, tot));
          c.pre = false;
          c.gc = false;
          c.wrel0 = ~0u;  // (classified in pass())
          c.cend = (i32)(run + tot);
          c.inbm = __ballot(c.inb);
          c.w0 = c.w1 = c.w2 = c.w3 = 0;
          carry_end = lane_bcast(end, 63);
          // (a chunk of long matches may be larger than the window: process() splits it)
          process(c);
          run += tot;
        }
      }
      de += 1;
      continue;
    }
    // ---- ordinary runs: lane r holds run r of the look (as many as LOOK_TOK tokens allow, at least one) ----
    u32 rcnt = (u32)lane < nplain ? (dv.y & DF_CNT) : 0u;
    u32 total;
    const u32 ts = wave_excl_sum(rcnt, total);                                   // first token of the run within the look
    const u32 nlook = (u32)__popcll(__ballot((u32)lane < nplain && ts + rcnt <= LOOK_TOK));  // (a prefix: ts grows with the lane)
    const bool rmine = (u32)lane < nlook;
    total = nlook < 64 ? lane_bcast(ts, (int)nlook) : total;
    const i32 rb = (i32)((((u64)dv.w << 32) | dv.z) - borg);                     // where its output starts, rel. borg
    // token -> run without a search: a bit per token of the look marks the first token of every run, so the run of token i
    // is the number of marks in [0, i] - 1; what a lane needs of its run comes from a table
    for (u32 i = lane; i < LOOK_TOK / 32 + 4; i += 64) P.rbits[i] = 0;
    wave_sync();
    if (rmine) {
      P.rtab[lane] = make_uint2(dv.x - ts, (u32)rb);
      atomicOr(&P.rbits[ts >> 5], 1u << (ts & 31));
    }
    wave_sync();
    u32 gruns = 0;  // runs that begin in front of the next gathered chunk
    u32 gb0 = P.rbits[0], gb1 = P.rbits[1];  // the marks of the next gathered chunk, read one chunk ahead
    RTICK(r_l1);
    RACC(0, r_l0, r_l1);
    auto gather = [&](u32 c0) -> Tok {  // (called with c0 = 0, 64, 128, ... in this order)
      RTICK(r_g0);
      Tok q;
      const u32 idx = c0 + (u32)lane;
      q.inb = idx < total;
      q.nin = total - c0 < 64u ? total - c0 : 64u;
      const u64 bits = (u64)uniform(gb0) | ((u64)uniform(gb1) << 32);
      gb0 = P.rbits[(c0 >> 5) + 2];  // (LOOK_TOK / 32 + 2 words: the read past the last chunk stays inside)
      gb1 = P.rbits[(c0 >> 5) + 3];
      const u32 r = gruns + (u32)__popcll(bits & ((2ull << lane) - 1)) - 1u;  // (token 0 of the look is marked: never negative)
      gruns += (u32)__popcll(bits);
      q.first = (bits >> lane) & 1;
      const uint2 e = P.rtab[r & 63u];
      q.base = (i32)e.y;
      q.t = area[q.inb ? e.x + idx : 0u];  // (no branch around the load and no use of it here: it stays in flight; lanes
                                           //  outside the look read some token, which `inb` keeps anyone from using)
      RTICK(r_g1);
      RACC(1, r_g0, r_g1);
      return q;
    };
    u32 carry_end = 0;  // `end` of the token in front of the chunk being prepared
    auto prep = [&](const Tok &q) -> Ck {
      RTICK(r_q0);
      Ck c;
      c.t = q.t;
      c.inb = q.inb;
      const u32 end = q.t >> 16;
      u32 pe = lane_prev(end);
      pe = lane == 0 ? carry_end : pe;
      pe = q.first ? 0u : pe;
      carry_end = lane_bcast(end, 63);
      c.len = q.inb ? end - pe : 0u;
      c.ob = q.base + (i32)pe;
      const bool lit = (q.t & REC_LIT) != 0;
      const u32 dist = (q.t & 0x7fffu) + 1u;
      const i32 so = c.ob - (i32)wrel - (i32)dist;  // against the window as it is NOW: what is flushed stays flushed
      c.gc = deposit_now(q.inb && !lit, c.len, dist, so);
      c.pre = c.gc;
      c.wrel0 = wrel;
      c.inbm = q.nin >= 64 ? ~0ull : (1ull << q.nin) - 1;
      c.cend = (i32)lane_bcast((u32)c.ob + c.len, (int)q.nin - 1);
      c.w0 = c.w1 = c.w2 = c.w3 = 0;  // (also ends the live range of the previous chunk's registers)
      if (c.pre) fetch(so, c.len, c.w0, c.w1, c.w2, c.w3);
      RTICK(r_q1);
      RACC(2, r_q0, r_q1);
      return c;
    };
    Tok t1 = gather(0);
    Ck c1 = prep(t1);
    Tok t2 = t1;
    if (64 < total) t2 = gather(64);
    for (u32 c0 = 0; c0 < total; c0 += 64) {
      Ck c = c1;
      if (c0 + 64 < total) c1 = prep(t2);          // decode + fetch the flushed sources of the next chunk
      if (c0 + 128 < total) t2 = gather(c0 + 128);  // load the tokens of the one after
      process(c);
    }
    de += nlook;
  }
  if (wfill) flush();
  RTICK(r_end);
  RACC(7, r_begin, r_end);
  return all_done;
}

}  // namespace ahip
