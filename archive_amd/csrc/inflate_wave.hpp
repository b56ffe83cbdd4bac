// inflate_wave.hpp -- one wave64 inflates one DEFLATE stream ("member").
//
// Reproduces, bit for bit on valid streams and on the documented malformed cases, the
// reference's pure-Dart decoder (all paths relative to /root/reference/lib/src):
//   codecs/zlib/inflate.dart:104-156   _inflate/_parseBlock          -> inflate_member()
//   codecs/zlib/inflate.dart:159-211   _readBits/_readCodeByTable    -> BitCursor + table lookup
//   codecs/zlib/inflate.dart:213-234   _parseUncompressedBlock       -> stored_block()
//   codecs/zlib/inflate.dart:239-298   _parseDynamicHuffmanBlock     -> dynamic_header()
//   codecs/zlib/inflate.dart:300-343   _decodeHuffman                -> huffman_block()
//   codecs/zlib/inflate.dart:345-401   _decode                       -> dynamic_header()
//   codecs/zlib/_huffman_table.dart:9-46  HuffmanTable               -> build_decode_table()
//   util/output_memory_stream.dart:79-98  writeBackReference         -> lz_copy()
//
// The reference keeps one 2^maxCodeLength-entry table per code (up to 128 KiB); here each code
// is a root-bits primary table in LDS plus a canonical (first-code / count / sorted-symbol)
// search for the rare longer codes.  Both give the same (symbol, length) for every bit pattern
// of a non-over-subscribed code, including the "unfilled entry = symbol 0, length 0" behaviour
// of incomplete codes.
//
// Stream position model: the reference's byte-at-a-time accumulator is replaced by an absolute
// bit cursor.  _readBits(n) fails iff cursor+n > 8*len; _readCodeByTable fails iff
// cursor+maxCodeLength > 8*len (quirk q2); at block ends the reference un-reads whole bytes, so
// its InputStream position is ceil(cursor/8).
#pragma once
#include "common.hpp"

namespace ahip {

// ---- decode-table entry ----
//  bits 0-3  code length, bits 4-7 extra-bit count, bits 8-12 flags, bits 16-31 value
constexpr u32 E_LIT = 0x100;   // value = 0x8000 | literal byte
constexpr u32 E_EOB = 0x200;   // end of block (symbol 256)
constexpr u32 E_BAD = 0x400;   // litlen 286/287, distance 30/31: reference returns -1
constexpr u32 E_LONG = 0x800;  // code longer than the primary table: bits 0-3 = index bits of its second-level table, value = that table's base
constexpr u32 E_HOLE = 0x1000; // unfilled litlen entry (symbol 0, length 0): literal-0 forever

#ifndef AHIP_LL_ROOT
#define AHIP_LL_ROOT 9
#endif
#ifndef AHIP_D_ROOT
#define AHIP_D_ROOT 7
#endif
constexpr int LL_ROOT = AHIP_LL_ROOT;  // primary table bits: litlen / distance
constexpr int D_ROOT = AHIP_D_ROOT;

// A literal's value is 0x8000 | byte: shifted up by 16 it IS the literal token (TK_LIT | byte << 16).
AHIP_DEVINL u32 litlen_entry(u32 sym, u32 len) {
  if (sym < 256) return ((0x8000u | sym) << 16) | E_LIT | len;
  if (sym == 256) return E_EOB | len;
  if (sym > 285) return E_BAD | len;
  u32 i = sym - 257;
  u32 xb = (i < 8 || i == 28) ? 0u : ((i - 4) >> 2);
  u32 base = (i < 8) ? (3 + i) : (i == 28 ? 258u : (3 + ((4 + (i & 3)) << xb)));
  return (base << 16) | (xb << 4) | len;
}
AHIP_DEVINL u32 dist_entry(u32 sym, u32 len) {
  if (sym > 29) return E_BAD | len;
  u32 xb = (sym < 4) ? 0u : ((sym - 2) >> 1);
  u32 base = (sym < 4) ? (sym + 1) : (1 + ((2 + (sym & 1)) << xb));
  return (base << 16) | (xb << 4) | len;
}

// What the decoders keep of a code next to its tables.
struct CodeDesc {
  u32 maxlen;      // reference HuffmanTable.maxCodeLength
  u32 pad;
};
// (The first canonical code of each length is needed by the table build alone, indexed by a lane's own code length: it
//  lives in the header scratch -- HeaderLds::cl, dead once the code lengths are decoded -- not in the tables' LDS, of which
//  every byte counts: 64 bytes here are what lets 80 items of bitstream ring fit nine LDS granules, inflate_par.hpp.)

// Second-level tables (codes longer than the primary index), one per primary prefix that long codes start with,
// indexed by the next (longest code under the prefix - root) stream bits.  Sized for every COMPLETE code over the
// alphabet (a dynamic-programming sweep over the canonical count vectors gives 340 entries for 286 litlen symbols at
// root 9 and 274 for 32 distance symbols at root 7); a code set that needs more -- an incomplete or padded one no
// encoder emits -- is decoded through the reference's own table like the over-subscribed sets (MS_OVERSUB route).
#ifndef AHIP_LL_SUB
#define AHIP_LL_SUB 352
#endif
#ifndef AHIP_D_SUB
#define AHIP_D_SUB 280
#endif
constexpr u32 LL_SUB = AHIP_LL_SUB, D_SUB = AHIP_D_SUB;

// Block-header scratch: only alive between a block's 3-bit header and the end of its table build,
// so the tokenizer overlays it on the (then idle) bitstream window.
struct HeaderLds {
  u32 cl[128];        // code-length code: single level, built exactly like the reference
  u8 lens[320 + 8];
};
struct WaveLds {
  u32 ll[1 << LL_ROOT];
  u32 dt[1 << D_ROOT];
  CodeDesc lld, dd;
  u32 ll_sub[LL_SUB];
  u32 d_sub[D_SUB];
};

struct BitCursor {
  const u8 *in;
  u64 in_len;
  u64 total_bits;
  u64 pos;  // absolute bit index
  // optional LDS copy of in[stage_byte .. stage_byte + stage_len): block headers are decoded
  // symbol by symbol, and every global read would cost an HBM/L2 round trip
  const u32 *stage;
  u64 stage_byte;
  u32 stage_len;
  // The reference pulls whole bytes into its accumulator and leaves them there when a block fails half way
  // (inflate.dart:159-211: no un-read on `return -1`), so its stream position after such a failure is
  // (pos + blen) / 8 with blen = the bits its accumulator would still hold.  Tracked by the serial readers only
  // (block headers; replay_to_failure() for a Huffman block).
  u32 blen;
};
AHIP_DEVINL void acc_need(BitCursor &b, u32 n) { while (b.blen < n) b.blen += 8; }

// >= 57 valid bits starting at the cursor; bytes past the end read as zero.
AHIP_DEVINL u64 peek_bits(const BitCursor &b) {
  u64 byte = b.pos >> 3;
  u32 sh = (u32)b.pos & 7;
  u64 w;
  if (b.stage && byte >= b.stage_byte && byte + 12 <= b.stage_byte + b.stage_len) {
    const u32 rel = (u32)(byte - b.stage_byte);  // stage_byte is 4-aligned relative to `in`
    const u32 *p = b.stage + (rel >> 2);
    const u32 w0 = p[0], w1 = p[1], w2 = p[2];
    const u32 bsh = (rel & 3) * 8;
    const u32 lo = __builtin_amdgcn_alignbit(w1, w0, bsh), hi = __builtin_amdgcn_alignbit(w2, w1, bsh);
    return (((u64)hi << 32) | lo) >> sh;
  }
  if (byte + 8 <= b.in_len) {
    w = load_u64_unaligned(b.in + byte);
  } else {
    w = 0;
    for (int k = 0; k < 8; ++k)
      if (byte + k < b.in_len) w |= (u64)b.in[byte + k] << (8 * k);
  }
  return w >> sh;
}
// _readBits: -1 when fewer than n bits are left, 0 for n == 0
AHIP_DEVINL int read_bits(BitCursor &b, u32 n) {
  if (n == 0) return 0;
  if (b.pos + n > b.total_bits) return -1;
  u32 v = (u32)peek_bits(b) & ((1u << n) - 1);
  acc_need(b, n);
  b.blen -= n;
  b.pos += n;
  return (int)v;
}

// Second-level lookup: `e` is the primary entry (E_LONG set), `bits` the stream bits the primary index came from.
AHIP_DEVINL u32 long_lookup(const u32 *sub, u32 e, u32 bits, int root) {
  return sub[(e >> 16) + __builtin_amdgcn_ubfe(bits, (u32)root, e & 15)];
}

// Build primary + second-level tables from `n` code lengths in LDS (wave-cooperative).
// Returns false for an over-subscribed set, or one whose second-level tables do not fit (neither reproduced here).
template <bool IS_DIST>
AHIP_DEVINL bool build_decode_table(const u8 *lens, int n, u32 *primary, int root, CodeDesc &cd, u32 *sub, u32 sub_cap,
                                    int lane, u16 *first_lds) {
  constexpr int CHUNKS = IS_DIST ? 1 : 5;
  u32 mylen[CHUNKS], myrank[CHUNKS];
  u32 cnt[16];
#pragma unroll
  for (int L = 0; L < 16; ++L) cnt[L] = 0;
  const u64 lt_mask = (1ull << lane) - 1;
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    int s = c * 64 + lane;
    u32 l = (s < n) ? lens[s] : 0u;
    mylen[c] = l;
    u32 r = 0;
#pragma unroll
    for (int L = 1; L < 16; ++L) {
      u64 m = __ballot(l == (u32)L);
      if (l == (u32)L) r = cnt[L] + __popcll(m & lt_mask);
      cnt[L] += __popcll(m);
    }
    myrank[c] = r;
  }
  // canonical first codes (uniform)
  u32 code = 0, maxlen = 0;
  bool over = false;
  u32 first[16];
  first[0] = 0;
#pragma unroll
  for (int L = 1; L < 16; ++L) {
    first[L] = code;
    if (cnt[L]) maxlen = L;
    if (code + cnt[L] > (1u << L)) over = true;
    code = (code + cnt[L]) << 1;
  }
  if (lane < 16) {
    u32 f = 0;
#pragma unroll
    for (int L = 0; L < 16; ++L)
      if (lane == L) f = first[L];
    first_lds[lane] = (u16)f;
  }
  if (lane == 0) cd.maxlen = maxlen;
  // default fill: unfilled entries decode as symbol 0 with length 0
  const u32 hole = IS_DIST ? dist_entry(0, 0) : (u32)E_HOLE;
  for (int i = lane; i < (1 << root); i += 64) primary[i] = hole;
  const bool has_long = (int)maxlen > root && !over;
  if (has_long)
    for (u32 i = lane; i < sub_cap; i += 64) sub[i] = hole;
  wave_sync();
  // One second-level table per primary prefix q (canonical bit order) that long codes start with, as wide as the
  // longest code under q: lengths never decrease along the canonical order, so that is the LAST code under q.
  bool fits = true;
  if (has_long) {
    u32 q0 = 1u << root;  // first prefix with a long code
#pragma unroll
    for (int L = 15; L >= 1; --L)
      if (L > root && cnt[L]) q0 = first[L] >> (L - root);
    u32 used = 0;
    for (u32 qb = q0; qb < (1u << root); qb += 64) {
      const u32 q = qb + (u32)lane;
      u32 sb = 0;
#pragma unroll
      for (int L = 1; L < 16; ++L) {
        if (L > root && cnt[L]) {
          const u32 lo = first[L] >> (L - root), hi = (first[L] + cnt[L] - 1) >> (L - root);
          sb = (q >= lo && q <= hi) ? (u32)(L - root) : sb;
        }
      }
      if (q >= (1u << root)) sb = 0;
      const u32 size = sb ? 1u << sb : 0u;
      u32 tot;
      const u32 base = used + wave_excl_sum(size, tot);
      if (sb && base + size <= sub_cap) primary[__brev(q) >> (32 - root)] = E_LONG | sb | (base << 16);
      used += tot;
    }
    fits = used <= sub_cap;
  }
  wave_sync();
  // symbols -> primary / second-level entries
  // (the entry of symbol c * 64 + lane depends on the lane alone, and the compiler would compute all of them once per
  //  kernel and keep them in a dozen VGPRs across the whole decode -- the registers the flow decoder is short of:
  //  the pin makes them local to this table build)
  u32 ln = (u32)lane;
  AHIP_PIN(ln);
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    u32 l = mylen[c];
    if (l) {
      u32 s = c * 64 + ln;
      u32 cde = first_lds[l] + myrank[c];
      const u32 e = IS_DIST ? dist_entry(s, l) : litlen_entry(s, l);
      if ((int)l <= root) {
        u32 rev = __brev(cde) >> (32 - l);
        for (u32 j = rev; j < (1u << root); j += (1u << l)) primary[j] = e;
      } else if (has_long && fits) {
        const u32 xl = l - root;  // bits below the prefix
        const u32 pe = primary[__brev(cde >> xl) >> (32 - root)];
        const u32 sb = pe & 15, base = pe >> 16;
        const u32 j0 = __brev(cde & ((1u << xl) - 1)) >> (32 - xl);
        for (u32 j = j0; j < (1u << sb); j += (1u << xl)) sub[base + j] = e;
      }
    }
  }
  wave_sync();
  return !over && fits;
}

// Fixed-Huffman code lengths (inflate.dart:408-735): 144x8, 112x9, 24x7, 8x8; 30 distance codes of 5.
AHIP_DEVINL void fixed_lengths(u8 *lens, int lane) {
  for (int s = lane; s < 288; s += 64) lens[s] = (s < 144) ? 8 : (s < 256) ? 9 : (s < 280) ? 7 : 8;
  if (lane < 30) lens[288 + lane] = 5;
  wave_sync();
}

struct OutCursor {
  u8 *base;   // address of output position 0 (= the stream's own first byte minus `hist`)
  u64 pos;    // bytes produced + hist
  u64 limit;  // window size + hist
  u64 hist;   // bytes of earlier output in front of this stream that a back-reference may reach (quirk q8)
  u32 far;    // some back-reference did reach into them
  u64 org;    // `pos` of the stream's own first byte (what the token store counts output offsets from)
};

// writeBackReference(distance, count): every lane copies bytes lane, lane+64, ...
// All sources lie in [pos-dist, pos), i.e. bytes that existed before this call.
template <bool WRITE>
AHIP_DEVINL void lz_copy(OutCursor &o, u32 dist, u32 len, int lane) {
  if (WRITE) {
    u8 *dst = o.base + o.pos;
    const u8 *src = dst - dist;
    if (dist >= len) {
      for (u32 i = lane; i < len; i += 64) dst[i] = src[i];
    } else {
      for (u32 i = lane; i < len; i += 64) dst[i] = src[i % dist];
    }
  }
  o.pos += len;
}

// One literal/length(+distance) token.  Returns 0 = continue, 1 = end of block, else MS_* + 100.
template <bool WRITE, bool CAREFUL>
AHIP_DEVINL u32 huffman_token(WaveLds &L, BitCursor &b, OutCursor &o, u32 ll_max, u32 d_max, int lane) {
  if (CAREFUL && b.pos + ll_max > b.total_bits) return 100 + MS_FALSE_EOS;
  u64 w = peek_bits(b);
  u32 e = uniform(L.ll[(u32)w & ((1u << LL_ROOT) - 1)]);
  if (e & E_LONG) e = uniform(long_lookup(L.ll_sub, e, (u32)w, LL_ROOT));
  u32 cl = e & 15;
  if (e & (E_LIT | E_EOB | E_BAD | E_HOLE)) {
    if (e & E_LIT) {
      if (o.pos >= o.limit) return 100 + MS_CAP;
      if (WRITE && lane == 0) o.base[o.pos] = (u8)(e >> 16);
      o.pos += 1;
      b.pos += cl;
      return 0;
    }
    if (e & E_EOB) { b.pos += cl; return 1; }
    if (e & E_BAD) return 100 + MS_FALSE;
    return 100 + MS_HANG;
  }
  // length symbol
  u32 used = cl;
  w >>= cl;
  u32 xb = (e >> 4) & 15;
  i32 len = (i32)(e >> 16);
  if (CAREFUL && xb && b.pos + used + xb > b.total_bits) {
    len -= 1;  // _readBits returned -1 and the reference adds it (inflate.dart:322)
  } else {
    len += (i32)((u32)w & ((1u << xb) - 1));
    w >>= xb;
    used += xb;
  }
  if (CAREFUL && b.pos + used + d_max > b.total_bits) return 100 + MS_FALSE_EOS;
  u32 d = uniform(L.dt[(u32)w & ((1u << D_ROOT) - 1)]);
  if (d & E_LONG) d = uniform(long_lookup(L.d_sub, d, (u32)w, D_ROOT));
  if (d & E_BAD) return 100 + MS_FALSE;
  u32 dl = d & 15;
  w >>= dl;
  used += dl;
  u32 dxb = (d >> 4) & 15;
  i32 dist = (i32)(d >> 16);
  if (CAREFUL && dxb && b.pos + used + dxb > b.total_bits) {
    dist -= 1;
  } else {
    dist += (i32)((u32)w & ((1u << dxb) - 1));
    used += dxb;
  }
  b.pos += used;
  if ((u64)dist > o.pos) return 100 + MS_FARREF;
  if (o.pos + (u64)len > o.limit) return 100 + MS_CAP;
  if ((u64)dist > o.pos - o.hist) o.far = 1;
  lz_copy<WRITE>(o, (u32)dist, (u32)len, lane);
  return 0;
}

template <bool WRITE>
AHIP_DEVINL u32 huffman_block(WaveLds &L, BitCursor &b, OutCursor &o, int lane) {
  const u32 ll_max = L.lld.maxlen, d_max = L.dd.maxlen;
  for (;;) {
    u32 r;
    // 16 readable bytes ahead: every EOS test of the careful path is trivially false
    if ((b.pos >> 3) + 16 <= b.in_len) r = huffman_token<WRITE, false>(L, b, o, ll_max, d_max, lane);
    else r = huffman_token<WRITE, true>(L, b, o, ll_max, d_max, lane);
    if (r == 0) continue;
    if (r == 1) return MS_OK;
    return r - 100;
  }
}

// The same header decode with everything on the scalar side: the (<= 562-byte) header sits in three VGPRs (lane l holds
// dwords l, 64 + l, 128 + l of the 640 bytes at `hw`), the bit buffer and the run-length state machine live in SGPRs, the
// 128-entry code-length table in two VGPRs read with v_readlane -- no LDS round trip per symbol (the LDS-table version
// spends ~550 cycles per code length on two of them).  Same results, same failure verdicts, same accumulator tracking;
// only usable when the 640 bytes lie inside the input (then the input cannot end inside the header).
AHIP_DEVINL u32 dynamic_header_fast(HeaderLds &L, BitCursor &b, const u8 *hw, u64 hw_byte, int lane, int &hlit_out, int &hdist_out) {
  const u32 r0 = load_u32_unaligned(hw + 4 * lane), r1 = load_u32_unaligned(hw + 256 + 4 * lane), r2 = lane < 32 ? load_u32_unaligned(hw + 512 + 4 * lane) : 0u;
  auto dword = [&](u32 i) -> u32 {  // i uniform, < 160
    const u32 a = lane_bcast(r0, (int)(i & 63)), c = lane_bcast(r1, (int)(i & 63)), d = lane_bcast(r2, (int)(i & 63));
    return i < 64 ? a : (i < 128 ? c : d);
  };
  const u32 bitoff = uniform((u32)(b.pos - hw_byte * 8));  // < 32; provably wave-uniform from here on: SGPRs
  u64 sbuf = ((u64)dword(1) << 32 | dword(0)) >> bitoff;
  u32 scnt = 64 - bitoff, nd = 2, used = 0;
  auto take = [&](u32 n) -> u32 {  // n <= 16; at least 32 valid bits are kept in sbuf
    const u32 v = (u32)sbuf & ((1u << n) - 1);
    sbuf >>= n; scnt -= n; used += n;
    if (scnt <= 32) { sbuf |= (u64)dword(nd) << scnt; scnt += 32; nd += 1; }
    return v;
  };
  // a failure verdict is only a signal: the caller decodes the header again with dynamic_header(), which also keeps
  // the reference's accumulator length (needed for the exact stream position after a failure)
  auto leave = [&](u32 verdict) { if (verdict == MS_OK) { b.pos += used; b.blen = (8 - ((u32)b.pos & 7)) & 7; } return verdict; };
  const int hlit = (int)take(5) + 257;
  if (hlit > 288) return leave(MS_FALSE);
  const int hdist = (int)take(5) + 1;
  if (hdist > 32) return leave(MS_FALSE);
  const int hclen = (int)take(4) + 4;
  if (hclen > 19) return leave(MS_FALSE);
  u32 cl_len[19];
#pragma unroll
  for (int i = 0; i < 19; ++i) cl_len[i] = 0;
  const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
  for (int i = 0; i < 19; ++i)
    if (i < hclen) cl_len[order[i]] = take(3);
  u32 cl_max = 0;
#pragma unroll
  for (int i = 0; i < 19; ++i) cl_max = cl_len[i] > cl_max ? cl_len[i] : cl_max;
  const u32 cl_size = 1u << cl_max;
  for (u32 i = lane; i < 128; i += 64) L.cl[i] = 0;
  wave_sync();
  if (lane == 0) {  // HuffmanTable(codeLengths): single level, the reference's fill order
    u32 code = 0, skip = 2;
    for (u32 bl = 1; bl <= cl_max; ++bl) {
#pragma unroll
      for (int i = 0; i < 19; ++i) {
        if (cl_len[i] == bl) {
          u32 rev = __brev(code) >> (32 - bl);
          for (u32 j = rev; j < cl_size; j += skip) L.cl[j] = (bl << 16) | (u32)i;
          ++code;
        }
      }
      code <<= 1;
      skip <<= 1;
    }
  }
  wave_sync();
  const u32 t0 = L.cl[lane], t1 = L.cl[64 + lane];
  const int num = hlit + hdist;
  int i = 0;
  u32 prev = 0;
  while (i < num) {
    const u32 idx = (u32)sbuf & (cl_size - 1);
    const u32 ea = lane_bcast(t0, (int)(idx & 63)), eb = lane_bcast(t1, (int)(idx & 63));
    const u32 e = idx < 64 ? ea : eb;
    {
      const u32 n = e >> 16;
      sbuf >>= n; scnt -= n; used += n;
      if (scnt <= 32) { sbuf |= (u64)dword(nd) << scnt; scnt += 32; nd += 1; }
    }
    const u32 code = e & 0xffff;
    int repeat;
    u32 fill;
    if (code < 16) { repeat = 1; fill = code; prev = code; }
    else if (code == 16) { repeat = (int)take(2) + 3; fill = prev; }
    else if (code == 17) { repeat = (int)take(3) + 3; fill = 0; prev = 0; }
    else { repeat = (int)take(7) + 11; fill = 0; prev = 0; }
    if (i + repeat > num) return leave(MS_RANGE);  // Dart: index past the end of the Uint8List
    if (repeat == 1) {
      if (lane == 0) L.lens[i] = (u8)fill;
    } else {
      if (lane < repeat) L.lens[i + lane] = (u8)fill;
      if (lane + 64 < repeat) L.lens[i + lane + 64] = (u8)fill;
      if (lane + 128 < repeat) L.lens[i + lane + 128] = (u8)fill;
    }
    i += repeat;
  }
  wave_sync();
  hlit_out = hlit;
  hdist_out = hdist;
  return leave(MS_OK);
}

// A Huffman block ended in a bad litlen / distance symbol (MS_FALSE, not at the end of the input).  Walk the block
// again from its first code, symbol by symbol, keeping the reference's accumulator length: b.pos / b.blen end up
// where the reference's `return -1` leaves them (inflate.dart:300-343).  Only ever runs on corrupt data.
AHIP_DEVINL void replay_to_failure(const WaveLds &L, BitCursor &b) {
  const u32 ll_max = L.lld.maxlen, d_max = L.dd.maxlen;
  for (u32 guard = 0; guard < (1u << 28); ++guard) {
    u64 w = peek_bits(b);
    u32 e = uniform(L.ll[(u32)w & ((1u << LL_ROOT) - 1)]);
    if (e & E_LONG) e = uniform(long_lookup(L.ll_sub, e, (u32)w, LL_ROOT));
    acc_need(b, ll_max);
    const u32 cl = e & 15;
    b.blen -= cl; b.pos += cl;
    if (e & E_LIT) continue;
    if (e & (E_EOB | E_BAD | E_HOLE)) return;
    w >>= cl;
    const u32 xb = (e >> 4) & 15;
    if (xb) { acc_need(b, xb); b.blen -= xb; b.pos += xb; w >>= xb; }
    u32 d = uniform(L.dt[(u32)w & ((1u << D_ROOT) - 1)]);
    if (d & E_LONG) d = uniform(long_lookup(L.d_sub, d, (u32)w, D_ROOT));
    acc_need(b, d_max);
    const u32 dl = d & 15;
    b.blen -= dl; b.pos += dl;
    if (d & E_BAD) return;
    const u32 dxb = (d >> 4) & 15;
    if (dxb) { acc_need(b, dxb); b.blen -= dxb; b.pos += dxb; }
  }
}

// ---- over-subscribed code lengths: the reference's own table ----
// HuffmanTable (_huffman_table.dart:9-46) never checks the Kraft sum: when there are more codes of a length than fit,
// `code` runs past 2^length, its low bits wrap, and later fills overwrite earlier ones.  That cannot be described
// canonically, so the rare member that needs it is decoded (by the late kernel, one wave) with the reference's
// single-level 2^maxCodeLength table, built here in the reference's order -- symbol after symbol; only the strided
// fill of one symbol is spread over the lanes -- in device scratch (up to 2 x 128 KiB).
struct ExactTabs { u32 *ll, *dt; u32 ll_max, d_max; };
AHIP_DEVINL u32 build_exact_table(const u8 *lens, int n, u32 *table, int lane) {
  u32 maxlen = 0;
  for (int i = 0; i < n; ++i) { const u32 l = lens[i]; maxlen = l > maxlen ? l : maxlen; }
  maxlen = uniform(maxlen);
  const u32 size = 1u << maxlen;
  for (u32 j = lane; j < size; j += 64) table[j] = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  u32 code = 0, skip = 2;
  for (u32 bl = 1; bl <= maxlen; ++bl) {
    for (int i = 0; i < n; ++i) {
      if (uniform(lens[i]) != bl) continue;
      const u32 rev = __brev(code) >> (32 - bl);  // the low `bl` bits of code, reversed (higher bits are dropped)
      for (u32 j = rev + (u32)lane * skip; j < size; j += 64 * skip) table[j] = (bl << 16) | (u32)i;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // later symbols overwrite earlier ones: keep the order
      ++code;
    }
    code <<= 1;
    skip <<= 1;
  }
  return maxlen;
}
// table[bits & (size - 1)] as one of our entries (an unfilled slot is symbol 0 with length 0)
template <bool IS_DIST>
AHIP_DEVINL u32 exact_entry(const u32 *table, u32 maxlen, u32 bits) {
  const u32 raw = uniform(table[bits & ((1u << maxlen) - 1)]);
  if (raw == 0) return IS_DIST ? dist_entry(0, 0) : (u32)E_HOLE;
  return IS_DIST ? dist_entry(raw & 0xffff, raw >> 16) : litlen_entry(raw & 0xffff, raw >> 16);
}
// _decodeHuffman with those tables: the careful form of huffman_token, symbol by symbol
template <bool WRITE>
AHIP_DEVINL u32 huffman_block_exact(const ExactTabs &X, BitCursor &b, OutCursor &o, int lane) {
  for (;;) {
    if (b.pos + X.ll_max > b.total_bits) return MS_FALSE_EOS;
    u64 w = peek_bits(b);
    const u32 e = exact_entry<false>(X.ll, X.ll_max, (u32)w);
    const u32 cl = e & 15;
    acc_need(b, X.ll_max);
    if (e & (E_LIT | E_EOB | E_BAD | E_HOLE)) {
      if (e & E_LIT) {
        if (o.pos >= o.limit) return MS_CAP;
        if (WRITE && lane == 0) o.base[o.pos] = (u8)(e >> 16);
        o.pos += 1;
        b.pos += cl; b.blen -= cl;
        continue;
      }
      if (e & E_HOLE) return MS_HANG;
      b.pos += cl; b.blen -= cl;
      return (e & E_EOB) ? MS_OK : MS_FALSE;
    }
    u32 used = cl;
    w >>= cl;
    const u32 xb = (e >> 4) & 15;
    i32 len = (i32)(e >> 16);
    b.blen -= cl;
    if (xb && b.pos + used + xb > b.total_bits) len -= 1;
    else { len += (i32)((u32)w & ((1u << xb) - 1)); w >>= xb; used += xb; if (xb) { acc_need(b, xb); b.blen -= xb; } }
    if (b.pos + used + X.d_max > b.total_bits) return MS_FALSE_EOS;
    const u32 d = exact_entry<true>(X.dt, X.d_max, (u32)w);
    acc_need(b, X.d_max);
    if (d & E_BAD) { b.pos += used + (d & 15); b.blen -= d & 15; return MS_FALSE; }
    const u32 dl = d & 15;
    b.blen -= dl;
    w >>= dl;
    used += dl;
    const u32 dxb = (d >> 4) & 15;
    i32 dist = (i32)(d >> 16);
    if (dxb && b.pos + used + dxb > b.total_bits) dist -= 1;
    else { dist += (i32)((u32)w & ((1u << dxb) - 1)); used += dxb; if (dxb) { acc_need(b, dxb); b.blen -= dxb; } }
    b.pos += used;
    if ((u64)dist > o.pos) return MS_FARREF;
    if (o.pos + (u64)len > o.limit) return MS_CAP;
    if ((u64)dist > o.pos - o.hist) o.far = 1;
    if (WRITE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the copy reads what earlier symbols wrote
    lz_copy<WRITE>(o, (u32)dist, (u32)len, lane);
  }
}

// _parseUncompressedBlock
template <bool WRITE>
AHIP_DEVINL u32 stored_block(BitCursor &b, OutCursor &o, int lane) {
  b.pos = (b.pos + 7) & ~7ull;  // the accumulator is dropped; it never holds a whole byte here
  b.blen = 0;
  int len = read_bits(b, 16);
  int nlen_raw = read_bits(b, 16);
  int nlen = nlen_raw ^ 0xffff;
  if (len != 0 && len != nlen) return (len < 0 || nlen_raw < 0) ? MS_FALSE_EOS : MS_FALSE;
  // LEN == 0 is never compared with its complement (inflate.dart:213-234) -- so an EMPTY stored block whose NLEN is cut
  // short by the end of the input passes; the reference's reader has taken what bytes there were by then, and the block
  // loop ends on isEOS instead of parsing the leftover byte as a block header
  if (nlen_raw < 0) b.pos = b.total_bits;
  u64 byte = b.pos >> 3;
  if ((u64)len > b.in_len - byte) return MS_FALSE;
  if (o.pos + (u64)len > o.limit) return MS_CAP;
  if (WRITE) {
    const u8 *src = b.in + byte;
    u8 *dst = o.base + o.pos;
    for (int i = lane; i < len; i += 64) dst[i] = src[i];
  }
  o.pos += (u64)len;
  b.pos += 8ull * (u64)len;
  return MS_OK;
}

// _parseDynamicHuffmanBlock header + _decode; leaves litlen/dist lengths in L.lens
AHIP_DEVINL u32 dynamic_header(HeaderLds &L, BitCursor &b, int lane, int &hlit_out, int &hdist_out) {
  int hlit = read_bits(b, 5);
  if (hlit < 0) return MS_FALSE_EOS;
  hlit += 257;
  if (hlit > 288) return MS_FALSE;
  int hdist = read_bits(b, 5);
  if (hdist < 0) return MS_FALSE_EOS;
  hdist += 1;
  if (hdist > 32) return MS_FALSE;
  int hclen = read_bits(b, 4);
  if (hclen < 0) return MS_FALSE_EOS;
  hclen += 4;
  if (hclen > 19) return MS_FALSE;
  // code-length code lengths, in the permuted order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
  // (inflate.dart:738-758)
  u32 cl_len[19];
#pragma unroll
  for (int i = 0; i < 19; ++i) cl_len[i] = 0;
  const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
  for (int i = 0; i < 19; ++i) {
    if (i < hclen) {
      int v = read_bits(b, 3);
      if (v < 0) return MS_FALSE_EOS;
      cl_len[order[i]] = (u32)v;
    }
  }
  // HuffmanTable(codeLengths): single level, same fill order as the reference (serial, lane 0)
  u32 cl_max = 0;
#pragma unroll
  for (int i = 0; i < 19; ++i) cl_max = cl_len[i] > cl_max ? cl_len[i] : cl_max;
  const u32 cl_size = 1u << cl_max;
  for (u32 i = lane; i < 128; i += 64) L.cl[i] = 0;
  wave_sync();
  if (lane == 0) {
    u32 code = 0, skip = 2;
    for (u32 bl = 1; bl <= cl_max; ++bl) {
#pragma unroll
      for (int i = 0; i < 19; ++i) {
        if (cl_len[i] == bl) {
          u32 rev = __brev(code) >> (32 - bl);
          for (u32 j = rev; j < cl_size; j += skip) L.cl[j] = (bl << 16) | (u32)i;
          ++code;
        }
      }
      code <<= 1;
      skip <<= 1;
    }
  }
  wave_sync();
  // _decode: run-length coded lengths
  const int num = hlit + hdist;
  int i = 0;
  u32 prev = 0;
  while (i < num) {
    if (b.pos + cl_max > b.total_bits) return MS_FALSE_EOS;
    u32 e = uniform(L.cl[(u32)peek_bits(b) & (cl_size - 1)]);
    acc_need(b, cl_max);
    b.blen -= e >> 16;
    b.pos += e >> 16;
    u32 code = e & 0xffff;
    int repeat;
    u32 fill;
    if (code < 16) {
      repeat = 1; fill = code; prev = code;
    } else if (code == 16) {
      repeat = read_bits(b, 2);
      if (repeat < 0) return MS_FALSE_EOS;
      repeat += 3; fill = prev;
    } else if (code == 17) {
      repeat = read_bits(b, 3);
      if (repeat < 0) return MS_FALSE_EOS;
      repeat += 3; fill = 0; prev = 0;
    } else {
      repeat = read_bits(b, 7);
      if (repeat < 0) return MS_FALSE_EOS;
      repeat += 11; fill = 0; prev = 0;
    }
    if (i + repeat > num) return MS_RANGE;  // Dart: index past the end of the Uint8List
    if (lane < repeat) L.lens[i + lane] = (u8)fill;
    if (lane + 64 < repeat) L.lens[i + lane + 64] = (u8)fill;
    if (lane + 128 < repeat) L.lens[i + lane + 128] = (u8)fill;
    i += repeat;
  }
  wave_sync();
  hlit_out = hlit;
  hdist_out = hdist;
  return MS_OK;
}

}  // namespace ahip
