// bzip2_kernels.hpp -- bzip2 block decoder on gfx950 (first correct version, block-parallel).
//
// Reference: /root/reference/lib/src/codecs/bzip2_decoder.dart (libbzip2's decompress.c in Dart).
// The reference walks blocks one after another; blocks are independent once their bit positions are
// known, so:
//   B0 bz_scan_magic   every bit position of the stream is tested for the 48-bit block magic
//                      0x314159265359 / end-of-stream magic 0x177245385090 (_readBlockType :90-111).
//   B1 bz_decode_block one wave64 per candidate block:
//        phase 1  header, selectors, code lengths, limit/base/perm tables (:114-246, :774-813) and the
//                 Huffman + MTF + RUNA/RUNB loop into tt[] (:267-388) -- serial by nature, executed
//                 wave-uniformly with tables in LDS;
//        phase 2  T^-1 (:406-439) as a 64-lane counting sort: every lane histograms its 1/64 of tt[],
//                 a per-symbol scan across lanes gives each lane its write cursors, then lanes scatter
//                 -- same result as the reference's stable serial loop;
//        phase 3  inverse-BWT pointer chase + run-length undo + MSB-first CRC-32 (:610-727) into the
//                 block's output slab (serial: every step depends on the previous load).
//   The host then follows the chain of blocks (a block's end bit must be the next block's magic),
//   verifies CRCs when asked, and a gather kernel packs the slabs into the caller's buffer.
// The obsolete randomised-block mode is not implemented (BZ_ST_UNSUPPORTED).
#pragma once
#include "common.hpp"

namespace ahip {

constexpr u32 BZ_MAX_SELECTORS = 18002;
constexpr u32 BZ_ST_OK = 0, BZ_ST_FALSE = 1, BZ_ST_RANGE = 2, BZ_ST_OVERFLOW = 16, BZ_ST_UNSUPPORTED = 17;

struct BzCand { u64 bit; u32 kind; u32 pad; };  // kind 0 = compressed block, 2 = end of stream
struct BzResult {
  u64 end_bit;   // bit position just after the block
  u64 out_len;
  u32 status, crc, stored_crc, nblock;
  u32 pad_orig_ptr, pad;  // origPtr kept for the second (direct) un-BWT pass of oversized blocks
};

// ---- B0: magic scan ----
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

struct BzLds {
  i32 limit[6][24], base[6][24];
  u16 perm[6][258];
  u8 len[6][258];
  i32 min_len[6];
  u8 mtf[256], seq2unseq[256];
  u32 unzftab[256];
};

// one wave per candidate block
__global__ __launch_bounds__(64) void bz_decode_block(const u8 *__restrict__ in, u64 n, const BzCand *__restrict__ cands,
                                                      u32 ncand, u32 block_size100k, u32 *__restrict__ tt_all,
                                                      u8 *__restrict__ sel_all, u8 *__restrict__ slabs, u64 slab_cap,
                                                      BzResult *__restrict__ results) {
  __shared__ BzLds L;
  const u32 blk = blockIdx.x, lane = threadIdx.x;
  if (blk >= ncand) return;
  const u32 nblock_max = 100000u * block_size100k;
  u32 *tt = tt_all + (u64)blk * nblock_max;
  u8 *sel = sel_all + (u64)blk * BZ_MAX_SELECTORS;
  u8 *slab = slabs + (u64)blk * slab_cap;
  BzResult R{0, 0, BZ_ST_OK, 0, 0, 0, 0, 0};
  BzBits b{in, n, 0, false, 0, 0, 0};
  bz_seek(b, cands[blk].bit + 48);
  u32 status = BZ_ST_OK;
  u32 nblock = 0, orig_ptr = 0;
  if (cands[blk].kind != 0) {  // end-of-stream marker: just the combined CRC
    u32 c = bz_bits(b, 16);
    c = (c << 16) | bz_bits(b, 16);
    R.stored_crc = c;
    R.end_bit = b.bit;
    R.status = b.fault ? BZ_ST_RANGE : BZ_ST_OK;
    if (lane == 0) results[blk] = R;
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
    if (ngroups < 2 || ngroups > 6) { status = BZ_ST_FALSE; break; }
    const u32 nsel = bz_bits(b, 15);
    if (nsel < 1) { status = BZ_ST_FALSE; break; }
    if (nsel > BZ_MAX_SELECTORS) { status = BZ_ST_RANGE; break; }  // Dart: store past the Uint8List
    {
      u32 pos = 0x543210;  // MTF list of group numbers, 4 bits each
      bool bad = false;
      for (u32 i = 0; i < nsel; ++i) {
        u32 j = 0;
        while (bz_bits(b, 1)) { if (++j >= ngroups) { bad = true; break; } }
        if (bad || b.fault) break;
        const u32 v = (pos >> (4 * j)) & 15;
        const u32 low = pos & ((1u << (4 * j)) - 1);
        pos = (pos & ~((1u << (4 * (j + 1))) - 1)) | (low << 4) | v;
        if (lane == 0) sel[i] = (u8)v;
      }
      if (b.fault) { status = BZ_ST_RANGE; break; }
      if (bad) { status = BZ_ST_FALSE; break; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // lane 0's selector stores -> every lane's loads
      wave_sync();
    }
    {
      bool bad = false;
      for (u32 t = 0; t < ngroups && !bad; ++t) {
        i32 c = (i32)bz_bits(b, 5);
        for (u32 i = 0; i < alpha; ++i) {
          for (;;) {
            if (c < 1 || c > 20) { bad = true; break; }
            if (!bz_bits(b, 1)) break;
            if (!bz_bits(b, 1)) c++; else c--;
            if (b.fault) { bad = true; break; }
          }
          if (bad) break;
          if (lane == 0) L.len[t][i] = (u8)c;
        }
      }
      if (b.fault) { status = BZ_ST_RANGE; break; }
      if (bad) { status = BZ_ST_FALSE; break; }
    }
    wave_sync();
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
    for (u32 i = lane; i < 256; i += 64) { L.mtf[i] = (u8)i; L.unzftab[i] = 0; }
    wave_sync();
    // MTF / RUNA / RUNB loop
    const u32 eob = num_in_use + 1;
    i32 group_no = -1;
    u32 group_pos = 0, gsel = 0;
    i32 gmin = 0;
    auto get_mtf_val = [&]() -> i32 {
      if (group_pos == 0) {
        group_no++;
        if (group_no >= (i32)nsel) return -1;
        group_pos = 50;
        gsel = sel[group_no];
        gmin = L.min_len[gsel];
      }
      group_pos--;
      i32 zn = gmin;
      i32 zvec = (i32)bz_bits(b, (u32)zn);
      for (;;) {
        if (zn > 20) return -1;
        if (zvec <= L.limit[gsel][zn]) break;
        zn++;
        zvec = (zvec << 1) | (i32)bz_bits(b, 1);
      }
      const i32 idx = zvec - L.base[gsel][zn];
      if (idx < 0 || idx >= 258) return -1;
      return (i32)L.perm[gsel][idx];
    };
    i32 next_sym = get_mtf_val();
    bool bad = next_sym < 0;
    while (!bad && !b.fault && (u32)next_sym != eob) {
      if (next_sym == 0 || next_sym == 1) {
        i32 es = -1, N = 1;
        do {
          if (N >= 2 * 1024 * 1024) { bad = true; break; }
          es += (next_sym == 0) ? N : 2 * N;
          N *= 2;
          next_sym = get_mtf_val();
        } while (next_sym == 0 || next_sym == 1);
        if (bad) break;
        es++;
        const u32 uc = L.seq2unseq[L.mtf[0]];
        if (nblock + (u32)es > nblock_max) { bad = true; break; }
        if (lane == 0) L.unzftab[uc] += (u32)es;
        for (u32 k = lane; k < (u32)es; k += 64) tt[nblock + k] = uc;
        nblock += (u32)es;
        if (next_sym < 0) { bad = true; break; }
        continue;
      }
      if (nblock >= nblock_max) { bad = true; break; }
      const u32 nn = (u32)next_sym - 1;
      const u32 v = L.mtf[nn];
      wave_sync();
      // move to front: lanes shift the first nn entries up by one, highest 64-entry piece first
      for (i32 k0 = (i32)((nn ? nn - 1 : 0) & ~63u); k0 >= 0 && nn; k0 -= 64) {
        const u32 k = (u32)k0 + lane;
        const u8 x = k < nn ? L.mtf[k] : 0;
        wave_sync();
        if (k < nn) L.mtf[k + 1] = x;
        wave_sync();
      }
      if (lane == 0) {
        L.mtf[0] = (u8)v;
        const u32 uc = L.seq2unseq[v];
        L.unzftab[uc]++;
        tt[nblock] = uc;
      }
      wave_sync();
      nblock++;
      next_sym = get_mtf_val();
      if (next_sym < 0) bad = true;
    }
    if (b.fault) { status = BZ_ST_RANGE; break; }
    if (bad) { status = BZ_ST_FALSE; break; }
    if (orig_ptr >= nblock) { status = BZ_ST_FALSE; break; }
  } while (0);

  // phases 2 and 3 are separate launches (bz_tinv_scatter needs a 64 x 256 x u32 cursor table in LDS)
  R.status = status;
  R.nblock = nblock;
  R.end_bit = b.bit;
  R.crc = orig_ptr;  // carried to the next kernels
  if (lane == 0) results[blk] = R;
}

// ---- phase 2 (own launch): T^-1 ----
// tt[i] holds the block's bytes (low 8 bits).  Result: tt[j] |= i << 8 for the j-th smallest (byte, i).
__global__ __launch_bounds__(64) void bz_tinv_scatter(u32 *__restrict__ tt_all, u32 block_size100k,
                                                      const BzCand *__restrict__ cands, BzResult *__restrict__ results) {
  __shared__ u32 cur[64 * 256];  // 64 KiB: cursor of (lane, symbol)
  const u32 blk = blockIdx.x, lane = threadIdx.x;
  if (cands[blk].kind != 0 || results[blk].status != BZ_ST_OK) return;
  const u32 nblock = results[blk].nblock;
  u32 *tt = tt_all + (u64)blk * (100000u * block_size100k);
  for (u32 i = lane; i < 64 * 256; i += 64) cur[i] = 0;
  wave_sync();
  const u32 seg = (nblock + 63) / 64;
  const u32 lo = lane * seg < nblock ? lane * seg : nblock;
  const u32 hi = lo + seg < nblock ? lo + seg : nblock;
  for (u32 i = lo; i < hi; ++i) cur[lane * 256 + (tt[i] & 0xff)]++;
  wave_sync();
  // symbol-major exclusive scan: position of the first (sym, lane) element in sorted order
  u32 sym_total[4];
  for (u32 g = 0; g < 4; ++g) {  // lane handles symbol g*64 + lane: sum over lanes l
    const u32 s = g * 64 + lane;
    u32 acc = 0;
    for (u32 l = 0; l < 64; ++l) { const u32 c = cur[l * 256 + s]; cur[l * 256 + s] = acc; acc += c; }
    sym_total[g] = acc;
  }
  wave_sync();
  // exclusive scan of symbol totals over all 256 symbols (cftab)
  u32 run = 0;
  for (u32 g = 0; g < 4; ++g) {
    u32 tot;
    const u32 ex = wave_excl_sum(sym_total[g], tot);
    const u32 basep = run + ex;
    const u32 s = g * 64 + lane;
    for (u32 l = 0; l < 64; ++l) cur[l * 256 + s] += basep;
    run += tot;
  }
  wave_sync();
  // scatter: stable within a lane's segment, lanes ordered by segment -> identical to the serial loop
  for (u32 i = lo; i < hi; ++i) {
    const u32 s = tt[i] & 0xff;
    const u32 pos = cur[lane * 256 + s]++;
    atomicOr(&tt[pos], i << 8);  // pos may lie in another lane's segment that is still being read (low byte untouched)
  }
}

// ---- phase 3 (own launch): inverse BWT + un-RLE + CRC, one lane per block ----
// direct_off == nullptr: every decoded block goes to its slab (a block that outgrows the slab keeps
// counting and reports BZ_ST_OVERFLOW with its true size).  direct_off != nullptr: second pass for
// exactly those blocks, written straight to direct_out + direct_off[blk].
__global__ __launch_bounds__(64) void bz_unbwt(const u32 *__restrict__ tt_all, u32 block_size100k, u32 ncand,
                                               const BzCand *__restrict__ cands, u8 *__restrict__ slabs, u64 slab_cap,
                                               BzResult *__restrict__ results, const u32 *__restrict__ crc_table,
                                               const u64 *__restrict__ direct_off, u8 *__restrict__ direct_out) {
  const u32 blk = blockIdx.x * 64 + threadIdx.x;
  if (blk >= ncand) return;
  if (cands[blk].kind != 0) return;
  if (direct_off) { if (direct_off[blk] == ~0ull) return; }
  else if (results[blk].status != BZ_ST_OK) return;
  const u32 nblock_max = 100000u * block_size100k;
  const u32 *tt = tt_all + (u64)blk * nblock_max;
  u8 *out = direct_off ? direct_out + direct_off[blk] : slabs + (u64)blk * slab_cap;
  if (direct_off) slab_cap = ~0ull;
  const u32 nblock = results[blk].nblock, orig_ptr = direct_off ? results[blk].pad_orig_ptr : results[blk].crc;
  u32 crc = 0xffffffffu;
  u64 olen = 0;
  u32 status = BZ_ST_OK;
  u32 t_pos = tt[orig_ptr] >> 8;
  u32 n_used = 0;
  const u32 save_pp = nblock + 1;
#define BZ_EMIT(ch)                                                     \
  do {                                                                  \
    if (olen < slab_cap) out[olen] = (u8)(ch); else status = BZ_ST_OVERFLOW; \
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
  if (direct_off) return;  // sizes and CRC were established by the first pass
  results[blk].status = status;
  results[blk].out_len = olen;
  results[blk].crc = crc ^ 0xffffffffu;
  if (status == BZ_ST_OVERFLOW) results[blk].pad_orig_ptr = orig_ptr;
}

__global__ __launch_bounds__(256) void bz_gather(const u8 *__restrict__ slabs, u64 slab_cap, const u32 *__restrict__ order,
                                                 const u64 *__restrict__ off, const u64 *__restrict__ len, u8 *__restrict__ out) {
  const u32 k = blockIdx.y;
  const u8 *src = slabs + (u64)order[k] * slab_cap;
  u8 *dst = out + off[k];
  const u64 n = len[k];
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace ahip
