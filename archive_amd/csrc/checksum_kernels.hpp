// checksum_kernels.hpp -- CRC-32 and Adler-32 of device-resident data on gfx950.
//
// Reference: /root/reference/lib/src/util/crc32.dart:6-27 (getCrc32: reflected polynomial 0xEDB88320, byte at
// a time) and util/adler32.dart:29-52 (getAdler32).  Both are serial recurrences; both are linear, which is
// what makes them data-parallel:
//
//   CRC-32   One byte step is  s <- Z(s ^ b),  Z(s) = tab[s & 0xff] ^ (s >> 8), i.e. multiplication by x^8 in
//            GF(2)[x]/P.  With a zero initial state the final state is  sum over 4-byte words w at offset o of
//            Z^(N-o)(w).  Lane l of a wave reads the dwords at 256 i + 4 l (perfectly coalesced rows of 256 B)
//            and keeps the Horner sum  a <- Z^256(a) ^ w  -- four LDS lookups in tables U0..U3 (Z^256 of each
//            byte of a), the same cost as slicing-by-4.  Per 64 KiB segment the lane sums are aligned with one
//            multiplication by x^(8(256-4l)), XORed across the wave, and shifted to the end of the data with
//            x^(8 * suffix) (square-and-multiply on precomputed x^(2^k)); the host folds in the initial value.
//   Adler-32 s1 = a0 + sum b_j,  s2 = b0 + n a0 + sum (n - j) b_j.  Lanes accumulate  sum b  and  sum j b
//            (64-bit), waves add them with one atomic each.
#pragma once
#include "common.hpp"

#ifndef AHIP_HD
#define AHIP_HD __host__ __device__ inline
#endif

namespace ahip {

constexpr u32 CK_ROW = 256;                 // bytes per wave row (64 lanes x 4)
constexpr u32 CK_SEG = 65536;               // bytes per wave segment
constexpr u32 CK_POLY = 0xEDB88320u;
constexpr u32 CK_TAB_WORDS = 256 * 5 + 64;  // tab, U0..U3, x^(2^k) for k < 64

// a(x) * b(x) mod P, reflected representation (bit 31 = x^0) -- zlib's multmodp
AHIP_HD u32 ck_mulmod(u32 a, u32 b) {
  u32 m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1)) == 0) break;
    }
    m >>= 1;
    b = (b & 1) ? ((b >> 1) ^ CK_POLY) : (b >> 1);
  }
  return p;
}
// x^(8 n) mod P;  pw[k] = x^(2^k) mod P
AHIP_HD u32 ck_xpow8(u64 nbytes, const u32 *pw) {
  u32 p = 1u << 31;
  u32 k = 3;
  while (nbytes) {
    if (nbytes & 1) p = ck_mulmod(pw[k & 63], p);
    nbytes >>= 1;
    ++k;
  }
  return p;
}

// host: fill the table block the kernel expects
inline void ck_build_tables(u32 *t) {
  for (u32 i = 0; i < 256; ++i) {
    u32 c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (CK_POLY ^ (c >> 1)) : (c >> 1);
    t[i] = c;
  }
  for (u32 k = 0; k < 4; ++k)
    for (u32 b = 0; b < 256; ++b) {
      u32 s = b << (8 * k);
      for (u32 z = 0; z < CK_ROW; ++z) s = t[s & 0xff] ^ (s >> 8);
      t[256 + 256 * k + b] = s;
    }
  u32 p = 1u << 30;  // x^1
  t[1280] = p;
  for (u32 k = 1; k < 64; ++k) t[1280 + k] = p = ck_mulmod(p, p);
}

// The wave's share (segments wave, wave + nwaves, ...) of the raw CRC state of d[0, n) -- zero initial value, no final
// inversion -- already shifted to the end of the data: the XOR of these over all waves is the state.  T = the five LDS
// tables, pw = the powers behind them.  *any = the wave had a segment at all.
// REV8: every data byte is taken with its bits reversed.  The reflected register over bit-reversed bytes is the
// mirror image of the MSB-first register over the bytes as they are, so __brev of the result is the state of the
// MSB-first CRC-32 (polynomial 04c11db7: bzip2's, bzip2_decoder.dart:640-727 / util/crc32.dart's BZip2 table).
template <bool REV8>
AHIP_DEVINL u32 ck_raw_wave(const u8 *__restrict__ d, u64 n, const u32 *T, const u32 *pw, u64 wave, u64 nwaves, u32 lane, bool *any_out) {
  auto word = [](u32 w) -> u32 { return REV8 ? __builtin_bswap32(__brev(w)) : w; };
  const u32 lane_shift = ck_xpow8(CK_ROW - 4 * lane, pw);  // aligns this lane's Horner sum with the end of a row
  const u64 nseg = (n + CK_SEG - 1) / CK_SEG;
  // the wave's segments are folded into one value aligned with the end of the latest one: consecutive
  // segments of a wave end nwaves * CK_SEG bytes apart (one precomputed multiplier), except a short last one
  const u32 gap_shift = ck_xpow8(nwaves * CK_SEG, pw);
  u32 accw = 0;
  u64 end_prev = 0;
  bool any = false;
  for (u64 sg = wave; sg < nseg; sg += nwaves) {
    const u64 s0 = sg * CK_SEG;
    const u32 len = (u32)((n - s0) < CK_SEG ? (n - s0) : CK_SEG);
    const u32 rows = len / CK_ROW, rem = len % CK_ROW;
    const u8 *p = d + s0 + 4 * lane;
    u32 a = 0;
    u32 i = 0;
    for (; i + 4 <= rows; i += 4) {  // four loads in flight
      u32 w[4];
#pragma unroll
      for (u32 u = 0; u < 4; ++u) w[u] = word(load_u32_unaligned(p + (u64)(i + u) * CK_ROW));
#pragma unroll
      for (u32 u = 0; u < 4; ++u)
        a = T[256 + (a & 0xff)] ^ T[512 + ((a >> 8) & 0xff)] ^ T[768 + ((a >> 16) & 0xff)] ^ T[1024 + (a >> 24)] ^ w[u];
    }
    for (; i < rows; ++i)
      a = T[256 + (a & 0xff)] ^ T[512 + ((a >> 8) & 0xff)] ^ T[768 + ((a >> 16) & 0xff)] ^ T[1024 + (a >> 24)] ^
          word(load_u32_unaligned(p + (u64)i * CK_ROW));
    u32 v = rows ? ck_mulmod(lane_shift, a) : 0u;
    // xor across the wave
    for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o);
    u32 c = v;
    if (rem) {  // the last, partial row (only the data's final segment has one), byte by byte
      u32 r = 0;
      const u8 *q = d + s0 + (u64)rows * CK_ROW;
      for (u32 k = 0; k < rem; ++k) r = T[(r ^ (REV8 ? __brev((u32)q[k]) >> 24 : (u32)q[k])) & 0xff] ^ (r >> 8);
      c = ck_mulmod(ck_xpow8(rem, pw), v) ^ r;
    }
    const u64 end = s0 + len;
    if (any) accw = ck_mulmod(end - end_prev == nwaves * CK_SEG ? gap_shift : ck_xpow8(end - end_prev, pw), accw) ^ c;
    else accw = c;
    end_prev = end;
    any = true;
  }
  *any_out = any;
  return any ? ck_mulmod(ck_xpow8(n - end_prev, pw), accw) : 0u;
}

// acc[0] ^= raw CRC state (zero initial value, no final inversion) of d[0, n)
__global__ __launch_bounds__(256) void crc32_kernel(const u8 *__restrict__ d, u64 n, const u32 *__restrict__ tables,
                                                    u32 *__restrict__ acc) {
  __shared__ u32 T[256 * 5];
  for (u32 i = threadIdx.x; i < 256 * 5; i += 256) T[i] = tables[i];
  __syncthreads();
  const u32 lane = threadIdx.x & 63;
  bool any;
  const u32 total = ck_raw_wave<false>(d, n, T, tables + 1280, (u64)blockIdx.x * 4 + (threadIdx.x >> 6), (u64)gridDim.x * 4, lane, &any);
  if (any && lane == 0 && total) atomicXor(acc, total);
}

// acc[0] += sum of bytes, acc[1] += sum of (index * byte), both over d[0, n)
__global__ __launch_bounds__(256) void adler32_kernel(const u8 *__restrict__ d, u64 n, unsigned long long *__restrict__ acc) {
  const u32 lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (u64)gridDim.x * 4;
  const u64 nseg = (n + CK_SEG - 1) / CK_SEG;
  u64 A = 0, Tm = 0;
  for (u64 sg = wave; sg < nseg; sg += nwaves) {
    const u64 s0 = sg * CK_SEG;
    const u32 len = (u32)((n - s0) < CK_SEG ? (n - s0) : CK_SEG);
    const u32 rows = len / CK_ROW, rem = len % CK_ROW;
    const u8 *p = d + s0 + 4 * lane;
    u32 sa = 0;      // sum of bytes            (< 2^26 per segment)
    u64 st = 0;      // sum of local index * byte
    for (u32 i = 0; i < rows; ++i) {
      const u32 w = load_u32_unaligned(p + (u64)i * CK_ROW);
      const u32 b0 = w & 0xff, b1 = (w >> 8) & 0xff, b2 = (w >> 16) & 0xff, b3 = w >> 24;
      const u32 s = b0 + b1 + b2 + b3;
      sa += s;
      st += (u64)(i * CK_ROW + 4 * lane) * s + (b1 + 2 * b2 + 3 * b3);
    }
    if (lane < rem) {  // the last, partial row: one byte per lane, four lanes' worth at most per lane
      for (u32 k = lane; k < rem; k += 64) {
        const u32 b = d[s0 + (u64)rows * CK_ROW + k];
        sa += b;
        st += (u64)(rows * CK_ROW + k) * b;
      }
    }
    A += sa;
    Tm += (s0 % 65521) * sa + st % 65521;  // index = s0 + local index; everything modulo 65521 on the host
  }
  // wave sums, one atomic pair per wave
  for (int o = 32; o; o >>= 1) { A += __shfl_xor(A, o); Tm += __shfl_xor(Tm, o); }
  if (lane == 0 && (A | Tm)) { atomicAdd(acc, (unsigned long long)A); atomicAdd(acc + 1, (unsigned long long)(Tm % 65521)); }
}

}  // namespace ahip
