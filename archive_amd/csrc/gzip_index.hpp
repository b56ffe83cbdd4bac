// gzip_index.hpp -- device-side member index for multi-member gzip.
//
// The reference discovers member boundaries serially, by inflating each member
// (codecs/zlib/_gzip_decoder_web.dart:27-58).  To shard members over wavefronts (and GPUs)
// the boundaries have to be known up front, so this builds them on the GPU:
//
//   1. gz_count_candidates / gz_write_candidates : every byte position holding `1f 8b 08`
//      (ID1 ID2 CM, _gzip_decoder_web.dart:99-109) becomes a candidate, in position order.
//   2. gz_parse_headers : the reference's _readHeader (:60-138) per candidate.  A BGZF-style
//      `BC` FEXTRA subfield (skipped by the reference, :119-122) yields the member size, so
//      next member = pos + BSIZE + 1 and ISIZE = u32 at next-4.
//   3. candidates without `BC`: a sizing run of the inflate kernel (no stores) yields the
//      true end position and size of every candidate.
//   4. gz_chain : the member chain is the orbit of position 0 under `next`; false candidates
//      (the magic inside compressed data) are not on it.  Pointer doubling + scans give the
//      ordered member list and each member's output offset.
#pragma once
#include "common.hpp"

namespace ahip {

constexpr u32 TILE_BYTES = 65536;  // 256 threads x 16 x 16 bytes (fewer tiles: the scan over their counts is one workgroup)
constexpr u64 POS_UNKNOWN = ~0ull;

constexpr u32 HF_BC = 1;     // size known from the BC subfield
constexpr u32 HF_RANGE = 2;  // header runs past the end of the input: RangeError in the reference
constexpr u32 HF_SIZED = 4;  // next_pos/size come from a sizing run
constexpr u32 HF_RETOK = 8;  // ... whose tokens cannot serve the decode proper (error, q8 reach, full token area, late path)
constexpr u32 HF_RANGE_SIZED = 16;  // HF_RANGE set by gz_apply_sizing (the member's trailer runs past the input), not by the header parse:
                                    // a second sizing pass (plan_build's size_oversub rebuild) looks at such a member again

struct GzHeader {
  u64 payload_off;  // first DEFLATE byte
  u64 next_pos;     // position after the 8-byte trailer (POS_UNKNOWN until sized)
  u64 size;         // decoded size (ISIZE when HF_BC, else from the sizing run)
  u32 flags;
  u32 status;       // expected MS_* of the member (MS_OK for BC members)
};

struct ChainSummary {
  u64 members;       // members on the chain
  u64 total_out;     // sum of their sizes
  u64 payload_bytes; // sum of (next_pos - pos) over the chain = compressed bytes incl. framing
  u64 tail_pos;      // stream position where the chain stopped (== in_len when it consumed everything)
  u32 first_is_gzip; // candidate 0 sits at the start position
  u32 range_error;   // a header or trailer on the chain runs past the end
  u32 unknown;       // candidates whose size is not known yet (valid after gz_parse_headers)
  u32 stopped_unknown; // the chain reached a member of unknown size: a sizing run is needed
  u32 retok;         // members on the chain that carry HF_RETOK
  u32 oversub;       // members on the chain a sizing run left UNSIZED: over-subscribed code lengths (MS_OVERSUB), which only the
                     // late kernel's exact tables decode -- the plan then sizes once more with that kernel in the launch
};

// 16-bit mask of candidate positions base+0 .. base+15
AHIP_DEVINL u32 candidate_mask16(const u8 *in, u64 n, u64 base) {
  if (base >= n) return 0;
  u64 w0, w1;
  u32 w2;
  if (base + 20 <= n) {
    const uint4 v = load_u128_unaligned(in + base);  // one 16-byte load + the dword behind it
    w0 = (u64)v.x | ((u64)v.y << 32);
    w1 = (u64)v.z | ((u64)v.w << 32);
    w2 = load_u32_unaligned(in + base + 16);
  } else {
    u8 t[20];
    for (int k = 0; k < 20; ++k) t[k] = (base + k < n) ? in[base + k] : 0;
    w0 = w1 = 0; w2 = 0;
    for (int k = 0; k < 8; ++k) { w0 |= (u64)t[k] << (8 * k); w1 |= (u64)t[8 + k] << (8 * k); }
    for (int k = 0; k < 4; ++k) w2 |= (u32)t[16 + k] << (8 * k);
  }
  u32 mask = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    u32 v;
    if (k < 6) v = (u32)(w0 >> (8 * k));
    else if (k < 8) v = (u32)((w0 >> (8 * k)) | (w1 << (64 - 8 * k)));
    else if (k < 14) v = (u32)(w1 >> (8 * (k - 8)));
    else v = (u32)((w1 >> (8 * (k - 8))) | ((u64)w2 << (64 - 8 * (k - 8))));
    if ((v & 0xffffff) == 0x088b1f && base + k + 3 <= n) mask |= 1u << k;
  }
  return mask;
}

AHIP_DEVINL u32 block_reduce_add_256(u32 v, u32 *sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  u32 t = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return t;
}

// tile_slots: the offsets (inside the tile) of a tile's candidates when it has at most TILE_SLOTS of them -- nearly
// every tile: members are tens of KiB apart -- so that gz_write_candidates does not have to read the input again.
constexpr u32 TILE_SLOTS = 8;
// What the header parser needs of a candidate at position p, picked up WHILE the scan streams over it: the 8 bytes in
// front of p (CRC-32 and ISIZE of the member that ends there) and the 32 bytes from p on (a BGZF header is 18 bytes).
// The parser then reads these records in candidate order instead of 2 x 65 536 random places of a 1.7 GB stream -- a
// random 32-byte read per candidate cost 0.23 ms per decode in address translation alone.
struct CandRec { u64 w[5]; };  // w[0]: bytes [p - 8, p); w[1..4]: bytes [p, p + 32); zero where the stream has none
AHIP_DEVINL void gz_gather_rec(const u8 *in, u64 n, u64 p, CandRec &r) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const i64 a = (i64)p - 8 + 8 * k;
    u64 v = 0;
    if (a >= 0 && (u64)a + 8 <= n) v = load_u64_unaligned(in + a);
    else for (int b = 0; b < 8; ++b) { const i64 q = a + b; if (q >= 0 && (u64)q < n) v |= (u64)in[q] << (8 * b); }
    r.w[k] = v;
  }
}
__global__ __launch_bounds__(256) void gz_count_candidates(const u8 *in, u64 start, u64 n, u32 *tile_counts, u16 *tile_slots,
                                                           CandRec *tile_recs) {
  __shared__ u32 sm[4];
  __shared__ u32 nslot;
  __shared__ u32 slot[TILE_SLOTS];
  if (threadIdx.x == 0) nslot = 0;
  __syncthreads();
  u32 c = 0;
  // A candidate starts with the byte 1f.  Inside the stream (the whole tile and the 20 bytes behind it exist) the tile's
  // sixteen 16-byte loads of a thread are issued back to back and screened for that byte with four operations a dword
  // (the zero-byte test on x ^ 1f1f1f1f); only the rows that have one -- 6 % -- go through the byte-by-byte comparison.
  // (One row after the other, each behind its own bounds checks, the loads waited for one another: 0.50 ms a decode.)
  const u64 tile0 = start + (u64)blockIdx.x * TILE_BYTES;
  u32 rows = 0xffffu;
  if (tile0 + TILE_BYTES + 20 <= n) {
    uint4 v[TILE_BYTES / 4096];
#pragma unroll
    for (u32 r = 0; r < TILE_BYTES / 4096; ++r) v[r] = load_u128_unaligned(in + tile0 + r * 4096 + threadIdx.x * 16);
    rows = 0;
#pragma unroll
    for (u32 r = 0; r < TILE_BYTES / 4096; ++r) {
      const u32 zx = v[r].x ^ 0x1f1f1f1fu, zy = v[r].y ^ 0x1f1f1f1fu, zz = v[r].z ^ 0x1f1f1f1fu, zw = v[r].w ^ 0x1f1f1f1fu;
      const u32 any = (((zx - 0x01010101u) & ~zx) | ((zy - 0x01010101u) & ~zy) | ((zz - 0x01010101u) & ~zz) | ((zw - 0x01010101u) & ~zw)) & 0x80808080u;
      rows |= any ? 1u << r : 0u;
    }
  }
  static_assert(TILE_BYTES / 4096 == 16, "one bit of `rows` per row of the tile");
  for (; rows; rows &= rows - 1) {
    const u32 rel = ((u32)__ffs(rows) - 1) * 4096 + threadIdx.x * 16;
    u32 mask = candidate_mask16(in, n, tile0 + rel);
    c += __popc(mask);
    while (mask) {  // (rare: one thread in a hundred thousand)
      const u32 k = (u32)__ffs(mask) - 1;
      mask &= mask - 1;
      const u32 i = atomicAdd(&nslot, 1u);
      if (i < TILE_SLOTS) slot[i] = rel + k;
    }
  }
  u32 t = block_reduce_add_256(c, sm);  // (its barriers also order the slot writes)
  if (threadIdx.x == 0) {
    tile_counts[blockIdx.x] = t;
    if (t && t <= TILE_SLOTS) {  // in position order
      u32 v[TILE_SLOTS];
      for (u32 i = 0; i < TILE_SLOTS; ++i) v[i] = i < t ? slot[i] : 0xffffffffu;
      for (u32 i = 1; i < TILE_SLOTS; ++i)
        for (u32 j = i; j > 0 && v[j - 1] > v[j]; --j) { const u32 x = v[j]; v[j] = v[j - 1]; v[j - 1] = x; }
      for (u32 i = 0; i < t; ++i) { tile_slots[(u64)blockIdx.x * TILE_SLOTS + i] = (u16)v[i]; slot[i] = v[i]; }
    }
  }
  __syncthreads();
  if (t && t <= TILE_SLOTS && threadIdx.x < t) {
    CandRec r;
    gz_gather_rec(in, n, start + (u64)blockIdx.x * TILE_BYTES + slot[threadIdx.x], r);
    tile_recs[(u64)blockIdx.x * TILE_SLOTS + threadIdx.x] = r;
  }
}

// Exclusive scan of `num` u32 values by ONE workgroup of 1024 threads; total to *total.
__global__ __launch_bounds__(1024) void scan_exclusive_u32(const u32 *in, u32 *out, u64 num, u32 *total) {
  __shared__ u32 wsum[16];
  __shared__ u32 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (u64 base = 0; base < num; base += 1024) {
    u64 i = base + threadIdx.x;
    u32 v = (i < num) ? in[i] : 0;
    u32 x = v;
    for (int o = 1; o < 64; o <<= 1) { u32 y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    u32 woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    u32 carry = carry_s;
    if (i < num) out[i] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

// 256 tiles per workgroup when their candidates are in the slots; a tile with more than TILE_SLOTS is scanned again
// by a workgroup of its own (grid: tiles / 256 + tiles workgroups, the first ones serve the slots)
__global__ __launch_bounds__(256) void gz_write_candidates(const u8 *in, u64 start, u64 n, const u32 *tile_counts,
                                                           const u32 *tile_offsets, u64 *cand_pos, const u16 *tile_slots,
                                                           u32 tiles, const CandRec *tile_recs, CandRec *cand_rec) {
  const u32 slot_wgs = (tiles + 255) / 256;
  if (blockIdx.x < slot_wgs) {
    const u32 tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= tiles) return;
    const u32 c = tile_counts[tile];
    if (c == 0 || c > TILE_SLOTS) return;
    const u32 off = tile_offsets[tile];
    for (u32 i = 0; i < c; ++i) {
      cand_pos[off + i] = start + (u64)tile * TILE_BYTES + tile_slots[(u64)tile * TILE_SLOTS + i];
      cand_rec[off + i] = tile_recs[(u64)tile * TILE_SLOTS + i];
    }
    return;
  }
  const u32 bx = blockIdx.x - slot_wgs;  // the tile this workgroup scans again if it has to
  if (tile_counts[bx] <= TILE_SLOTS) return;  // uniform per block
  __shared__ u32 wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u32 tile_off = tile_offsets[bx];
  for (u32 r = 0; r < TILE_BYTES / 4096; ++r) {  // sub-tiles in position order
    u64 base = start + (u64)bx * TILE_BYTES + r * 4096 + threadIdx.x * 16;
    u32 mask = candidate_mask16(in, n, base);
    u32 c = __popc(mask), x = c;
    for (int o = 1; o < 64; o <<= 1) { u32 y = __shfl_up(x, o); if (lane >= o) x += y; }
    __syncthreads();
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    u32 off = tile_off + x - c;
    for (int k = 0; k < w; ++k) off += wsum[k];
    while (mask) {
      int k = __ffs(mask) - 1;
      mask &= mask - 1;
      CandRec r;
      gz_gather_rec(in, n, base + k, r);
      cand_rec[off] = r;
      cand_pos[off++] = base + k;
    }
    tile_off += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

// _GZipDecoder._readHeader for one candidate (signature and CM already matched).
// Fast path: the first 32 bytes of the header in four registers -- a BGZF header is 18 bytes, a plain one 10 -- so the
// flags, XLEN and the subfields cost ONE round trip to memory instead of one per field; FNAME / FCOMMENT (a scan for a
// zero byte) or a longer FEXTRA take the byte-wise path, which is the same parse.
// FNAME / FCOMMENT: the position behind the first zero byte at or after q (n when there is none) -- what the reference's
// readString loop leaves (_gzip_decoder_web.dart:124-131).  64 bytes per round trip to memory: the few dozen FALSE
// candidates inside compressed data have random flag bytes, and one of them walking byte by byte (a dependent load
// each) to the next zero byte set the whole kernel's time -- 0.25 ms for 0.03 ms of work.
AHIP_DEVINL u64 gz_skip_string(const u8 *in, u64 n, u64 q) {
  while (q + 64 <= n) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = load_u128_unaligned(in + q + 16 * k);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 d[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 z = (d[j] - 0x01010101u) & ~d[j] & 0x80808080u;  // the lowest set flag marks the first zero byte
        if (z) return q + 16 * k + 4 * j + ((u32)__builtin_ctz(z) >> 3) + 1;
      }
    }
    q += 64;
  }
  while (q < n) { if (in[q++] == 0) break; }
  return q;
}
AHIP_DEVINL u32 hdr_byte(const u64 (&w)[4], u32 k) {  // k < 32
  const u64 a = k < 16 ? (k < 8 ? w[0] : w[1]) : (k < 24 ? w[2] : w[3]);
  return (u32)(a >> (8 * (k & 7))) & 0xffu;
}
__global__ __launch_bounds__(256) void gz_parse_headers(const u8 *in, u64 n, const u64 *cand_pos, u32 K,
                                                        GzHeader *hdr, ChainSummary *sum, const CandRec *cand_rec) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K) return;
  u64 p = cand_pos[i];
  GzHeader h;
  h.payload_off = 0; h.next_pos = POS_UNKNOWN; h.size = 0; h.flags = 0; h.status = MS_OK;
  bool range = false;
  u64 q = p + 3;
  u32 flags = 0;
  u64 bc_next = POS_UNKNOWN;
  bool parsed = false;
  if (p + 32 <= n) {
    u64 w[4];
    const CandRec rec = cand_rec[i];
    w[0] = rec.w[1]; w[1] = rec.w[2]; w[2] = rec.w[3]; w[3] = rec.w[4];
    flags = hdr_byte(w, 3);
    u32 r = 10;  // offset of the next field
    bool ok = (flags & 0x18) == 0;
    if (ok && (flags & 0x04)) {
      const u32 xlen = hdr_byte(w, 10) | (hdr_byte(w, 11) << 8);
      r = 12;
      const u32 xend = r + xlen;
      ok = xend <= 30;  // (+ 2 bytes of FHCRC at most: all of it inside the 32 bytes)
      if (ok) {
        u32 sp = r;
        while (sp + 4 <= xend) {
          const u32 slen = hdr_byte(w, sp + 2) | (hdr_byte(w, sp + 3) << 8);
          if (hdr_byte(w, sp) == 66 && hdr_byte(w, sp + 1) == 67 && slen == 2 && sp + 6 <= xend) {
            bc_next = p + (hdr_byte(w, sp + 4) | (hdr_byte(w, sp + 5) << 8)) + 1;
            break;
          }
          sp += 4 + slen;
        }
        r = xend;
      }
    }
    if (ok) {
      if (flags & 0x02) r += 2;
      q = p + r;
      parsed = true;
    }
  }
  if (!parsed) {
    bc_next = POS_UNKNOWN;
    q = p + 3;
    if (q < n) flags = in[q]; else range = true;
    q = p + 10;  // flags, mtime(4), xfl, os
    if (q > n) range = true;
    if (!range && (flags & 0x04)) {
      if (q + 2 > n) range = true;
      else {
        u32 xlen = in[q] | ((u32)in[q + 1] << 8);
        q += 2;
        u64 xend = q + xlen;
        if (xend > n) xend = n;  // readBytes clamps
        // look for SI1='B' SI2='C' SLEN=2 (BGZF); the reference skips the whole field
        u64 s = q;
        while (s + 4 <= xend) {
          u32 slen = in[s + 2] | ((u32)in[s + 3] << 8);
          if (in[s] == 66 && in[s + 1] == 67 && slen == 2 && s + 6 <= xend) {
            u32 bsize = in[s + 4] | ((u32)in[s + 5] << 8);
            bc_next = p + bsize + 1;
            break;
          }
          s += 4 + slen;
        }
        q = xend;
      }
    }
    if (!range && (flags & 0x08)) q = gz_skip_string(in, n, q);
    if (!range && (flags & 0x10)) q = gz_skip_string(in, n, q);
    if (!range && (flags & 0x02)) { if (q + 2 > n) range = true; else q += 2; }
  }
  h.payload_off = q;
  if (range) {
    h.flags |= HF_RANGE;
    h.next_pos = n;
  } else if (bc_next != POS_UNKNOWN && bc_next <= n && bc_next >= q + 8) {
    h.flags |= HF_BC;
    h.next_pos = bc_next;
    // ISIZE: the 4 bytes in front of the next member -- which the next candidate's record holds when it IS the next member
    if (i + 1 < K && cand_pos[i + 1] == bc_next) h.size = (u32)(cand_rec[i + 1].w[0] >> 32);
    else h.size = load_u32_unaligned(in + bc_next - 4);
  } else {
    atomicAdd(&sum->unknown, 1u);
  }
  hdr[i] = h;
}

// Sizing run plumbing: one MemberDesc per candidate, then fold the results back.
__global__ __launch_bounds__(256) void gz_make_sizing_descs(const GzHeader *hdr, u32 K, MemberDesc *descs) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K) return;
  MemberDesc d;
  d.in_off = hdr[i].payload_off;
  d.out_off = 0;
  d.out_limit = ~0ull;
  d.expect_end = POS_UNKNOWN;
  d.in_end = 0;
  d.hist = 32768;  // output offsets are not known yet: permissive here, exact in the decode proper
  d.pad = 0;
  descs[i] = d;
}
__global__ __launch_bounds__(256) void gz_apply_sizing(GzHeader *hdr, u32 K, const MemberResult *res, u64 n) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K) return;
  GzHeader h = hdr[i];
  if ((h.flags & HF_RANGE) && !(h.flags & HF_RANGE_SIZED)) return;  // the header itself runs past the input
  MemberResult r = res[i];
  h.flags = (h.flags & ~(HF_BC | HF_RETOK | HF_RANGE | HF_RANGE_SIZED)) | HF_SIZED;
  if (r.status != MS_OK || (r.blocks & MR_FAR)) h.flags |= HF_RETOK;
  h.size = r.out_len;
  h.status = r.status;
  h.next_pos = r.end_pos + 8;  // CRC32 + ISIZE are read unconditionally (:40-41)
  if (h.next_pos > n) { h.flags |= HF_RANGE | HF_RANGE_SIZED; h.next_pos = n; }
  hdr[i] = h;
}

// Member chain from `start`: the orbit of candidate 0 under `next`.  Four small kernels, only the second of which is a
// single workgroup (and it only touches the exceptions):
//   gz_link        every candidate's successor index nxt[i] (the next candidate in position order, nearly always; a
//                  binary search otherwise); reach[i] = 1; candidates with nxt[i] != i + 1 are appended to a list
//   gz_chain_fix   the chain runs through consecutive candidates except where a false magic inside compressed data
//                  (about one per 16 MiB) or garbage interrupts it: sort the exceptions, let one thread hop between
//                  them, clear reach[] over what the hops skip.  More exceptions than the list holds (adversarial
//                  input): pointer doubling over all candidates, as before.
//   gz_chain_sums  per 1024 candidates: members, output bytes, compressed bytes on the chain
//   gz_chain_emit  ordered member list + output offsets (the workgroup's prefix = the sums in front of it)
constexpr u32 EXC_CAP = 3072, EXC_SORT = 4096;  // listed exceptions; the power of two their sort pads to
struct ChainExc { u32 idx, nxt; };
__global__ __launch_bounds__(256) void gz_link(const u64 *cand_pos, const GzHeader *hdr, u32 K, u64 n, u32 *nxt, u32 *reach,
                                               ChainExc *exc, u32 *n_exc) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i > K) return;
  if (i == K) { nxt[K] = K; reach[K] = 0; return; }
  const u64 np = hdr[i].next_pos;
  u32 j = K;
  if (!(hdr[i].flags & HF_RANGE) && np < n) {
    if (i + 1 < K && cand_pos[i + 1] == np) j = i + 1;
    else {
      u32 lo = i + 1, hi = K;  // candidates are sorted; next_pos > pos
      while (lo < hi) { u32 mid = (lo + hi) >> 1; if (cand_pos[mid] < np) lo = mid + 1; else hi = mid; }
      if (lo < K && cand_pos[lo] == np) j = lo;
    }
  }
  nxt[i] = j;
  reach[i] = 1;
  if (j != i + 1) {  // (the last candidate of a clean stream is one: its successor is K)
    const u32 slot = atomicAdd(n_exc, 1u);
    if (slot < EXC_CAP) exc[slot] = ChainExc{i, j};
  }
}
__global__ __launch_bounds__(1024) void gz_chain_fix(const u64 *cand_pos, u32 K, u64 start, const u32 *nxt, u32 *jmp, u32 *jmp2,
                                                     u32 *reach, const ChainExc *exc, const u32 *n_exc_p) {
  const u32 tid = threadIdx.x;
  const bool first_ok = K > 0 && cand_pos[0] == start;
  const u32 n_exc = *n_exc_p;
  if (!first_ok) {  // nothing is on the chain
    for (u32 i = tid; i < K; i += 1024) reach[i] = 0;
    return;
  }
  if (n_exc <= EXC_CAP) {
    __shared__ u32 e_idx[EXC_SORT], e_nxt[EXC_SORT];
    __shared__ u32 skip_lo[EXC_CAP + 1], skip_hi[EXC_CAP + 1];  // candidates [lo, hi) the chain jumps over
    __shared__ u32 n_skip;
    u32 P = 1;
    while (P < n_exc) P <<= 1;
    for (u32 k = tid; k < P; k += 1024) {
      const bool have = k < n_exc;
      e_idx[k] = have ? exc[k].idx : 0xffffffffu;
      e_nxt[k] = have ? exc[k].nxt : 0u;
    }
    __syncthreads();
    for (u32 size = 2; size <= P; size <<= 1)  // bitonic sort by candidate index (the list was appended in no order)
      for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
        for (u32 k = tid; k < P / 2; k += 1024) {
          const u32 a = ((k & ~(stride - 1)) << 1) | (k & (stride - 1)), b = a | stride;
          const bool up = (a & size) == 0;
          const u32 ia = e_idx[a], ib = e_idx[b];
          if ((ia > ib) == up) { e_idx[a] = ib; e_idx[b] = ia; const u32 t = e_nxt[a]; e_nxt[a] = e_nxt[b]; e_nxt[b] = t; }
        }
        __syncthreads();
      }
    if (tid == 0) {
      u32 r = 0, i = 0, p = 0;
      for (;;) {  // at candidate i, on the chain
        while (p < n_exc && e_idx[p] < i) ++p;
        if (p == n_exc) break;  // consecutive to the last candidate (whose successor K is itself an exception: not reached here)
        const u32 at = e_idx[p], to = e_nxt[p];
        // the chain runs i .. at, then jumps to `to` (K: it ends) over (at, to)
        skip_lo[r] = at + 1; skip_hi[r] = to < K ? to : K; ++r;
        if (to >= K) break;
        i = to;
      }
      n_skip = r;
    }
    __syncthreads();
    for (u32 r = 0; r < n_skip; ++r)
      for (u32 i = skip_lo[r] + tid; i < skip_hi[r]; i += 1024) reach[i] = 0;
  } else {
    // (the first build of a stream WITHOUT size hints: no member's end is known, every candidate is an "exception" and the
    //  chain stops at candidate 0 -- nothing to double over; 0.29 ms of every such decode, profiles/r04_nobc_kernel_stats.md)
    if (nxt[0] >= K) {
      for (u32 i = tid; i <= K; i += 1024) reach[i] = (i == 0) ? 1u : 0u;
      return;
    }
    for (u32 i = tid; i <= K; i += 1024) { jmp[i] = nxt[i]; reach[i] = (i == 0) ? 1u : 0u; }
    if (tid == 0) jmp2[K] = K;
    __syncthreads();
    u32 *ja = jmp, *jb = jmp2;
    for (u64 span = 1; span < (u64)K; span <<= 1) {
      for (u32 i = tid; i < K; i += 1024) {
        u32 j = ja[i];
        if (reach[i] && j < K) reach[j] = 1;
        jb[i] = (j < K) ? ja[j] : K;
      }
      __syncthreads();
      u32 *t = ja; ja = jb; jb = t;
    }
  }
}
struct ChainPart { u64 members, out_bytes, in_bytes; };
AHIP_DEVINL void block_scan3_1024(u64 &xa, u64 &xb, u64 &xc, u64 &ta, u64 &tb, u64 &tc, u64 (*ws)[16]) {  // inclusive scans; t* = totals
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    u64 ya = __shfl_up(xa, o), yb = __shfl_up(xb, o), yc = __shfl_up(xc, o);
    if (lane >= o) { xa += ya; xb += yb; xc += yc; }
  }
  if (lane == 63) { ws[0][w] = xa; ws[1][w] = xb; ws[2][w] = xc; }
  __syncthreads();
  u64 oa = 0, ob = 0, oc = 0;
  ta = tb = tc = 0;
  for (int k = 0; k < 16; ++k) {
    if (k < w) { oa += ws[0][k]; ob += ws[1][k]; oc += ws[2][k]; }
    ta += ws[0][k]; tb += ws[1][k]; tc += ws[2][k];
  }
  xa += oa; xb += ob; xc += oc;
  __syncthreads();
}
__global__ __launch_bounds__(1024) void gz_chain_sums(const u64 *cand_pos, const GzHeader *hdr, u32 K, const u32 *reach,
                                                      ChainPart *part) {
  __shared__ u64 ws[3][16];
  const u32 i = blockIdx.x * 1024 + threadIdx.x;
  const bool on = i < K && reach[i];
  u64 xa = on ? 1 : 0, xb = on ? hdr[i].size : 0, xc = on ? (hdr[i].next_pos - cand_pos[i]) : 0;
  u64 ta, tb, tc;
  block_scan3_1024(xa, xb, xc, ta, tb, tc, ws);
  if (threadIdx.x == 0) part[blockIdx.x] = ChainPart{ta, tb, tc};
}
__global__ __launch_bounds__(1024) void gz_chain_emit(const u64 *cand_pos, const GzHeader *hdr, u32 K, u64 start, const u32 *nxt,
                                                      const u32 *reach, const ChainPart *part, MemberDesc *members,
                                                      u32 *expect_status, ChainSummary *sum, u32 *retok_ids, u32 retok_cap) {
  __shared__ u64 ws[3][16];
  const u32 tid = threadIdx.x;
  // what the workgroups in front of this one hold
  u64 pa = 0, pb = 0, pc = 0;
  for (u32 k = tid; k < blockIdx.x; k += 1024) { pa += part[k].members; pb += part[k].out_bytes; pc += part[k].in_bytes; }
  { u64 ta, tb, tc; block_scan3_1024(pa, pb, pc, ta, tb, tc, ws); pa = ta; pb = tb; pc = tc; }
  const u32 i = blockIdx.x * 1024 + tid;
  const bool on = i < K && reach[i];
  GzHeader h{};
  if (on) h = hdr[i];
  const u64 a = on ? 1 : 0, b = on ? h.size : 0, c = on ? (h.next_pos - cand_pos[i]) : 0;
  u64 xa = a, xb = b, xc = c, ta, tb, tc;
  block_scan3_1024(xa, xb, xc, ta, tb, tc, ws);
  if (on) {
    const u64 m = pa + xa - a;
    MemberDesc d;
    d.in_off = h.payload_off;
    d.out_off = pb + xb - b;
    d.out_limit = h.size;
    d.expect_end = (h.flags & HF_RANGE) ? POS_UNKNOWN : h.next_pos - 8;
    d.in_end = 0;
    d.hist = d.out_off < 32768 ? (u32)d.out_off : 32768u;  // every member appends to the same OutputStream (q8)
    d.pad = i;  // its candidate: where a sizing run left its tokens
    members[m] = d;
    expect_status[m] = h.status;
    if (h.flags & HF_RETOK) {  // listed (in no particular order) for the launch that tokenizes them again
      const u32 slot = atomicAdd(&sum->retok, 1u);
      if (slot < retok_cap) retok_ids[slot] = (u32)m;
    }
    if (nxt[i] == K) {  // last member of the chain
      sum->tail_pos = h.next_pos;
      if (!(h.flags & (HF_BC | HF_SIZED | HF_RANGE))) sum->stopped_unknown = 1;
    }
    if (h.flags & HF_RANGE) atomicOr(&sum->range_error, 1u);
    if ((h.flags & HF_SIZED) && h.status == MS_OVERSUB) atomicAdd(&sum->oversub, 1u);
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    const bool first_ok = K > 0 && cand_pos[0] == start;
    sum->members = pa + ta;
    sum->total_out = pb + tb;
    sum->payload_bytes = pc + tc;
    sum->first_is_gzip = first_ok ? 1u : 0u;
    if (!first_ok) sum->tail_pos = start;
  }
}

__global__ __launch_bounds__(256) void gz_gather_sizes(const u32 *ids, u32 count, const MemberDesc *members, u64 *sizes) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) sizes[i] = members[ids[i]].out_limit;
}

// Post-decode check of the trusted (BC/ISIZE) index and of per-member verdicts.
struct RunSummary {
  u32 mismatches;   // members whose size / end position / status differ from the index
  u32 worst_status; // max MS_* over members (MS_OK.. order is by severity for 0..3)
  u32 first_bad;    // index of the first mismatching member
  u32 any_range, any_hang, any_false;
};
__global__ __launch_bounds__(256) void gz_verify(const MemberDesc *members, const u32 *expect_status,
                                                 const MemberResult *res, u32 M, RunSummary *rs) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  MemberResult r = res[i];
  MemberDesc d = members[i];
  // (a source before the first byte of the whole output is a verdict of its own -- the sizing run, which does not
  //  know the output offsets yet, lets it pass)
  bool bad = r.status != MS_FARREF && (r.out_len != d.out_limit || r.status != expect_status[i] ||
                                       (d.expect_end != POS_UNKNOWN && r.end_pos != d.expect_end));
  if (bad) { atomicAdd(&rs->mismatches, 1u); atomicMin(&rs->first_bad, i); }
  if (r.status == MS_RANGE || r.status == MS_FARREF) atomicOr(&rs->any_range, 1u);  // FARREF: source before index 0
  if (r.status == MS_HANG) atomicOr(&rs->any_hang, 1u);
  if (r.status == MS_FALSE || r.status == MS_EOS) atomicOr(&rs->any_false, 1u);
}

}  // namespace ahip
