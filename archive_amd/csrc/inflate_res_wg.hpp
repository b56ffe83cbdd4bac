// inflate_res_wg.hpp -- the resolver as a WORKGROUP per member: several waves, ONE output window in LDS.
//
// The reference copies a back-reference out of "the output so far" (util/output_memory_stream.dart:79-98,
// inflate.dart:300-343).  resolve_member() (inflate_par.hpp) keeps 2.5 KB of that output per wave in LDS and asks global
// memory for everything farther back: 4.15e8 fetches of a 128-byte line for 6 - 16 bytes each, 53 GB per decode of the
// benchmark stream, two thirds of them L2 misses (profiles/r05_pmc_traffic.md) -- that kernel runs at 85 % of the
// fabric's bandwidth moving bytes nobody asked for.  Here DEFLATE's whole reach is in LDS:
//
//   ring      WG_RING bytes (32 KiB of history + the room the chunks in flight need) shared by the WG_WAVES waves of the
//             workgroup; byte p of the member lives at ring[(p + A) mod WG_RING], A = the output address mod 16, so that the
//             ring goes out to global memory in aligned 16-byte units.  A token's bytes are written LINEARLY from its first
//             byte's index: the one token per revolution that runs past the ring's end lands in a guard behind it and is
//             copied to the ring's start before its chunk is declared complete.
//   chunks    64 consecutive tokens, one per lane, exactly like resolve_member()'s; chunk j of a look at the directory
//             belongs to wave j mod WG_WAVES.  Tokens carry their output offsets (inflate_par.hpp, token store), so a
//             wave needs nothing from the chunks in front of its own to know where its bytes go.
//   frontier  one word in LDS: every byte in front of it is final.  Chunks COMPLETE in stream order -- the wave that
//             finishes chunk j moves the frontier to the chunk's end -- but their work does not wait for that:
//               at once     literals; matches whose source ends in front of the frontier as the wave last saw it
//                           (82 % of the matches of log text with four waves: tools/analysis/wg_resolver_model.c);
//               behind the chunk in front   the other matches, in rounds: the first one left is always ready, a later
//                           one when its source ends in front of the first one's destination (1.8 matches per chunk copy
//                           from inside their own chunk);
//               whole wave  a match that overlaps its own source, is longer than 32 bytes, or whose source runs across the
//                           ring's end: byte per lane, period doubling.
//   capacity  a chunk writes only while its end lies less than WG_RING - 32 KiB in front of the frontier -- the slots it
//             overwrites are then older than anything a chunk in flight may still read; a chunk larger than that (64
//             matches of 258 bytes are 16 KiB) goes token by token once it is the oldest.
//   flush     whoever completes a chunk hands every whole KiB in front of the new frontier to global memory with aligned
//             16-byte stores (nobody reads them back here: the history is the ring).
//
// Special directory entries (a run the serial decoder wrote: DF_BIG; a stored block: DF_STORED) are walked by the whole
// workgroup in lock step.  Members the fast path leaves to inflate_late_kernel (a reach into EARLIER members' output,
// quirk q8) never come here.
#pragma once
#include "inflate_par.hpp"

namespace ahip {

#ifndef AHIP_WG_WAVES
#define AHIP_WG_WAVES 4
#endif
#ifndef AHIP_WG_RING
#define AHIP_WG_RING 36864
#endif
constexpr u32 WG_WAVES = AHIP_WG_WAVES;
constexpr u32 WG_THREADS = WG_WAVES * 64;
constexpr u32 WG_RING = AHIP_WG_RING;
constexpr u32 WG_GUARD = 320;                     // >= 258 (a token written past the ring's end) and >= 32 (a read past it)
constexpr u32 WG_ALLOW = WG_RING - 32768 - 64;    // a chunk may write this far in front of the frontier
constexpr u32 WG_FLUSH = 1024;                    // bytes per flush block (64 lanes x 16)
static_assert(WG_RING % 16 == 0 && WG_RING >= 32768 + 1024 && WG_ALLOW >= 512 && WG_WAVES >= 1 && WG_WAVES <= 8, "ring geometry");

struct ResWgLds {
  u8 ring[WG_RING + WG_GUARD] __attribute__((aligned(16)));
  u32 rbits[LOOK_TOK / 32 + 4];  // the current look at the directory: bit i = token i of the look is the first of its run
  uint2 rtab[64];                //   run r of the look: {area offset of its token 0 - index of that token in the look, output offset (low 32 bits)}
  u32 rcount[64 + 4];            //   runs that begin in front of chunk j of the look
  u32 frontier;                  // low 32 bits of the member-relative position in front of which every byte is final
  u32 member;                    // the member the workgroup works on (handed from wave 0 to the others)
};

#ifndef AHIP_WG_SLEEP_N
#define AHIP_WG_SLEEP_N 2
#endif
#define AHIP_WG_SLEEP() __builtin_amdgcn_s_sleep(AHIP_WG_SLEEP_N)
AHIP_DEVINL u32 wg_poll(const u32 *p) { return uniform(AHIP_LDS_POLL(p)); }
AHIP_DEVINL u32 wg_ring_wrap(u32 x) { return x >= WG_RING ? x - WG_RING : x; }  // x in [0, 2 WG_RING)

struct WgCtx {
  ResWgLds *P;
  u8 *out_base;  // the member's first output byte
  u32 A;         // out_base mod 16
  int wave, lane;
  u32 *cyc;      // -DAHIP_PROFILE_RES: shader-clock cycles / 16 of this wave's phases (tools/kstats.py, AHIP_KSTATS_RES=wg)
                 //   0 look setup + barriers  1 gather + prep  2 waiting for room  3 literals + early copies
                 //   4 waiting for the chunk in front  5 late rounds  6 publish + flush  7 whole member
};
#ifdef AHIP_PROFILE_RES
#define WTICK(var) const u64 var = __builtin_amdgcn_s_memtime()
#define WACC(slot, t0, t1) x.cyc[slot] += (u32)(((t1) - (t0)) >> 4)
#else
#define WTICK(var) do { } while (0)
#define WACC(slot, t0, t1) do { } while (0)
#endif

// [q0, q1) of the q space (q = position + A, q0 a multiple of 16) from the ring to global memory: aligned 16-byte units by
// the 64 lanes, the units the member's first / last byte cuts short byte by byte.  `total` = the member's bytes so far + A
// is the end of what exists; nothing in front of q = A exists either.
AHIP_DEVINL void wg_flush_units(const WgCtx &x, u32 q0, u32 q1, u32 q_end) {
  u8 *g = x.out_base - x.A;  // q = 0
  const u32 b0 = q0 % WG_RING;  // (uniform: scalar arithmetic)
  for (u32 u = (u32)x.lane * 16; q0 + u < q1; u += 64 * 16) {
    const u32 q = q0 + u;
    const u32 ri = wg_ring_wrap(b0 + u % WG_RING);
    const bool full = q >= x.A && q + 16 <= q_end;
    if (full) *(uint4 *)(g + q) = *(const uint4 *)(x.P->ring + ri);
    if (AHIP_ANY_HINT(!full)) {  // the member's first and last unit
      if (!full) {
#pragma nounroll
        for (u32 b = 0; b < 16; ++b)
          if (q + b >= x.A && q + b < q_end) g[q + b] = x.P->ring[ri + b];
      }
    }
  }
}

// Exactly `len` (3 .. 32) bytes from ring[sidx ..) to ring[idx ..), source and destination apart by at least `len`, as
// two pieces that overlap on BOTH sides -- loaded where they are stored, no shift in between: [0, 8) and [len - 8, len)
// for 8 .. 16 bytes, [8, 16) and [len - 16, len - 8) on top for more; two dwords for 4 .. 7, a half-word and a byte for 3.
// What does not depend on WHICH matches are copied (the eight addresses, the length classes as lane masks) is worked out
// once per chunk (WgCopy): a chunk copies in up to four goes (early, three levels behind the chunk in front), and a go is
// sixteen DS instructions under four exec masks -- written out by hand: the compiler's version of the same (a branch and
// an exec save / restore per condition) was 136 instructions a go (tools/analysis/isa_regions.py).
struct WgCopy {
  u32 sp, spt, spl, sp4;  // LDS byte addresses of the source: first byte, len - 8, len - 16, len - 4
  u32 dp, dpt, dpl, dp4;  // ... of the destination
  u64 m16, m8, m4, m3;    // length classes (lanes): > 16, 8 .. 32, 4 .. 7, 3
};
#ifdef AHIP_HOST_EMU
AHIP_DEVINL u32 wg_lds_addr(const u8 *ring) { (void)ring; return 0; }
static u8 *wg_emu_ring;  // (one workgroup at a time in the emulation)
#else
AHIP_DEVINL u32 wg_lds_addr(const u8 *ring) { return (u32)(uintptr_t)(const __attribute__((address_space(3))) u8 *)ring; }
#endif
AHIP_DEVINL WgCopy wg_copy_prepare(u8 *ring, bool simple, u32 idx, u32 sidx, u32 len) {
  WgCopy k;
#ifdef AHIP_HOST_EMU
  wg_emu_ring = ring;
#endif
  const u32 base = wg_lds_addr(ring);
  k.sp = base + sidx;
  k.dp = base + idx;
  const u32 t8 = len >= 8 ? len - 8 : 0u;
  k.spt = k.sp + t8; k.dpt = k.dp + t8;
  k.spl = k.sp + len - 16; k.dpl = k.dp + len - 16;
  k.sp4 = k.sp + len - 4; k.dp4 = k.dp + len - 4;
  k.m16 = __ballot(simple && len > 16);
  k.m8 = __ballot(simple && len >= 8);
  k.m4 = __ballot(simple && len >= 4 && len < 8);
  k.m3 = __ballot(simple && len < 4);
  return k;
}
// the lanes of `act` (a wave mask)
AHIP_DEVINL void wg_copy_simple(const WgCopy &k, u64 act) {
#ifdef AHIP_HOST_EMU
  const int lane = wave_emu::lane;
  u8 *R = wg_emu_ring;
  auto cp = [&](u32 d, u32 s, u32 n) { u8 t[8]; memcpy(t, R + s, n); memcpy(R + d, t, n); };
  if ((act & k.m8) >> lane & 1) {
    u8 t0[8], t1[8], t2[8], t3[8];
    const bool l16 = (k.m16 >> lane) & 1;
    memcpy(t0, R + k.sp, 8); memcpy(t1, R + k.spt, 8);
    if (l16) { memcpy(t2, R + k.sp + 8, 8); memcpy(t3, R + k.spl, 8); }
    memcpy(R + k.dp, t0, 8); memcpy(R + k.dpt, t1, 8);
    if (l16) { memcpy(R + k.dp + 8, t2, 8); memcpy(R + k.dpl, t3, 8); }
  }
  if ((act & k.m4) >> lane & 1) { u8 t0[4], t1[4]; memcpy(t0, R + k.sp, 4); memcpy(t1, R + k.sp4, 4); memcpy(R + k.dp, t0, 4); memcpy(R + k.dp4, t1, 4); }
  if ((act & k.m3) >> lane & 1) cp(k.dp, k.sp, 3);
#else
  const u64 a8 = act & k.m8, a16 = act & k.m16, a4 = act & k.m4, a3 = act & k.m3;
  u64 sv, va, vb, vc, vd;
  u32 wa, wb, ha, qa;
  // (a class nobody of `act` is in costs a compare and a branch, not its DS instructions: a store occupies the LDS for 4 - 6
  //  cycles whatever its exec mask, and the goes behind the chunk in front have a handful of lanes)
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_cmp_eq_u64 %[a8], 0\n\t"
      "s_cbranch_scc1 1f\n\t"
      "s_mov_b64 exec, %[a8]\n\t"
      "ds_read_b64 %[va], %[sp]\n\t"
      "ds_read_b64 %[vb], %[spt]\n\t"
      "s_cmp_eq_u64 %[a16], 0\n\t"
      "s_cbranch_scc1 1f\n\t"
      "s_mov_b64 exec, %[a16]\n\t"
      "ds_read_b64 %[vc], %[sp] offset:8\n\t"
      "ds_read_b64 %[vd], %[spl]\n\t"
      "1:\n\t"
      "s_cmp_eq_u64 %[a4], 0\n\t"
      "s_cbranch_scc1 2f\n\t"
      "s_mov_b64 exec, %[a4]\n\t"
      "ds_read_b32 %[wa], %[sp]\n\t"
      "ds_read_b32 %[wb], %[sp4]\n\t"
      "2:\n\t"
      "s_cmp_eq_u64 %[a3], 0\n\t"
      "s_cbranch_scc1 3f\n\t"
      "s_mov_b64 exec, %[a3]\n\t"
      "ds_read_u16 %[ha], %[sp]\n\t"
      "ds_read_u8 %[qa], %[sp] offset:2\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "ds_write_b16 %[dp], %[ha]\n\t"
      "ds_write_b8 %[dp], %[qa] offset:2\n\t"
      "3:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_cmp_eq_u64 %[a4], 0\n\t"
      "s_cbranch_scc1 4f\n\t"
      "s_mov_b64 exec, %[a4]\n\t"
      "ds_write_b32 %[dp], %[wa]\n\t"
      "ds_write_b32 %[dp4], %[wb]\n\t"
      "4:\n\t"
      "s_cmp_eq_u64 %[a8], 0\n\t"
      "s_cbranch_scc1 5f\n\t"
      "s_mov_b64 exec, %[a8]\n\t"
      "ds_write_b64 %[dp], %[va]\n\t"
      "ds_write_b64 %[dpt], %[vb]\n\t"
      "s_cmp_eq_u64 %[a16], 0\n\t"
      "s_cbranch_scc1 5f\n\t"
      "s_mov_b64 exec, %[a16]\n\t"
      "ds_write_b64 %[dp], %[vc] offset:8\n\t"
      "ds_write_b64 %[dpl], %[vd]\n\t"
      "5:\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      : [sv] "=&s"(sv), [va] "=&v"(va), [vb] "=&v"(vb), [vc] "=&v"(vc), [vd] "=&v"(vd), [wa] "=&v"(wa), [wb] "=&v"(wb), [ha] "=&v"(ha),
        [qa] "=&v"(qa)
      : [a8] "s"(a8), [a16] "s"(a16), [a4] "s"(a4), [a3] "s"(a3), [sp] "v"(k.sp), [spt] "v"(k.spt), [spl] "v"(k.spl), [sp4] "v"(k.sp4),
        [dp] "v"(k.dp), [dpt] "v"(k.dpt), [dpl] "v"(k.dpl), [dp4] "v"(k.dp4)
      : "memory", "scc");
#endif
}

// One match by the whole wave, a byte per lane: destination linear from ring index `idx` (it may run into the guard),
// source `dist` back with the ring's wrap; a match that reads its own output is filled by period doubling.  Everything in
// front of its destination is final.
AHIP_DEVINL void wg_copy_wave(u8 *ring, u32 idx, u32 len, u32 dist, int lane) {
  const u32 n0 = dist < len ? dist : len;
  const u32 s0 = idx >= dist ? idx - dist : idx + WG_RING - dist;  // ring index of the first source byte
  for (u32 k = (u32)lane; k < n0; k += 64) ring[idx + k] = ring[wg_ring_wrap(s0 + k)];
  wave_sync();
  u32 filled = n0;  // a multiple of dist from here on
  while (filled < len) {
    const u32 n = filled < len - filled ? filled : len - filled;
    for (u32 k = (u32)lane; k < n; k += 64) ring[idx + filled + k] = ring[idx + k];
    wave_sync();
    filled += n;
  }
}

// a chunk of 64 tokens as a wave holds it
struct WgCk {
  u32 t, len, ob;     // token word, bytes, output position (low 32 bits, member-relative)
  bool inb;
  u32 nin;            // (uniform) lanes of the chunk
  u32 cstart, cend;   // (uniform) the chunk's output [cstart, cend)
};

// the frontier moves from `prev` to `pos`; the whole KiB blocks (of the q space) that END in (prev, pos] go out -- which
// blocks a step hands over follows from its own two positions, nothing shared
AHIP_DEVINL void wg_publish(const WgCtx &x, u32 prev, u32 pos) {
  ResWgLds &P = *x.P;
  wave_sync();
  if (x.lane == 0) AHIP_LDS_POST(&P.frontier, pos);
  wave_sync();
#ifndef AHIP_ABL_WG_NOFLUSH  // dev ablation (wrong bytes): what the flush costs
  const u32 q0 = (prev + x.A) & ~(WG_FLUSH - 1), q1 = (pos + x.A) & ~(WG_FLUSH - 1);
  if (q1 != q0) wg_flush_units(x, q0, q1, pos + x.A);
#endif
}

// A chunk whose output does not fit between the frontier and the ring's capacity: token by token, once it is the oldest.
AHIP_DEVINL void wg_chunk_big(const WgCtx &x, const WgCk &c) {
  ResWgLds &P = *x.P;
  while (wg_poll(&P.frontier) != c.cstart) __builtin_amdgcn_s_sleep(1);
  wave_sync();
  for (u32 j = 0; j < c.nin; ++j) {
    const u32 t = lane_bcast(c.t, (int)j), len = lane_bcast(c.len, (int)j), ob = lane_bcast(c.ob, (int)j);
    const u32 idx = (ob + x.A) % WG_RING;
    if (t & REC_LIT) {
      if (x.lane == 0) P.ring[idx] = (u8)t;
    } else {
      wg_copy_wave(P.ring, idx, len, (t & 0x7fffu) + 1u, x.lane);
      if (idx + len > WG_RING) {  // the token ran into the guard
        for (u32 k = (u32)x.lane; k < idx + len - WG_RING; k += 64) P.ring[k] = P.ring[WG_RING + k];
      }
    }
    wg_publish(x, ob, ob + len);
  }
}

// Returns false when its loop bound was reached (it cannot be: every round finishes at least the first match left).
AHIP_DEVINL bool wg_chunk(const WgCtx &x, const WgCk &c) {
  ResWgLds &P = *x.P;
  const int lane = x.lane;
  if (c.cend - c.cstart > WG_ALLOW) { wg_chunk_big(x, c); return true; }
  // ---- room: the slots this chunk overwrites must be older than what any chunk in flight may read ----
  AHIP_ASM_NOTE("WGN chunk room");
  WTICK(w_0);
  u32 Fs = wg_poll(&P.frontier);
  while ((i32)(c.cend - Fs) > (i32)WG_ALLOW) { __builtin_amdgcn_s_sleep(1); Fs = wg_poll(&P.frontier); }
  wave_sync();
  WTICK(w_1);
  WACC(2, w_0, w_1);
  AHIP_ASM_NOTE("WGN classify");
  const bool lit = (c.t & REC_LIT) != 0;
  const bool isM = c.inb && !lit;
  const u32 dist = (c.t & 0x7fffu) + 1u;
  const u32 cidx = (c.cstart + x.A) % WG_RING;                // (uniform)
  const u32 idx = wg_ring_wrap(cidx + (c.ob - c.cstart));     // ring index of the token's first byte
  const u32 sidx = idx >= dist ? idx - dist : idx + WG_RING - dist;
  if (c.inb && lit) P.ring[idx] = (u8)c.t;
  // what of the source is not the match's own output ends here
  const u32 need_end = c.ob - dist + (dist < c.len ? dist : c.len);
  const bool simple = isM && c.len <= 32 && dist >= c.len && sidx + c.len <= WG_RING;
  // The one token that runs past the ring's end is written linearly into the guard; its tail goes to the ring's start as
  // soon as it has been written (a later match of this very chunk may copy from it).
  const u64 overm = __ballot(c.inb && idx + c.len > WG_RING);
  auto fix_over = [&]() {
    const int j = __builtin_ctzll(overm);
    const u32 n = lane_bcast(idx + c.len, j) - WG_RING;
    wave_sync();
    for (u32 k = (u32)lane; k < n; k += 64) P.ring[k] = P.ring[WG_RING + k];
    wave_sync();
  };
  AHIP_ASM_NOTE("WGN early");
  // ---- at once: sources that end in front of the frontier ----
  const bool early = simple && (i32)(need_end - Fs) <= 0;
  const WgCopy cp = wg_copy_prepare(P.ring, simple, idx, sidx, c.len);
#ifndef AHIP_ABL_WG_NOEARLY  // dev ablation (wrong bytes): what the early copies cost
  wg_copy_simple(cp, __ballot(early));
#endif
  AHIP_ASM_NOTE("WGN levels");
  const bool late = isM && !early;
  u64 R = __ballot(late);
  if (overm & ~R) fix_over();  // (a match that went early)
  // ---- the rest behind the chunk in front, in LEVELS worked out before the wait: a match whose source ends in front of the
  //      chunk, or overlaps no destination of another match that is still to be written, is level 1; one that copies from a
  //      level-l match is level l + 1.  Three levels cover 97.6 % of the chunks of log text (wg_resolver_model.c); a chunk with
  //      more, or with a match the simple copy does not take, goes through the rounds below. ----
  u64 M1 = R, M2 = 0, M3 = 0;
  bool rounds = __ballot(late && !simple) != 0;
  if (!rounds) {
    const u64 Qi = __ballot(late && (i32)(need_end - c.cstart) > 0);  // sources that reach into the own chunk
    M1 &= ~Qi;
    for (u64 rem = Qi; rem; rem &= rem - 1) {
      const int i = __builtin_ctzll(rem);
      const u32 so_i = lane_bcast(c.ob - dist, i), ne_i = lane_bcast(need_end, i);
      const u64 ov = __ballot(late && lane < i && (i32)(c.ob - ne_i) < 0 && (i32)(c.ob + c.len - so_i) > 0);
      const u64 bit = 1ull << i;
      if (ov & M3) { rounds = true; break; }
      else if (ov & M2) M3 |= bit;
      else if (ov & M1) M2 |= bit;
      else M1 |= bit;
    }
  }
  WTICK(w_2);
  WACC(3, w_1, w_2);
  AHIP_ASM_NOTE("WGN wait");
#ifndef AHIP_ABL_WG_NOWAIT  // dev ablation (wrong bytes): no chunk waits for the one in front
  while (wg_poll(&P.frontier) != c.cstart) AHIP_WG_SLEEP();
#endif
  wave_sync();
  WTICK(w_3);
  WACC(4, w_2, w_3);
#ifdef AHIP_ABL_WG_NOLATE  // dev ablation (wrong bytes): what the late copies cost
  R = 0;
#endif
  AHIP_ASM_NOTE("WGN late");
  if (!rounds && R) {
    wg_copy_simple(cp, M1);
    if (overm & M1) fix_over();
    if (M2) {
      wave_sync();
      wg_copy_simple(cp, M2);
      if (overm & M2) fix_over();
      if (M3) {
        wave_sync();
        wg_copy_simple(cp, M3);
        if (overm & M3) fix_over();
      }
    }
    R = 0;
  }
  AHIP_ASM_NOTE("WGN rounds");
  for (u32 guard = 0; R && guard < 65; ++guard) {  // rounds: the first match left is always ready, a later one when its source ends in front of the first one's destination
    const int f = __builtin_ctzll(R);
    if (!lane_bcast((u32)simple, f)) {
      wg_copy_wave(P.ring, lane_bcast(idx, f), lane_bcast(c.len, f), lane_bcast(dist, f), lane);
      if ((overm >> f) & 1) fix_over();
      R &= R - 1;
      continue;
    }
    const u32 wo_f = lane_bcast(c.ob, f);
    const bool ready = ((R >> lane) & 1) && simple && (lane == f || (i32)(need_end - wo_f) <= 0);
    const u64 rm = __ballot(ready);
    wg_copy_simple(cp, rm);
    wave_sync();
    if (overm & rm) fix_over();
    R &= ~rm;
  }
  AHIP_ASM_NOTE("WGN publish");
  WTICK(w_4);
  WACC(5, w_3, w_4);
  wg_publish(x, c.cstart, c.cend);
  AHIP_ASM_NOTE("WGN chunk end");
  WTICK(w_5);
  WACC(6, w_4, w_5);
  return R == 0;
}

// One member by the WG_WAVES waves of a workgroup: token runs (area, dir) -> bytes at out_base.
// Returns false when a loop bound that cannot be reached was reached (the caller reports MS_INTERNAL).
AHIP_DEVINL bool resolve_member_wg(ResWgLds &P, const u8 *in, const u32 *area, const DirEnt *dir, u32 ndir, u8 *out_base, int wave,
                                   int lane, u32 *cyc = nullptr) {
  const WgCtx x{&P, out_base, (u32)((uintptr_t)out_base & 15u), wave, lane, cyc};
  WTICK(w_begin);
  const u32 tid = (u32)wave * 64 + (u32)lane;
  if (tid == 0) P.frontier = 0;
  __syncthreads();
  bool all_done = true;
  u32 pos = 0;  // (uniform, the same in every wave) the member's bytes in front of the current look
  u32 de = 0;
  // (the directory entries of a look are asked for one look ahead: every wave reads them, nothing else can start before)
  DirEnt dvn = (u32)lane < ndir ? dir[lane] : make_uint4(0u, 0u, 0u, 0u);
  while (de < ndir) {
    WTICK(w_l0);
    const u32 ei = de + (u32)lane;
    const bool have = ei < ndir;
    const DirEnt dv = dvn;
    const u64 special = __ballot(have && (dv.y & (DF_BIG | DF_STORED)) != 0);
    const u32 nleft = ndir - de < 64u ? ndir - de : 64u;
    const u32 nplain = special ? (u32)__builtin_ctzll(special) : nleft;
    if (nplain == 0) {
      dvn = de + 1 + (u32)lane < ndir ? dir[de + 1 + lane] : make_uint4(0u, 0u, 0u, 0u);
      // ---- a special entry: the whole workgroup in lock step (everything in front of it is complete) ----
      const u32 y = lane_bcast(dv.y, 0), a0 = lane_bcast(dv.x, 0);
      const u32 cnt = y & DF_CNT;
      if (y & DF_STORED) {  // stored block: input -> ring -> output, a piece at a time
        const u64 src = (u64)uniform(area[a0]) | ((u64)uniform(area[a0 + 1]) << 32);
        for (u32 off = 0; off < cnt; off += 2048) {
          const u32 n = cnt - off < 2048u ? cnt - off : 2048u;
          const u32 b0 = (pos + x.A) % WG_RING;
          for (u32 i = tid; i < n; i += WG_THREADS) P.ring[wg_ring_wrap(b0 + i)] = in[src + off + i];
          __syncthreads();
          if (wave == 0) wg_publish(x, pos, pos + n);
          pos += n;
          __syncthreads();
        }
      } else {  // a run whose `end` fields may wrap: lengths exact modulo 2^16, offsets from a prefix sum; wave 0 alone
        if (wave == 0) {
          u32 carry_end = 0, run = pos;
          for (u32 c0 = 0; c0 < cnt; c0 += 64) {
            WgCk c;
            c.inb = c0 + (u32)lane < cnt;
            c.t = c.inb ? area[a0 + c0 + lane] : 0u;
            const u32 end = c.t >> 16;
            u32 pe = lane_prev(end);
            pe = lane == 0 ? carry_end : pe;
            c.len = c.inb ? ((end - pe) & 0xffffu) : 0u;
            u32 tot;
            c.ob = run + wave_excl_sum(c.len, tot);
            c.nin = cnt - c0 < 64u ? cnt - c0 : 64u;
            c.cstart = run;
            c.cend = run + tot;
            carry_end = lane_bcast(end, 63);
            all_done &= wg_chunk(x, c);
            run += tot;
          }
        }
        __syncthreads();
        pos = wg_poll(&P.frontier);
        __syncthreads();
      }
      de += 1;
      continue;
    }
    // ---- ordinary runs: lane r of every wave holds run r of the look; wave 0 writes the look's tables ----
    const u32 rcnt = (u32)lane < nplain ? (dv.y & DF_CNT) : 0u;
    u32 total;
    const u32 ts = wave_excl_sum(rcnt, total);
    const u32 nlook = (u32)__popcll(__ballot((u32)lane < nplain && ts + rcnt <= LOOK_TOK));
    total = nlook < 64 ? lane_bcast(ts, (int)nlook) : total;
    dvn = de + nlook + (u32)lane < ndir ? dir[de + nlook + lane] : make_uint4(0u, 0u, 0u, 0u);
    if (wave == 0) {
      for (u32 i = lane; i < LOOK_TOK / 32 + 4; i += 64) P.rbits[i] = 0;
      wave_sync();
      if ((u32)lane < nlook) {
        P.rtab[lane] = make_uint2(dv.x - ts, dv.z);
        atomicOr(&P.rbits[ts >> 5], 1u << (ts & 31));
      }
      wave_sync();
      const u32 pc = (u32)__builtin_popcount(P.rbits[2 * lane]) + (u32)__builtin_popcount(P.rbits[2 * lane + 1]);
      u32 dummy;
      P.rcount[lane] = wave_excl_sum(pc, dummy);
    }
    __syncthreads();
    WTICK(w_l1);
    WACC(0, w_l0, w_l1);
    const u32 nch = (total + 63) / 64;
    struct Tok { u32 t, base, pend; bool first, inb; u32 nin; };
    auto gather = [&](u32 j) -> Tok {  // chunk j of the look
      Tok q;
      const u32 c0 = j * 64, idx = c0 + (u32)lane;
      q.inb = idx < total;
      q.nin = total - c0 < 64u ? total - c0 : 64u;
      const u64 bits = (u64)uniform(P.rbits[2 * j]) | ((u64)uniform(P.rbits[2 * j + 1]) << 32);
      const u32 gr = uniform(P.rcount[j]);
      const u32 r = gr + (u32)__popcll(bits & ((2ull << lane) - 1)) - 1u;  // (token 0 of the look is marked: never negative where it counts)
      q.first = (bits >> lane) & 1;
      const uint2 e = P.rtab[r & 63u];
      q.base = e.y;
      q.t = area[q.inb ? e.x + idx : 0u];
      // `end` of the token in front of the chunk (lane 0's predecessor when it is not the first of its run)
      // (every lane asks for the same word; nothing waits for it here -- prep() uses it a chunk later)
      q.pend = 0;
      if (j > 0 && !(bits & 1)) {
        const uint2 ep = P.rtab[(gr - 1u) & 63u];
        q.pend = area[ep.x + c0 - 1u];
      }
      return q;
    };
    auto prep = [&](const Tok &q) -> WgCk {
      WgCk c;
      c.t = q.t;
      c.inb = q.inb;
      c.nin = q.nin;
      const u32 end = q.t >> 16;
      u32 pe = lane_prev(end);
      pe = lane == 0 ? (q.pend >> 16) : pe;
      pe = q.first ? 0u : pe;
      c.len = q.inb ? end - pe : 0u;
      c.ob = q.base + pe;
      c.cstart = lane_bcast(c.ob, 0);
      c.cend = lane_bcast(c.ob + c.len, (int)q.nin - 1);
      return c;
    };
    const u32 W = WG_WAVES;
    if ((u32)wave < nch) {
      Tok t1 = gather((u32)wave);
      Tok t2 = t1;
      if ((u32)wave + W < nch) t2 = gather((u32)wave + W);
      for (u32 j = (u32)wave; j < nch; j += W) {
  AHIP_ASM_NOTE("WGN prep");
        WTICK(w_g0);
        const WgCk c = prep(t1);
  AHIP_ASM_NOTE("WGN gather");
        t1 = t2;
        if (j + 2 * W < nch) t2 = gather(j + 2 * W);
  AHIP_ASM_NOTE("WGN gather end");
        WTICK(w_g1);
        WACC(1, w_g0, w_g1);
        all_done &= wg_chunk(x, c);
      }
    }
    WTICK(w_l2);
    __syncthreads();
    pos = wg_poll(&P.frontier);
    __syncthreads();
    WTICK(w_l3);
    WACC(0, w_l2, w_l3);
    de += nlook;
  }
  // ---- what is left in the ring ----
  const u32 fq = (pos + x.A) & ~(WG_FLUSH - 1);
  if (wave == 0 && pos + x.A > fq) wg_flush_units(x, fq, pos + x.A, pos + x.A);
  __syncthreads();
  WTICK(w_end);
  WACC(7, w_begin, w_end);
  return all_done;   // (per wave: the caller reports any wave's failure)
}

}  // namespace ahip
