// wave_emu.hpp -- run the wave-level device code of archive_amd/csrc on the CPU (test infrastructure only).
//
// One wave64 = 64 host threads executing the same function; every cross-lane primitive of common.hpp (ballot,
// readlane, DPP shifts, bpermute, wave barrier) is an exchange through a shared slot array between two spinning
// barriers.  That is exact for code whose cross-lane operations sit in wave-uniform control flow (all 64 lanes reach
// them): the resolver (resolve_member and everything under it).  The tokenizer's decode steps call __any under a
// per-lane condition and cannot be run this way.
// Included by common.hpp instead of <hip/hip_runtime.h> when AHIP_HOST_EMU is defined (g++ only).
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
#define __device__
#define __forceinline__ inline
#define __shared__ static

namespace wave_emu {
constexpr int W = 64;
inline thread_local int lane = 0;
struct Ctx {
  std::atomic<int> arrived{0};
  std::atomic<int> phase{0};
  uint64_t slot[W];
};
inline Ctx ctx;
// A WORKGROUP of several waves (the workgroup-per-member resolver): every wave has a context of its own (a thread's
// `cur` points at its wave's), the workgroup barrier counts all of its threads.
inline thread_local Ctx *cur = &ctx;
inline thread_local int wave = 0;
inline void barrier_on(Ctx &c, int n) {
  const int ph = c.phase.load(std::memory_order_acquire);
  if (c.arrived.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
    c.arrived.store(0, std::memory_order_relaxed);
    c.phase.store(ph + 1, std::memory_order_release);
  } else {
    int spins = 0;
    while (c.phase.load(std::memory_order_acquire) == ph)
      if (++spins > 64) std::this_thread::yield();  // fewer cores than lanes
  }
}
inline void barrier() { barrier_on(*cur, W); }
inline Ctx wg_ctx;
inline int wg_threads = W;
inline void wg_barrier() { barrier_on(wg_ctx, wg_threads); }
// publish v, let f look at all 64 values, leave together
template <class F>
inline auto exchange(uint64_t v, F f) -> decltype(f((const uint64_t *)nullptr)) {
  cur->slot[lane] = v;
  barrier();
  auto r = f((const uint64_t *)cur->slot);
  barrier();
  return r;
}
}  // namespace wave_emu

static inline unsigned long long __ballot(bool p) {
  return wave_emu::exchange(p ? 1u : 0u, [](const uint64_t *s) { unsigned long long m = 0; for (int i = 0; i < 64; ++i) m |= (unsigned long long)(s[i] & 1) << i; return m; });
}
static inline int __any(bool p) { return __ballot(p) != 0; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i); return r; }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { wave_emu::barrier(); }
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }
static inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
static inline void __syncthreads() { wave_emu::wg_barrier(); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline int __builtin_amdgcn_readlane(int v, int src) {
  return wave_emu::exchange((uint32_t)v, [src](const uint64_t *s) { return (int)(uint32_t)s[src & 63]; });
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return __builtin_amdgcn_readlane(v, 0); }
static inline int __builtin_amdgcn_ds_bpermute(int addr, int v) {
  const int me = wave_emu::lane; (void)me;
  return wave_emu::exchange((uint32_t)v, [addr](const uint64_t *s) { return (int)(uint32_t)s[((unsigned)addr >> 2) & 63]; });
}
// the DPP controls common.hpp uses (gfx9 encodings); lanes without a source, or in a row the mask disables, keep `old`
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  (void)bank_mask; (void)bound_ctrl;
  const int l = wave_emu::lane;
  return wave_emu::exchange((uint32_t)src, [=](const uint64_t *s) {
    const int row = l >> 4, in_row = l & 15;
    if (!((row_mask >> row) & 1)) return old;
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; return in_row >= n ? (int)(uint32_t)s[l - n] : old; }  // row_shr:n
    if (ctrl == 0x138) return l >= 1 ? (int)(uint32_t)s[l - 1] : old;                                                        // wave_shr:1
    if (ctrl == 0x142) return row >= 1 ? (int)(uint32_t)s[row * 16 - 1] : old;                                              // row_bcast:15
    if (ctrl == 0x143) return row >= 2 ? (int)(uint32_t)s[31] : old;                                                        // row_bcast:31
    __builtin_trap();
  });
}
static inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned width) {
  off &= 31; width &= 31;
  return width ? (v >> off) & ((1u << width) - 1) : 0u;
}
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
  return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31));
}
static inline unsigned atomicMin(unsigned *p, unsigned v) {
  unsigned cur = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) { }
  return cur;
}
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_ACQ_REL); }
static inline unsigned atomicAnd(unsigned *p, unsigned v) { return __atomic_fetch_and(p, v, __ATOMIC_ACQ_REL); }
