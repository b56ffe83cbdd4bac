// archive_hip.hip -- kernels + C-ABI of libarchive_hip.so (see include/archive_hip.h).
// gfx950 only.  No CPU decode path exists in this library: without a GPU every entry point
// that would decode returns AHIP_E_DEVICE.
#include "../../include/archive_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <atomic>
#include <vector>

#include <chrono>
#include "common.hpp"
#include "gzip_index.hpp"
#include "bzip2_kernels.hpp"
#include "bzip2_chain.hpp"
#include "checksum_kernels.hpp"
#include "deflate_kernels.hpp"
#include "inflate_par.hpp"
#include "inflate_res_wg.hpp"
#include "sm_inflate.hpp"

using namespace ahip;

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
// One wave64 per member, one wave per workgroup: no workgroup barrier is ever needed.  The decode
// is split into two kernels so that each keeps a small LDS footprint (more resident waves hide
// the LDS/L2 latency this work is bound by):
//   inflate_tokenize_kernel  Huffman side: tables + bitstream window in LDS (10.7 KiB/wave) -> the
//                            member's token stream in device scratch
//   inflate_resolve_kernel   LZ77 side: token queue + output window in LDS (9.6 KiB/wave) -> bytes
constexpr int WAVES_PER_BLOCK = 1;
// minimum waves per SIMD the register allocation must admit (tuning knobs; 1 = no constraint)
#ifndef AHIP_RES_MIN_WAVES
#define AHIP_RES_MIN_WAVES 5  // <= 96 VGPRs: measured faster than the 105 the compiler takes when left alone (4 waves per SIMD)
#endif
#ifndef AHIP_TOK_MIN_WAVES
#define AHIP_TOK_MIN_WAVES 3
#endif

struct TokKernelLds {
  WaveLds w;
  TokLds p;
#ifdef AHIP_TOK_LDS_PAD  // dev: fewer resident waves, to measure what residency is worth
  u32 pad[AHIP_TOK_LDS_PAD / 4];
#endif
};

// Persistent workgroups: the grid is sized to what the runtime says is resident at once; members are handed out by a
// device counter (next_member), so a workgroup that starts late simply finds less to do.
//  tokens == nullptr: sizing run (end position, size and verdict only).
//  tokens / dir: the launch group's token areas and run directories (TokSink, tok_layout).
// members the fast kernels leave to inflate_late_kernel
AHIP_DEVINL bool member_is_late(const MemberResult &r) {
  return r.status == MS_TOKFULL || r.status == MS_OVERSUB || (r.blocks & MR_FAR);
}
AHIP_DEVINL TokSink member_sink(u32 *tokens, DirEnt *dir, u64 out_rel, u64 out_limit, u32 k) {
  TokSink sk{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false, false};
  if (tokens) {
    u64 toff, doff;
    tok_layout(out_rel, out_limit, k, toff, sk.col_cap, doff, sk.dir_cap);
    sk.area = tokens + toff;
    sk.dir = dir + doff;
  }
  return sk;
}
// A launch over a LIST of members (ids[k], ascending) instead of a range; rel[k] = where member ids[k]'s token area
// starts, counted in output bytes like tok_layout()'s out_rel.
struct MemberSel { const u32 *ids; const u64 *rel; };
AHIP_DEVINL u32 member_index(const MemberSel &sel, u32 first, u32 k) { return sel.ids ? uniform(sel.ids[k]) : first + k; }
// Token areas laid out along the input (tok_layout_in): `pos` = the K candidate positions of the stream.
// ways W > 1: candidate c keeps its tokens in buffer c % W (way_words token words, way_dirs directory entries each) and its area
// reaches to candidate c + W -- a `1f 8b 08` inside a member's compressed data is a candidate too (one per 16 MiB or so), and an
// area that ended there was full: 36 of the benchmark's 65 536 members without size hints were tokenized again in a launch pair
// of their own (0.73 ms of 18.5).  Same idea as SmBase::ways (sm_inflate.hpp).
struct InLayout { const u64 *pos; u32 K; u64 in_len; u32 ways; u64 way_words, way_dirs; };
AHIP_DEVINL void in_layout(const InLayout &lay, u32 c, u64 &toff, u32 &col_cap, u64 &doff, u32 &dir_cap) {
  const u32 W = lay.ways > 1 ? lay.ways : 1u;
  const u64 p0 = uniform64(lay.pos[c]);
  const u64 p1 = c + W < lay.K ? uniform64(lay.pos[c + W]) : lay.in_len;
  tok_layout_in(p0, p1 > p0 ? p1 - p0 : 0, c, toff, col_cap, doff, dir_cap);
  if (W > 1) { toff += (u64)(c % W) * lay.way_words; doff += (u64)(c % W) * lay.way_dirs; }
}
AHIP_DEVINL TokSink candidate_sink(u32 *tokens, DirEnt *dir, const InLayout &lay, u32 c) {
  TokSink sk{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false, true};
  u64 toff, doff;
  in_layout(lay, c, toff, sk.col_cap, doff, sk.dir_cap);
  sk.area = tokens + toff;
  sk.dir = dir + doff;
  return sk;
}
// Work is handed out by a counter (`next`), one member at a time in index order: a workgroup that only becomes
// resident late (the occupancy the runtime reports is not always what the hardware grants) simply finds less to do.
// KEEP: a sizing run that keeps its tokens, laid out along the input (InLayout).
AHIP_DEVINL u32 next_member(u32 *next, int lane) {
  u32 k = 0;
  if (lane == 0) k = atomicAdd(next, 1u);
  return uniform(k);  // lane 0's value
}
// KEEP: a sizing run that keeps its tokens, laid out along the input (InLayout).  (Two instances of the same code: the
// second one used to come out of the register allocator five VGPRs over its budget -- the per-lane constants of the
// table build, hoisted out of the member loop; they are pinned inside it now, build_decode_table -- and as ONE kernel
// with a run-time flag the common instance was 0.7 % slower.)
#ifdef AHIP_TOK_VGPR  // dev: a hard register budget (the launch bound is capped by what LDS admits)
#define AHIP_TOK_ATTR __attribute__((amdgpu_num_vgpr(AHIP_TOK_VGPR)))
#else
#define AHIP_TOK_ATTR
#endif
template <bool KEEP>
__global__ __launch_bounds__(64, AHIP_TOK_MIN_WAVES) AHIP_TOK_ATTR void inflate_tokenize_kernel(const u8 *__restrict__ in, u64 in_len,
                                                             const MemberDesc *__restrict__ members, u32 first_member,
                                                             u32 n_members, u32 *__restrict__ tokens, DirEnt *__restrict__ dir,
                                                             u64 group_out0, MemberResult *__restrict__ results,
                                                             u32 *__restrict__ late, InLayout lay, MemberSel sel) {
  __shared__ TokKernelLds lds;
  const int lane = threadIdx.x;
  for (u32 k = next_member(late + 1, lane); k < n_members; k = next_member(late + 1, lane)) {
    const u32 m = member_index(sel, first_member, k);
    MemberDesc d = members[m];
    d.in_off = uniform64(d.in_off);
    d.out_off = uniform64(d.out_off);
    d.out_limit = uniform64(d.out_limit);
    // header scratch lives in the upper part of the ring buffer (the staged header bytes use the first 640)
    static_assert(sizeof(HeaderLds) + 1024 <= sizeof(TokLds), "header scratch must fit behind the staged header");
    HeaderLds &hdr = *(HeaderLds *)((u8 *)lds.p.inbuf + 1024);
    const u64 lim = uniform64(d.in_end) ? uniform64(d.in_end) : in_len;
    const bool sizing = !tokens || KEEP;
    TokSink sk = KEEP ? candidate_sink(tokens, dir, lay, m)
                      : member_sink(tokens, dir, sel.ids ? uniform64(sel.rel[k]) : d.out_off - group_out0, d.out_limit, k);
    // keep the sink in scalar registers whichever layout made it
    sk.col_cap = uniform(sk.col_cap);
    sk.dir_cap = uniform(sk.dir_cap);
#ifdef AHIP_MEMBER_CYC
    const u64 t0 = __builtin_readcyclecounter();
#endif
    inflate_member<false, true>(lds.w, hdr, &lds.p, in, lim, d, (u8 *)nullptr, sk, results[m], lane);
#ifdef AHIP_MEMBER_CYC
    if (lane == 0) results[m].fallbacks = (u32)((__builtin_readcyclecounter() - t0) >> 4);
#endif
    // rare: inflate_late_kernel finishes these (a sizing run only cares about the ones whose size it does not know yet)
    if (lane == 0 && (sizing ? results[m].status == MS_OVERSUB : member_is_late(results[m]))) atomicAdd(late, 1u);
  }
}

// KEPT: the tokens are those a sizing run kept (InLayout, `sized`); members it could not serve are skipped.
template <bool KEPT>
__global__ __launch_bounds__(64, AHIP_RES_MIN_WAVES) void inflate_resolve_kernel(const u8 *__restrict__ in,
                                                            const MemberDesc *__restrict__ members, u32 first_member,
                                                            u32 n_members, u8 *out, const u32 *__restrict__ tokens,
                                                            const DirEnt *__restrict__ dir, u64 group_out0,
                                                            MemberResult *__restrict__ results, InLayout lay,
                                                            const MemberResult *__restrict__ sized, MemberSel sel,
                                                            u32 *__restrict__ next) {
  __shared__ ResLds lds;
#ifdef AHIP_RES_LDS_PAD  // dev: fewer resident waves, to measure what residency is worth
  __shared__ u32 res_pad[AHIP_RES_LDS_PAD / 4];
  if (n_members == 0xffffffffu) res_pad[threadIdx.x] = 1;  // (keeps the array allocated)
#endif
  const int lane = threadIdx.x;
  for (u32 k = next_member(next, lane); k < n_members; k = next_member(next, lane)) {
    const u32 m = member_index(sel, first_member, k);
    const u64 out_off = uniform64(members[m].out_off), out_limit = uniform64(members[m].out_limit);
    u32 ndir;
    u64 toff, doff;
    u32 cc, dc;
    if (KEPT) {  // the tokens of the sizing run
      const u32 c = uniform(members[m].pad);
      if (uniform(sized[c].status) != MS_OK || (uniform(sized[c].blocks) & MR_FAR)) continue;  // HF_RETOK: tokenized again, afterwards
      if (lane == 0) results[m] = sized[c];
      if (uniform64(members[m].in_off) >= lay.in_len) continue;  // a long member: decoded by many waves, elsewhere
      ndir = (u32)uniform64(sized[c].tok_words);
      in_layout(lay, c, toff, cc, doff, dc);
    } else {
      if (uniform(results[m].status) == MS_TOKFULL || uniform(results[m].status) == MS_OVERSUB || (uniform(results[m].blocks) & MR_FAR)) continue;  // inflate_late_kernel
      ndir = (u32)uniform64(results[m].tok_words);
      tok_layout(sel.ids ? uniform64(sel.rel[k]) : out_off - group_out0, out_limit, k, toff, cc, doff, dc);
    }
    u32 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!resolve_member(lds, in, tokens + toff, dir + doff, ndir, out + out_off, cyc, lane) && lane == 0) results[m].status = MS_INTERNAL;
#ifdef AHIP_PROFILE_RES
    if (lane == 0) for (int q = 0; q < 8; ++q) results[m].cyc[q] = cyc[q];
#endif
  }
}

// The resolver as a workgroup per member (inflate_res_wg.hpp): WG_WAVES waves share one LDS ring that holds DEFLATE's whole
// reach, so no back-reference goes to global memory.  Same arguments, same member hand-out (one counter step per
// WORKGROUP) and same skips as inflate_resolve_kernel, which stays for the late kernel's members and the A/B switch
// AHIP_RES_WG=0.
union ResWgKernelLds {
  ResWgLds wg;
  ResLds one;  // a member of 2 GiB and more (positions are 32-bit here): the one-wave resolver, wave 0 alone
};
// What a workgroup needs to know of a member before it can start: handed from wave 0 to the others through LDS.
struct WgMember {
  u32 k;           // index in the launch (>= n_members: nothing left)
  u32 m;           // member
  u32 ndir;        // runs in its directory; 0xffffffff: not for this kernel (late kernel, re-tokenized, long member)
  u32 big;         // produced >= 2 GiB
  u64 out_off, toff, doff;
};
// With four members per CU nothing hides a member's start-up -- the hand-out counter, its descriptor, its verdict, then
// its directory and its first tokens are four dependent round trips to memory --, so the hand-out runs TWO members ahead:
// while member k is resolved, wave 0 holds the counter's answer for the member after next and has the descriptor and the
// verdict of the next one on their way.
template <bool KEPT>
__global__ __launch_bounds__(WG_THREADS) void inflate_resolve_wg_kernel(const u8 *__restrict__ in,
                                                            const MemberDesc *__restrict__ members, u32 first_member,
                                                            u32 n_members, u8 *out, const u32 *__restrict__ tokens,
                                                            const DirEnt *__restrict__ dir, u64 group_out0,
                                                            MemberResult *__restrict__ results, InLayout lay,
                                                            const MemberResult *__restrict__ sized, MemberSel sel,
                                                            u32 *__restrict__ next) {
  __shared__ ResWgKernelLds lds;
  __shared__ WgMember hand;
  const int lane = threadIdx.x & 63, wave = (int)uniform(threadIdx.x >> 6);
  // raw loads of a member's start-up data (wave 0; every lane loads the same words) ...
  struct Raw { u32 k, m; u64 out_off, out_limit, in_off, tok_words, out_len; u32 status, blocks, pad; };
  auto ask = [&](u32 k) -> Raw {
    Raw r{};
    r.k = k;
    if (k >= n_members) return r;
    r.m = sel.ids ? sel.ids[k] : first_member + k;
    const MemberDesc &d = members[r.m];
    r.out_off = d.out_off; r.out_limit = d.out_limit; r.in_off = d.in_off; r.pad = d.pad;
    const MemberResult &v = KEPT ? sized[r.pad] : results[r.m];
    r.status = v.status; r.blocks = v.blocks; r.tok_words = v.tok_words; r.out_len = v.out_len;
    return r;
  };
  // ... and what follows from them
  auto post = [&](const Raw &r) {
    WgMember h{};
    h.k = uniform(r.k);
    if (h.k < n_members) {
      h.m = uniform(r.m);
      h.out_off = uniform64(r.out_off);
      const u64 out_limit = uniform64(r.out_limit);
      const u32 status = uniform(r.status), blocks = uniform(r.blocks);
      h.ndir = (u32)uniform64(r.tok_words);
      h.big = uniform64(r.out_len) >= (1ull << 31);
      u32 cc, dc;
      if (KEPT) {  // the tokens of the sizing run
        const u32 c = uniform(r.pad);
        if (status != MS_OK || (blocks & MR_FAR)) h.ndir = 0xffffffffu;  // HF_RETOK: tokenized again, afterwards
        else {
          if (lane == 0) results[h.m] = sized[c];
          if (uniform64(r.in_off) >= lay.in_len) h.ndir = 0xffffffffu;  // a long member: decoded by many waves, elsewhere
        }
        in_layout(lay, c, h.toff, cc, h.doff, dc);
      } else {
        if (status == MS_TOKFULL || status == MS_OVERSUB || (blocks & MR_FAR)) h.ndir = 0xffffffffu;  // inflate_late_kernel
        tok_layout(sel.ids ? uniform64(sel.rel[h.k]) : h.out_off - group_out0, out_limit, h.k, h.toff, cc, h.doff, dc);
      }
    }
    if (lane == 0) hand = h;
  };
  u32 k2 = 0;  // (thread 0) the counter's answer for the member after next
  Raw nxt{};
  if (wave == 0) {
    u32 k0 = 0, k1 = 0;
    if (lane == 0) { k0 = atomicAdd(next, 1u); k1 = atomicAdd(next, 1u); k2 = atomicAdd(next, 1u); }
    k0 = uniform(k0); k1 = uniform(k1);
    post(ask(k0));
    nxt = ask(k1);
  }
  __syncthreads();
  for (;;) {
    const WgMember h = hand;
    if (uniform(h.k) >= n_members) break;
    const u32 m = uniform(h.m), ndir = uniform(h.ndir);
    __syncthreads();  // (everybody has read the hand-out)
    if (ndir != 0xffffffffu) {
      const u64 out_off = uniform64(h.out_off), toff = uniform64(h.toff), doff = uniform64(h.doff);
      bool ok;
      if (uniform(h.big)) {
        ok = true;
        u32 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (wave == 0) ok = resolve_member(lds.one, in, tokens + toff, dir + doff, ndir, out + out_off, cyc, lane);
        __syncthreads();
      } else {
#ifdef AHIP_PROFILE_RES
        u32 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        ok = resolve_member_wg(lds.wg, in, tokens + toff, dir + doff, ndir, out + out_off, wave, lane, cyc);
        if (lane == 0) for (int q = 0; q < 8; ++q) if (q < 7 || wave == 0) atomicAdd(&results[m].cyc[q], cyc[q]);  // (phases summed over the waves)
#else
        ok = resolve_member_wg(lds.wg, in, tokens + toff, dir + doff, ndir, out + out_off, wave, lane);
#endif
      }
      if (!ok && lane == 0) results[m].status = MS_INTERNAL;
    }
    if (wave == 0) {
      post(nxt);                        // the next member's data has been here for a while
      const u32 kn = uniform(k2);
      if (lane == 0) k2 = atomicAdd(next, 1u);
      nxt = ask(kn);
    }
    __syncthreads();
  }
}

// What the fast kernels leave behind -- in member order, one wave, after everything else (normally nothing: the
// counter is zero and the kernel returns at once):
//   MR_FAR      a back-reference reaches into the output of EARLIER members (the reference's gzip decoder appends all
//               members to one OutputStream, so that is legal there, quirk q8): resolved now that those bytes exist;
//   MS_OVERSUB  over-subscribed code lengths: decoded with the reference's own overwritten table (`exact`, scratch);
//   MS_TOKFULL  token area / run directory overflow: decoded by the byte-writing serial decoder.
template <bool WRITE>
__global__ __launch_bounds__(64) void inflate_late_kernel(const u8 *__restrict__ in, u64 in_len,
                                                         const MemberDesc *__restrict__ members, u32 first_member,
                                                         u32 n_members, u8 *out, const u32 *__restrict__ tokens,
                                                         const DirEnt *__restrict__ dir, u64 group_out0,
                                                         MemberResult *__restrict__ results, const u32 *__restrict__ late,
                                                         u32 *__restrict__ exact, MemberSel sel) {
  if (*late == 0) return;
  __shared__ WaveLds lds;
  __shared__ HeaderLds hdr;
  __shared__ ResLds par;
  const int lane = threadIdx.x;
  // 64 members per look (one per lane, the next look's loads already in flight); the late ones of a look in order.
  // A sizing run writes no output, so its members are independent: there the looks are dealt out over the workgroups of
  // the launch (each with its own exact-table scratch); the decode proper is one workgroup walking all looks in order.
  const u32 look0 = WRITE ? 0u : blockIdx.x * 64u, look_step = WRITE ? 64u : gridDim.x * 64u;
  if (!WRITE) exact += (size_t)blockIdx.x * (2u * 32768u);
  u32 st_n = 0, bl_n = 0;
  auto idx_of = [&](u32 k) { return sel.ids ? sel.ids[k] : first_member + k; };
  if (look0 + (u32)lane < n_members) { st_n = results[idx_of(look0 + lane)].status; bl_n = results[idx_of(look0 + lane)].blocks; }
  for (u32 base = look0; base < n_members; base += look_step) {
    const u32 st_c = st_n, bl_c = bl_n;
    if (base + look_step + (u32)lane < n_members) {
      st_n = results[idx_of(base + look_step + lane)].status;
      bl_n = results[idx_of(base + look_step + lane)].blocks;
    }
    const bool have = base + (u32)lane < n_members;
    u64 todo = __ballot(have && (WRITE ? (st_c == MS_TOKFULL || st_c == MS_OVERSUB || (bl_c & MR_FAR)) : st_c == MS_OVERSUB));
    while (todo) {
      const int j = __builtin_ctzll(todo);
      todo &= todo - 1;
      const u32 k = base + (u32)j, m = member_index(sel, first_member, k);
      const u32 status = lane_bcast(st_c, j);
      MemberDesc d = members[m];
      d.in_off = uniform64(d.in_off);
      d.out_off = uniform64(d.out_off);
      d.out_limit = uniform64(d.out_limit);
      d.hist = uniform(d.hist);
      if (status == MS_TOKFULL || status == MS_OVERSUB) {
        const u64 lim = uniform64(d.in_end) ? uniform64(d.in_end) : in_len;
        inflate_member<WRITE, false>(lds, hdr, nullptr, in, lim, d, out, TokSink{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false, false}, results[m], lane,
                                     nullptr, exact);
        if (!WRITE && lane == 0) results[m].blocks |= MR_FAR;  // sized here: no tokens of it exist
      } else if (WRITE) {
        u64 toff, doff;
        u32 cc, dc;
        tok_layout(sel.ids ? uniform64(sel.rel[k]) : d.out_off - group_out0, d.out_limit, k, toff, cc, doff, dc);
        u32 cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!resolve_member(par, in, tokens + toff, dir + doff, (u32)uniform64(results[m].tok_words), out + d.out_off, cyc, lane) && lane == 0)
          results[m].status = MS_INTERNAL;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // the next late member may read these bytes
    }
  }
}

// The serial decoder alone (one lane-uniform symbol at a time): kept as the checked fallback of
// the parallel kernel and as an A/B baseline (AHIP_SERIAL=1).
template <bool WRITE>
__global__ __launch_bounds__(64) void inflate_members_serial_kernel(const u8 *__restrict__ in, u64 in_len,
                                                                   const MemberDesc *__restrict__ members,
                                                                   u32 n_members, u8 *out,
                                                                   MemberResult *__restrict__ results) {
  __shared__ WaveLds lds;
  __shared__ HeaderLds hdr;
  const int lane = threadIdx.x;
  const u32 m = blockIdx.x;
  if (m >= n_members) return;
  MemberDesc d = members[m];
  d.in_off = uniform64(d.in_off);
  d.out_off = uniform64(d.out_off);
  d.out_limit = uniform64(d.out_limit);
  const u64 lim = uniform64(d.in_end) ? uniform64(d.in_end) : in_len;
  inflate_member<WRITE, false>(lds, hdr, nullptr, in, lim, d, out, TokSink{nullptr, 0, nullptr, 0, 0, ~0u, 0, 0, false}, results[m], lane);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;
std::recursive_mutex g_mu;
bool g_inited = false;

int32_t fail(int32_t code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return fail(AHIP_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                   \
  } while (0)

// A copy the host waits for, ORDERED ON THE CALLER'S STREAM: worker contexts run on non-blocking streams, which the legacy
// null stream (plain hipMemcpy) does not synchronise with.
static inline hipError_t copy_on(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t st) {
  hipError_t e = hipMemcpyAsync(dst, src, n, kind, st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

// Device scratch with a process-wide free list: hipMalloc/hipFree cost far more than the index
// kernels they would serve, so released blocks are kept and handed to the next plan.
struct DevBlock { void *p; size_t cap; };
thread_local std::vector<DevBlock> g_pool;  // one per host thread: a thread works on ONE device (see Worker below)

// A worker thread (one device context, see Worker below) frees what its thread_local buffers still hold when it ends;
// the main thread's are left to the process exit (the HIP runtime may be gone by the time its TLS objects are destroyed).
thread_local bool tl_free_bufs_at_exit = false;
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() {
    if (p && tl_free_bufs_at_exit) (void)hipFree(p);
    p = nullptr; cap = 0;
  }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    release();
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); ++i)
      if (g_pool[i].cap >= n && (best < 0 || g_pool[i].cap < g_pool[best].cap)) best = i;
    if (best >= 0 && g_pool[best].cap <= 4 * n + 4096) {
      p = g_pool[best].p; cap = g_pool[best].cap;
      g_pool.erase(g_pool.begin() + best);
      return hipSuccess;
    }
    size_t want = n + (n >> 2) + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want; else p = nullptr;
    return e;
  }
  void release() {
    if (!p) return;
    if (g_pool.size() < 256) g_pool.push_back({p, cap}); else (void)hipFree(p);
    p = nullptr; cap = 0;
  }
  template <class T> T *as() const { return (T *)p; }
};

thread_local DevBuf g_scratch, g_tokens;
// What the token scratch of this thread holds: every (re)writer takes a fresh process-wide number, so a plan that kept
// the tokens of its sizing run can tell whether they are still there.
static std::atomic<u64> g_tok_gen_counter{0};
thread_local u64 g_tok_gen = 0;
hipError_t scratch_reserve(size_t bytes, void **p) {
  g_tok_gen = ++g_tok_gen_counter;
  hipError_t e = g_scratch.reserve(bytes);
  *p = g_scratch.p;
  return e;
}
hipError_t tokens_reserve(size_t bytes, void **p) {
  g_tok_gen = ++g_tok_gen_counter;
  hipError_t e = g_tokens.reserve(bytes);
  *p = g_tokens.p;
  return e;
}

inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// AHIP_RES_WG=1: the resolver as a workgroup per member with DEFLATE's whole reach in one LDS ring (inflate_res_wg.hpp) instead of
// the one-wave-per-member kernel.  Bit-exact and without a single far fetch (resolver reads 53 -> 5 GB per decode), but 10.6 ms
// against 7.8 on the benchmark stream: an unaligned DS access costs one LDS cycle per active lane, and with the history in LDS
// every match pays that twice (profiles/r06_experiments.md section 1).  Kept selectable, measured, not the default.
bool use_res_wg() {
  static int v = -1;
  if (v < 0) { const char *e = getenv("AHIP_RES_WG"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}
bool use_serial_kernel() {
  static int v = -1;
  if (v < 0) { const char *e = getenv("AHIP_SERIAL"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}

// device scratch: per-workgroup token slabs of the tokenizer, and the token streams handed to the resolver
struct DevBuf;
hipError_t scratch_reserve(size_t bytes, void **p);
hipError_t tokens_reserve(size_t bytes, void **p);

// Output bytes decoded per tokenize/resolve launch pair.  The token scratch takes 6 B per output byte (1.5 words) and
// the run directory 1 B (a 16-byte entry per 16 bytes): a 6 GiB group holds about 42 GiB of scratch.
// AHIP_GROUP_OUT_MAX (bytes) shrinks it so that tests can drive the multi-group path with small streams.
static u64 group_out_max() {
  static u64 v = 0;
  if (!v) {
    const char *e = getenv("AHIP_GROUP_OUT_MAX");
    v = e && atoll(e) > 0 ? (u64)atoll(e) : (6ull << 30);
  }
  return v;
}
#define GROUP_OUT_MAX group_out_max()

// members[first .. first+count) with output offsets [out0, out1): tokenize (+ resolve when WRITE)
static thread_local InLayout g_kept_lay{nullptr, 0, 0, 0, 0, 0};  // how the sizing run that kept its tokens laid them out
static thread_local int tok_resident = 0, res_resident = 0;  // workgroups of the tokenizer / resolver resident at once (per device context)
static thread_local DevBuf g_late, g_exact, g_tokens2, g_scratch2;
// The token / directory scratch is shared by every launch of the thread: a launch on another stream than the
// previous one first waits for that one to be done with it.
static thread_local hipEvent_t scratch_free = nullptr;
static thread_local hipStream_t scratch_user = nullptr;
static hipError_t scratch_acquire(hipStream_t st) {
  hipError_t e = hipSuccess;
  if (!scratch_free) { e = hipEventCreateWithFlags(&scratch_free, hipEventDisableTiming); if (e != hipSuccess) return e; scratch_user = st; }
  if (scratch_user != st) { e = hipStreamWaitEvent(st, scratch_free, 0); if (e != hipSuccess) return e; scratch_user = st; }
  return e;
}
// lay_pos (sizing runs only): the K candidate positions -- the run keeps its tokens, laid out along the input, and
// *gen_out names the scratch contents they are (0: not kept).
template <bool WRITE>
hipError_t launch_inflate_group(const u8 *in, u64 n, const MemberDesc *members, u32 first, u32 count, u64 out0, u64 out1,
                                u8 *out, MemberResult *res, hipStream_t st, const u64 *lay_pos = nullptr, u64 *gen_out = nullptr,
                                bool skip_late = false) {
  if (gen_out) *gen_out = 0;
  if (!tok_resident) {
    int dev = 0, cus = 0, a = 0, b = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, inflate_tokenize_kernel<false>, 64, 0);
    if (e != hipSuccess) return e;
    if (use_res_wg()) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, inflate_resolve_wg_kernel<false>, (int)WG_THREADS, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, inflate_resolve_kernel<false>, 64, 0);
    if (e != hipSuccess) return e;
    // (the number the runtime reports is sometimes one workgroup per CU more than the hardware grants -- MI355X_MICROARCH.md,
    // "Residency"; members are handed out by a counter, so the surplus workgroups just find nothing left to do)
    if (const char *e1 = getenv("AHIP_TOK_WGS_PER_CU")) a = atoi(e1);  // tuning overrides
    if (const char *e2 = getenv("AHIP_RES_WGS_PER_CU")) b = atoi(e2);
    tok_resident = cus * (a > 0 ? a : 1);
    res_resident = cus * (b > 0 ? b : 1);
  }
  const u32 grid1 = count < (u32)tok_resident ? count : (u32)tok_resident;
  void *tp = nullptr, *dp = nullptr;
  hipError_t e = hipSuccess;
  e = scratch_acquire(st);
  if (e != hipSuccess) return e;
  InLayout lay{nullptr, 0, n};
  if (!WRITE && lay_pos && first == 0 && !getenv("AHIP_NO_TOKEN_REUSE") && n <= (4ull << 30)) {
    // (12 B per input byte: when the device cannot spare that, the sizing run simply does not keep its tokens -- it needs
    //  none itself -- and the decode proper tokenizes again)
    const u64 way_words = ((u64)n * IN_R + (u64)count * IN_PAD + 64 + 15) & ~15ull, way_dirs = (u64)(n / 32) + (u64)count * 64 + 64;
    u32 ways = 1;
    for (u32 w : {4u, 2u}) if (ways == 1 && way_words * 4 * w <= (48ull << 30)) ways = w;  // (in_layout: buffers of areas)
    if (const char *ew = getenv("AHIP_IN_WAYS")) { const int v = atoi(ew); if (v == 1 || v == 2 || v == 4) ways = (u32)v; }  // (dev)
    for (;; ways >>= 1) {
      e = tokens_reserve((size_t)way_words * ways * 4, &tp);
      if (e == hipSuccess) e = scratch_reserve((size_t)way_dirs * ways * DIR_BYTES, &dp);
      if (e == hipSuccess || ways == 1) break;
      (void)hipGetLastError();
    }
    if (e == hipSuccess) {
      lay = InLayout{lay_pos, count, n, ways, way_words, way_dirs};
      g_kept_lay = lay;  // (what launch_resolve_kept lays the same tokens out by)
      if (gen_out) *gen_out = g_tok_gen;
    } else {
      (void)hipGetLastError();
      tp = nullptr; dp = nullptr;
    }
  }
  if (WRITE) {
    // 1.5 token words + 1/16 directory entry per output byte, plus a fixed allowance per member (tok_layout)
    const u64 span = out1 - out0;
    if (span > (1ull << 40)) return hipErrorInvalidValue;
    e = tokens_reserve(((size_t)(span * 3 / 2) + (size_t)count * 1024 + 64) * 4, &tp);
    if (e != hipSuccess) return e;
    e = scratch_reserve(((size_t)(span / 16) + (size_t)count * 64 + 64) * DIR_BYTES, &dp);
    if (e != hipSuccess) return e;
  }
  if (getenv("AHIP_DEBUG")) fprintf(stderr, "[ahip] inflate group first=%u count=%u grid=%u/%d out=%llu write=%d\n", first, count, grid1, res_resident, (unsigned long long)(out1 - out0), (int)WRITE);
  // the late list: a counter + the scratch of the exact (over-subscribed) tables
  DevBuf &dlate = g_late, &dexact = g_exact;
  e = dlate.reserve(64);
  if (e != hipSuccess) return e;
  constexpr u32 SIZING_LATE_WGS = 64;  // workgroups of a sizing run's late kernel
  e = dexact.reserve((size_t)2 * 32768 * 4 * (WRITE ? 1u : SIZING_LATE_WGS));
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(dlate.p, 0, 64, st);  // [0] late members, [1] / [2] the tokenizer's / resolver's next member
  if (e != hipSuccess) return e;
  // AHIP_KTIME=1 (dev): HIP events around the two kernels, printed per launch (the call then synchronises)
  static thread_local hipEvent_t kt[3] = {nullptr, nullptr, nullptr};
  const bool ktime = getenv("AHIP_KTIME") != nullptr;
  if (ktime) {
    for (auto &ev : kt) if (!ev) (void)hipEventCreate(&ev);
    (void)hipEventRecord(kt[0], st);
  }
  if (lay.pos)
    hipLaunchKernelGGL(inflate_tokenize_kernel<true>, dim3(grid1), dim3(64), 0, st, in, n, members, first, count, (u32 *)tp, (DirEnt *)dp,
                       out0, res, dlate.as<u32>(), lay, MemberSel{nullptr, nullptr});
  else
    hipLaunchKernelGGL(inflate_tokenize_kernel<false>, dim3(grid1), dim3(64), 0, st, in, n, members, first, count, (u32 *)tp, (DirEnt *)dp,
                       out0, res, dlate.as<u32>(), lay, MemberSel{nullptr, nullptr});
  if (ktime) (void)hipEventRecord(kt[1], st);
  if (WRITE) {
    const u32 grid2 = count < (u32)res_resident ? count : (u32)res_resident;
    if (use_res_wg())
      hipLaunchKernelGGL(inflate_resolve_wg_kernel<false>, dim3(grid2), dim3(WG_THREADS), 0, st, in, members, first, count, out,
                         (const u32 *)tp, (const DirEnt *)dp, out0, res, InLayout{nullptr, 0, n}, (const MemberResult *)nullptr,
                         MemberSel{nullptr, nullptr}, dlate.as<u32>() + 2);
    else
      hipLaunchKernelGGL(inflate_resolve_kernel<false>, dim3(grid2), dim3(64), 0, st, in, members, first, count, out,
                         (const u32 *)tp, (const DirEnt *)dp, out0, res, InLayout{nullptr, 0, n}, (const MemberResult *)nullptr,
                         MemberSel{nullptr, nullptr}, dlate.as<u32>() + 2);
  }
  if (ktime) {
    (void)hipEventRecord(kt[2], st);
    (void)hipEventSynchronize(kt[2]);
    float a = 0, b = 0;
    (void)hipEventElapsedTime(&a, kt[0], kt[1]);
    (void)hipEventElapsedTime(&b, kt[1], kt[2]);
    fprintf(stderr, "[ahip] ktime members %u: tokenize %.3f ms, resolve %.3f ms\n", count, a, b);
  }
  // (skip_late, a sizing run only: the over-subscribed candidates stay unsized.  Nearly all of them are FALSE candidates --
  //  `1f 8b 08` inside compressed data, whose garbage headers are over-subscribed more often than not -- that the member
  //  chain never reaches, and sizing them with the reference's exact tables was 1.07 ms of every decode of a stream without
  //  size hints, profiles/r04_nobc_kernel_stats.md.  The chain counts the unsized ones it does reach: plan_build.)
  if (!skip_late)
    hipLaunchKernelGGL(inflate_late_kernel<WRITE>, dim3(WRITE ? 1u : SIZING_LATE_WGS), dim3(64), 0, st, in, n, members, first, count, out, (const u32 *)tp,
                       (const DirEnt *)dp, out0, res, dlate.as<u32>(), dexact.as<u32>(), MemberSel{nullptr, nullptr});
  e = hipEventRecord(scratch_free, st);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

// The decode proper from the tokens a sizing run kept (launch_inflate_group<false> with lay_pos): every member's
// candidate (MemberDesc::pad) has its runs in the scratch, so only the resolver runs; results are the sizing run's.
hipError_t launch_resolve_kept(const u8 *in, u64 n, const MemberDesc *members, u32 M, u8 *out, MemberResult *res,
                               const u64 *cand_pos, u32 K, const MemberResult *sized, hipStream_t st) {
  hipError_t e = scratch_acquire(st);
  if (e != hipSuccess) return e;
  const u32 resident = res_resident > 0 ? (u32)res_resident : 4096u;
  const u32 grid = M < resident ? M : resident;
  e = g_late.reserve(64);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(g_late.p, 0, 64, st);
  if (e != hipSuccess) return e;
  if (use_res_wg())
    hipLaunchKernelGGL(inflate_resolve_wg_kernel<true>, dim3(grid), dim3(WG_THREADS), 0, st, in, members, 0u, M, out, (const u32 *)g_tokens.p,
                       (const DirEnt *)g_scratch.p, (u64)0, res, InLayout{cand_pos, K, n, g_kept_lay.ways, g_kept_lay.way_words, g_kept_lay.way_dirs}, sized, MemberSel{nullptr, nullptr},
                       g_late.as<u32>() + 2);
  else
    hipLaunchKernelGGL(inflate_resolve_kernel<true>, dim3(grid), dim3(64), 0, st, in, members, 0u, M, out, (const u32 *)g_tokens.p,
                       (const DirEnt *)g_scratch.p, (u64)0, res, InLayout{cand_pos, K, n, g_kept_lay.ways, g_kept_lay.way_words, g_kept_lay.way_dirs}, sized, MemberSel{nullptr, nullptr},
                       g_late.as<u32>() + 2);
  e = hipEventRecord(scratch_free, st);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}
// ... and the members whose kept tokens are no use (HF_RETOK; ids ascending, rel = running sum of their sizes, `span`
// its total): tokenize + resolve + late as usual, with token areas of their own so that the kept ones survive.
hipError_t launch_inflate_listed(const u8 *in, u64 n, const MemberDesc *members, const u32 *ids, const u64 *rel, u32 count,
                                 u64 span, u8 *out, MemberResult *res, hipStream_t st) {
  if (count == 0) return hipSuccess;
  hipError_t e = scratch_acquire(st);
  if (e != hipSuccess) return e;
  e = g_tokens2.reserve(((size_t)(span * 3 / 2) + (size_t)count * 1024 + 64) * 4);
  if (e != hipSuccess) return e;
  e = g_scratch2.reserve(((size_t)(span / 16) + (size_t)count * 64 + 64) * DIR_BYTES);
  if (e != hipSuccess) return e;
  e = g_late.reserve(64);
  if (e != hipSuccess) return e;
  e = g_exact.reserve(2 * 32768 * 4);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(g_late.p, 0, 64, st);
  if (e != hipSuccess) return e;
  const MemberSel sel{ids, rel};
  const u32 r1 = tok_resident > 0 ? (u32)tok_resident : 2048u, r2 = res_resident > 0 ? (u32)res_resident : 4096u;
  hipLaunchKernelGGL(inflate_tokenize_kernel<false>, dim3(count < r1 ? count : r1), dim3(64), 0, st, in, n, members, 0u, count,
                     g_tokens2.as<u32>(), g_scratch2.as<DirEnt>(), (u64)0, res, g_late.as<u32>(), InLayout{nullptr, 0, n}, sel);
  if (use_res_wg())
    hipLaunchKernelGGL(inflate_resolve_wg_kernel<false>, dim3(count < r2 ? count : r2), dim3(WG_THREADS), 0, st, in, members, 0u, count, out,
                       (const u32 *)g_tokens2.p, (const DirEnt *)g_scratch2.p, (u64)0, res, InLayout{nullptr, 0, n},
                       (const MemberResult *)nullptr, sel, g_late.as<u32>() + 2);
  else
    hipLaunchKernelGGL(inflate_resolve_kernel<false>, dim3(count < r2 ? count : r2), dim3(64), 0, st, in, members, 0u, count, out,
                       (const u32 *)g_tokens2.p, (const DirEnt *)g_scratch2.p, (u64)0, res, InLayout{nullptr, 0, n},
                       (const MemberResult *)nullptr, sel, g_late.as<u32>() + 2);
  hipLaunchKernelGGL(inflate_late_kernel<true>, dim3(1), dim3(64), 0, st, in, n, members, 0u, count, out, (const u32 *)g_tokens2.p,
                     (const DirEnt *)g_scratch2.p, (u64)0, res, g_late.as<u32>(), g_exact.as<u32>(), sel);
  e = hipEventRecord(scratch_free, st);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

// host_out_off: output offset of every member plus one trailing total (M + 1 entries), or nullptr when
// the whole range is known to be small / a sizing run.
// [m_begin, m_end): the members to launch (with host_out_off; default all M)
template <bool WRITE>
hipError_t launch_inflate(const u8 *in, u64 n, const MemberDesc *members, u32 M, u8 *out, MemberResult *res,
                          hipStream_t st, const u64 *host_out_off = nullptr, u64 total_out = 0, const u64 *lay_pos = nullptr,
                          u64 *gen_out = nullptr, u32 m_begin = 0, u32 m_end = 0xffffffffu, bool skip_late = false) {
  if (gen_out) *gen_out = 0;
  if (M == 0) return hipSuccess;
  if (m_end > M) m_end = M;
  if (m_begin >= m_end) return hipSuccess;
  if (use_serial_kernel()) {
    hipLaunchKernelGGL(inflate_members_serial_kernel<WRITE>, dim3(M), dim3(64), 0, st, in, n, members, M, out, res);
    return hipGetLastError();
  }
  if (!WRITE || !host_out_off) {
    const u64 total = host_out_off ? host_out_off[M] : total_out;
    return launch_inflate_group<WRITE>(in, n, members, 0, M, 0, total, out, res, st, lay_pos, gen_out, skip_late);
  }
  u32 first = m_begin;
  while (first < m_end) {
    u32 last = first + 1;
    while (last < m_end && host_out_off[last + 1] - host_out_off[first] <= GROUP_OUT_MAX) ++last;
    hipError_t e = launch_inflate_group<WRITE>(in, n, members, first, last - first, host_out_off[first], host_out_off[last],
                                               out, res, st);
    if (e != hipSuccess) return e;
    first = last;
  }
  return hipSuccess;
}

void stop_workers();  // defined with the worker threads (multi-device section)
int32_t ensure_init() {
  if (g_inited) return AHIP_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(AHIP_E_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count=0"));
  g_inited = true;
  return AHIP_OK;
}

}  // namespace

// A plan = the member index of one gzip stream, resident in device memory.
struct ahip_gzip_plan {
  const u8 *d_in = nullptr;
  u64 in_len = 0;
  bool sized = false;  // sizes come from a sizing run (exact) rather than BC/ISIZE (trusted, verified)
  u64 tok_gen = 0;     // ... which kept its tokens in the scratch, as contents number tok_gen (0: it did not)
  u32 retok_n = 0;     //     except for retok_n members of the chain (ids / running sizes in retok_ids / retok_rel)
  u64 retok_span = 0;
  u32 K = 0;           // candidates
  bool cands_ready = false;  // cand_pos / hdr hold this stream's candidates (a rebuild with sizes from the data keeps them)
  bool size_oversub = false; // the sizing run also sizes over-subscribed candidates (inflate_late_kernel<false>): only once the chain met one
  ChainSummary sum{};
  DevBuf tile_counts, tile_offsets, tile_slots, cand_pos, hdr, scratch_u32, members, expect_status, results, sizing_descs,
      sizing_results, dsum, drun, retok_ids, retok_rel, chain_aux, tile_recs, cand_rec;
  bool ran = false;
  hipStream_t run_stream = nullptr;  // the stream the last ahip_gzip_plan_run was enqueued on
  std::vector<u64> host_out_off;  // M + 1 entries: output offset of every member, then the total
  struct Big { u32 cand, member; u64 in_off, out_off, out_len; bool reaches; };  // reaches: into the output of earlier members (q8)
  std::vector<Big> big;           // long members decoded by many waves each (sm_inflate), outside the member launch
  ~ahip_gzip_plan() {
    for (DevBuf *b : {&tile_counts, &tile_offsets, &tile_slots, &cand_pos, &hdr, &scratch_u32, &members, &expect_status, &results,
                      &sizing_descs, &sizing_results, &dsum, &drun, &retok_ids, &retok_rel, &chain_aux, &tile_recs, &cand_rec})
      b->release();
  }
};

namespace {

// hist0: bytes of EARLIER output (in front of d_out) the stream's back-references may reach -- what the gzip members before
// this one appended to the shared OutputStream (quirk q8); 0 for a stream with an output of its own.  A result with
// MR_REACH / MR_FAR in `blocks` did reach into them.
int32_t sm_inflate(const u8 *d_in, u64 n, u64 off, u8 *d_out, u64 out_cap, bool write, MemberResult *res, bool *handled,
                   hipStream_t st, u32 hist0 = 0);
int32_t inflate_one_wave(const u8 *d_in, u64 n, u64 off, u8 *d_out, u64 out_cap, bool write, MemberResult *res, hipStream_t st, u32 hist0 = 0);
u64 sm_min_bytes();

// Build (or rebuild with force_sizing) the member index for d_in[start..n).
int32_t plan_build(ahip_gzip_plan *pl, bool force_sizing, hipStream_t st) {
  const u8 *in = pl->d_in;
  const u64 n = pl->in_len, start = 0;
  pl->sized = false;
  pl->tok_gen = 0;
  pl->retok_n = 0;
  pl->retok_span = 0;
  pl->sum = ChainSummary{};
  HIP_TRY(pl->dsum.reserve(sizeof(ChainSummary)));
  HIP_TRY(hipMemsetAsync(pl->dsum.p, 0, sizeof(ChainSummary), st));
  if (n < 3) {
    pl->K = 0;
    pl->sum.tail_pos = start;
    return AHIP_OK;
  }
  u32 K = pl->K;
  if (!(force_sizing && pl->cands_ready)) {  // (the candidates and their parsed headers do not depend on where sizes come from)
    pl->K = 0;
    const u32 tiles = cdiv(n - start, TILE_BYTES);
    HIP_TRY(pl->tile_counts.reserve((size_t)tiles * 4 + 4));
    HIP_TRY(pl->tile_offsets.reserve((size_t)tiles * 4 + 4));
    u32 *d_total = pl->tile_offsets.as<u32>() + tiles;
    HIP_TRY(pl->tile_slots.reserve((size_t)tiles * TILE_SLOTS * 2 + 16));
    HIP_TRY(pl->tile_recs.reserve((size_t)tiles * TILE_SLOTS * sizeof(CandRec) + 16));
    hipLaunchKernelGGL(gz_count_candidates, dim3(tiles), dim3(256), 0, st, in, start, n, pl->tile_counts.as<u32>(), pl->tile_slots.as<u16>(),
                       pl->tile_recs.as<CandRec>());
    hipLaunchKernelGGL(scan_exclusive_u32, dim3(1), dim3(1024), 0, st, pl->tile_counts.as<u32>(),
                       pl->tile_offsets.as<u32>(), (u64)tiles, d_total);
    K = 0;
    HIP_TRY(hipMemcpyAsync(&K, d_total, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    pl->K = K;
    if (K == 0) {
      pl->sum.tail_pos = start;
      return AHIP_OK;
    }
    HIP_TRY(pl->cand_pos.reserve((size_t)K * 8));
    HIP_TRY(pl->hdr.reserve((size_t)K * sizeof(GzHeader)));
    HIP_TRY(pl->cand_rec.reserve((size_t)K * sizeof(CandRec) + 16));
    hipLaunchKernelGGL(gz_write_candidates, dim3(tiles + (tiles + 255) / 256), dim3(256), 0, st, in, start, n, pl->tile_counts.as<u32>(),
                       pl->tile_offsets.as<u32>(), pl->cand_pos.as<u64>(), pl->tile_slots.as<u16>(), tiles, pl->tile_recs.as<CandRec>(),
                       pl->cand_rec.as<CandRec>());
    hipLaunchKernelGGL(gz_parse_headers, dim3(cdiv(K, 256)), dim3(256), 0, st, in, n, pl->cand_pos.as<u64>(), K,
                       pl->hdr.as<GzHeader>(), pl->dsum.as<ChainSummary>(), pl->cand_rec.as<CandRec>());
    pl->cands_ready = true;
  }
  if (force_sizing) {
    // sizing run over every candidate: exact end position, size and verdict, no stores
    HIP_TRY(pl->sizing_descs.reserve((size_t)K * sizeof(MemberDesc)));
    HIP_TRY(pl->sizing_results.reserve((size_t)K * sizeof(MemberResult)));
    hipLaunchKernelGGL(gz_make_sizing_descs, dim3(cdiv(K, 256)), dim3(256), 0, st, pl->hdr.as<GzHeader>(), K,
                       pl->sizing_descs.as<MemberDesc>());
    // Long members (a file that is ONE gzip member is the common case) are measured by many waves each, in stream
    // order, BEFORE the all-candidates launch: every `1f 8b 08` inside a measured member is a false candidate (about
    // one per 16 MiB of compressed data) and is skipped together with the member itself (descriptor pointed at the
    // end of the input).  A candidate looks long when the next candidate is >= sm_min_bytes away; once eight have
    // been measured the all-members launch (one wave each, all at once) wins unless a member is really long.
    pl->big.clear();
    std::vector<std::pair<u32, MemberResult>> measured;
    bool nothing_left = false;
    if (K <= (1u << 20) && n >= sm_min_bytes() && !getenv("AHIP_NO_SM")) {
      std::vector<GzHeader> hh(K);
      std::vector<u64> cp(K);
      std::vector<MemberDesc> sd(K);
      HIP_TRY(hipMemcpyAsync(hh.data(), pl->hdr.p, (size_t)K * sizeof(GzHeader), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(cp.data(), pl->cand_pos.p, (size_t)K * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(sd.data(), pl->sizing_descs.p, (size_t)K * sizeof(MemberDesc), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      u64 covered_until = 0;
      bool changed = false;
      for (u32 i = 0; i < K; ++i) {
        if (cp[i] < covered_until) { sd[i].in_off = n; changed = true; continue; }  // inside a measured member
        const u64 lim = i + 1 < K ? cp[i + 1] : n;
        if ((hh[i].flags & HF_RANGE) || hh[i].payload_off >= lim) continue;
        const u64 gap = lim - hh[i].payload_off;
        if (gap < sm_min_bytes() || (measured.size() >= 8 && gap < (32ull << 20))) continue;
        MemberResult r{};
        bool handled = false;
        // (output offsets are not known yet: every member but the stream's first may reach 32 KiB back, like the sizing
        //  run of the ordinary members; the decode proper passes the exact figure)
        const u32 hist0 = (i == 0 && cp[0] == start) ? 0u : 32768u;
        int32_t rc = sm_inflate(in, n, hh[i].payload_off, nullptr, 0, false, &r, &handled, st, hist0);
        if (rc != AHIP_OK) return rc;
        if (!handled) { rc = inflate_one_wave(in, n, hh[i].payload_off, nullptr, ~0ull, false, &r, st, hist0); if (rc != AHIP_OK) return rc; }
        const bool reaches = (r.blocks & (MR_REACH | MR_FAR)) != 0;
        r.blocks &= ~(MR_REACH | MR_FAR);  // (of no concern to the member index: the long members are decoded apart)
        measured.push_back({i, r});
        pl->big.push_back({i, 0xffffffffu, hh[i].payload_off, 0, 0, reaches});
        sd[i].in_off = n;  // nothing to read: the member launch is done with it at once
        changed = true;
        if (r.status == MS_OK) covered_until = r.end_pos;
      }
      if (changed)
        HIP_TRY(hipMemcpyAsync(pl->sizing_descs.p, sd.data(), (size_t)K * sizeof(MemberDesc), hipMemcpyHostToDevice, st));
      nothing_left = true;
      for (u32 i = 0; i < K; ++i) nothing_left = nothing_left && sd[i].in_off >= n;
    }
    // (when the long members cover every candidate the launch below sizes nothing: it then must not take the token
    //  scratch either -- the chunked path has just kept ITS tokens there for the decode proper)
    // (the late kernel -- exact tables for over-subscribed candidates -- only when a first build found one ON the chain)
    HIP_TRY(launch_inflate<false>(in, n, pl->sizing_descs.as<MemberDesc>(), K, (u8 *)nullptr,
                                  pl->sizing_results.as<MemberResult>(), st, nullptr, 0, nothing_left ? nullptr : pl->cand_pos.as<u64>(), &pl->tok_gen,
                                  0, 0xffffffffu, !pl->size_oversub));
    for (auto &mr : measured)
      HIP_TRY(hipMemcpyAsync(pl->sizing_results.as<MemberResult>() + mr.first, &mr.second, sizeof(MemberResult), hipMemcpyHostToDevice, st));
    if (!measured.empty()) HIP_TRY(hipStreamSynchronize(st));
    if (getenv("AHIP_DEBUG_SIZING")) {  // what the candidates turned out to be (and, in -DAHIP_MEMBER_CYC builds, what they cost)
      std::vector<MemberResult> rr(K);
      std::vector<u64> cp(K);
      HIP_TRY(hipMemcpyAsync(rr.data(), pl->sizing_results.p, (size_t)K * sizeof(MemberResult), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(cp.data(), pl->cand_pos.p, (size_t)K * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      u32 hist[16] = {0};
      std::vector<u32> order(K);
      for (u32 i = 0; i < K; ++i) { hist[rr[i].status & 15]++; order[i] = i; }
      std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return rr[a].fallbacks > rr[b].fallbacks; });
      fprintf(stderr, "[ahip] sizing: K=%u status histogram", K);
      for (int q = 0; q < 16; ++q) if (hist[q]) fprintf(stderr, " %d:%u", q, hist[q]);
      fprintf(stderr, "\n");
      for (u32 j = 0; j < K && j < 12; ++j) {
        const MemberResult &r = rr[order[j]];
        fprintf(stderr, "[ahip]   cand %u at %llu: cyc/16 %u status %u blocks %u(0x%x) out %llu end %llu runs %llu\n", order[j],
                (unsigned long long)cp[order[j]], r.fallbacks, r.status, r.blocks & 0xffffff, r.blocks >> 24, (unsigned long long)r.out_len,
                (unsigned long long)r.end_pos, (unsigned long long)r.tok_words);
      }
      if (K > 24) { const MemberResult &r = rr[order[K / 2]]; fprintf(stderr, "[ahip]   median cyc/16 %u out %llu\n", r.fallbacks, (unsigned long long)r.out_len); }
    }
    hipLaunchKernelGGL(gz_apply_sizing, dim3(cdiv(K, 256)), dim3(256), 0, st, pl->hdr.as<GzHeader>(), K,
                       pl->sizing_results.as<MemberResult>(), n);
    pl->sized = true;
  }
  constexpr u32 RETOK_CAP = 4096;
  HIP_TRY(pl->retok_ids.reserve((size_t)RETOK_CAP * 4));
  HIP_TRY(pl->scratch_u32.reserve((size_t)(K + 1) * 4 * 4));
  HIP_TRY(pl->members.reserve((size_t)K * sizeof(MemberDesc)));
  HIP_TRY(pl->expect_status.reserve((size_t)K * 4));
  u32 *s = pl->scratch_u32.as<u32>();
  {
    // the chain (gzip_index.hpp): successors by all workgroups, the exceptions by one, sums and the member list by all
    const u32 parts = cdiv(K, 1024);
    HIP_TRY(pl->chain_aux.reserve(16 + (size_t)EXC_CAP * sizeof(ChainExc) + (size_t)parts * sizeof(ChainPart)));
    u32 *n_exc = pl->chain_aux.as<u32>();
    ChainExc *exc = (ChainExc *)(pl->chain_aux.as<u8>() + 16);
    ChainPart *part = (ChainPart *)(pl->chain_aux.as<u8>() + 16 + (size_t)EXC_CAP * sizeof(ChainExc));
    u32 *nxt = s, *jmp = s + (K + 1), *jmp2 = s + 2 * (size_t)(K + 1), *reach = s + 3 * (size_t)(K + 1);
    HIP_TRY(hipMemsetAsync(n_exc, 0, 16, st));
    hipLaunchKernelGGL(gz_link, dim3(cdiv((u64)K + 1, 256)), dim3(256), 0, st, pl->cand_pos.as<u64>(), pl->hdr.as<GzHeader>(), K, n, nxt, reach,
                       exc, n_exc);
    hipLaunchKernelGGL(gz_chain_fix, dim3(1), dim3(1024), 0, st, pl->cand_pos.as<u64>(), K, start, nxt, jmp, jmp2, reach, exc, n_exc);
    hipLaunchKernelGGL(gz_chain_sums, dim3(parts), dim3(1024), 0, st, pl->cand_pos.as<u64>(), pl->hdr.as<GzHeader>(), K, reach, part);
    hipLaunchKernelGGL(gz_chain_emit, dim3(parts), dim3(1024), 0, st, pl->cand_pos.as<u64>(), pl->hdr.as<GzHeader>(), K, start, nxt, reach,
                       part, pl->members.as<MemberDesc>(), pl->expect_status.as<u32>(), pl->dsum.as<ChainSummary>(),
                       pl->retok_ids.as<u32>(), RETOK_CAP);
  }
  HIP_TRY(hipMemcpyAsync(&pl->sum, pl->dsum.p, sizeof(ChainSummary), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  if (pl->sized && pl->sum.oversub && !pl->size_oversub) {
    // an over-subscribed member ON the chain (no encoder writes one; the reference decodes it with its overwritten table):
    // its size and everything behind it are not known -- size once more, the late kernel in the launch this time
    pl->size_oversub = true;
    return plan_build(pl, true, st);
  }
  if (pl->tok_gen && pl->sum.retok > RETOK_CAP) pl->tok_gen = 0;  // too many to list: the decode tokenizes everything again
  if (pl->tok_gen && pl->sum.retok) {
    // the members tokenized again: ascending, with the running sum of their sizes (where their token areas go)
    const u32 R = pl->sum.retok;
    std::vector<u32> ids(R);
    std::vector<u64> sz(R), rel(R);
    HIP_TRY(hipMemcpyAsync(ids.data(), pl->retok_ids.p, (size_t)R * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::sort(ids.begin(), ids.end());
    HIP_TRY(hipMemcpyAsync(pl->retok_ids.p, ids.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(pl->retok_rel.reserve((size_t)R * 8));
    hipLaunchKernelGGL(gz_gather_sizes, dim3(cdiv(R, 256)), dim3(256), 0, st, pl->retok_ids.as<u32>(), R, pl->members.as<MemberDesc>(),
                       pl->retok_rel.as<u64>());
    HIP_TRY(hipMemcpyAsync(sz.data(), pl->retok_rel.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    u64 acc = 0;
    for (u32 i = 0; i < R; ++i) { rel[i] = acc; acc += sz[i]; if (acc > (1ull << 40)) break; }
    if (acc > (1ull << 40)) pl->tok_gen = 0;
    else {
      HIP_TRY(hipMemcpyAsync(pl->retok_rel.p, rel.data(), (size_t)R * 8, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));  // rel / ids are host vectors about to go out of scope
      pl->retok_n = R;
      pl->retok_span = acc;
    }
  }
  if (!pl->sum.first_is_gzip) { pl->sum.members = 0; pl->sum.total_out = 0; pl->sum.tail_pos = start; }
  if (force_sizing && !pl->big.empty() && pl->sum.members) {
    // which members are the long ones (false candidates among them never reach the chain); neutralise those
    // descriptors for the member launch and remember where their output goes
    std::vector<MemberDesc> md(pl->sum.members);
    HIP_TRY(copy_on(md.data(), pl->members.p, md.size() * sizeof(MemberDesc), hipMemcpyDeviceToHost, st));
    std::vector<ahip_gzip_plan::Big> keep;
    for (auto bg : pl->big)
      for (size_t m = 0; m < md.size(); ++m)
        if (md[m].in_off == bg.in_off) {
          bg.member = (u32)m; bg.out_off = md[m].out_off; bg.out_len = md[m].out_limit;
          keep.push_back(bg);
          md[m].in_off = n;
          HIP_TRY(copy_on(pl->members.as<MemberDesc>() + m, &md[m], sizeof(MemberDesc), hipMemcpyHostToDevice, st));
          break;
        }
    pl->big = keep;
  } else if (!force_sizing) {
    pl->big.clear();
  }
  // output offsets on the host: the decode is launched in groups whose token streams fit the scratch
  pl->host_out_off.clear();
  bool stream_order = false;  // a long member reaches into earlier output: plan_run decodes range by range
  for (const auto &bg : pl->big) stream_order = stream_order || bg.reaches;
  if (pl->sum.members && (pl->sum.total_out > GROUP_OUT_MAX || stream_order)) {
    std::vector<MemberDesc> md(pl->sum.members);
    HIP_TRY(copy_on(md.data(), pl->members.p, md.size() * sizeof(MemberDesc), hipMemcpyDeviceToHost, st));
    pl->host_out_off.resize(md.size() + 1);
    for (size_t i = 0; i < md.size(); ++i) pl->host_out_off[i] = md[i].out_off;
    pl->host_out_off[md.size()] = pl->sum.total_out;
  }
  // Candidates without a BC subfield only matter when the member chain actually walks into one
  // (false candidates inside compressed data never do): then sizes have to come from the data.
  if (!force_sizing && pl->sum.first_is_gzip && pl->sum.stopped_unknown) return plan_build(pl, true, st);
  return AHIP_OK;
}

int32_t plan_run(ahip_gzip_plan *pl, u8 *d_out, size_t out_cap, hipStream_t st) {
  const u32 M = (u32)pl->sum.members;
  if (pl->sum.total_out > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
  HIP_TRY(pl->drun.reserve(sizeof(RunSummary)));
  RunSummary init{};
  init.first_bad = 0xffffffffu;
  HIP_TRY(hipMemcpyAsync(pl->drun.p, &init, sizeof init, hipMemcpyHostToDevice, st));
  pl->ran = true;
  pl->run_stream = st;
  if (M == 0) return AHIP_OK;
  HIP_TRY(pl->results.reserve((size_t)M * sizeof(MemberResult)));
  const u64 whole[2] = {0, pl->sum.total_out};  // one group: only its total is needed
  if (getenv("AHIP_DEBUG")) fprintf(stderr, "[ahip] plan_run: sized=%d kept tokens %llu (scratch holds %llu) retok=%u\n", (int)pl->sized,
                                    (unsigned long long)pl->tok_gen, (unsigned long long)g_tok_gen, pl->sum.retok);
  // Long members: many waves each, straight into place, with the bytes of earlier members in front of them as history
  // (quirk q8: the reference's members append to one OutputStream, _gzip_decoder_web.dart:27-41).  Normally they come
  // BEFORE the member launch, because its late kernel resolves back-references into earlier members' output in stream
  // order at its very end -- the bytes of a long member in front of such a member have to exist by then.  A long member
  // that itself reaches into earlier output (the sizing run noticed) needs everything in front of IT first: the stream is
  // then decoded in stream order, range of ordinary members / long member / range / ...
  std::vector<MemberResult> big_res(pl->big.size());
  bool any_reach = false;
  for (const auto &bg : pl->big) any_reach = any_reach || bg.reaches;
  const bool kept = pl->sized && pl->tok_gen && pl->tok_gen == g_tok_gen && !use_serial_kernel() && !any_reach;
  auto run_big = [&](size_t b) -> int32_t {
    const auto &bg = pl->big[b];
    const u32 hist0 = (u32)std::min<u64>(bg.out_off, 32768);
    bool handled = false;
    int32_t rc = sm_inflate(pl->d_in, pl->in_len, bg.in_off, d_out + bg.out_off, bg.out_len, true, &big_res[b], &handled, st, hist0);
    if (rc != AHIP_OK) return rc;
    if (!handled) rc = inflate_one_wave(pl->d_in, pl->in_len, bg.in_off, d_out + bg.out_off, bg.out_len, true, &big_res[b], st, hist0);
    big_res[b].blocks &= ~(MR_REACH | MR_FAR);
    return rc;
  };
  if (any_reach) {
    if (pl->host_out_off.size() != (size_t)M + 1) return fail(AHIP_E_DEVICE, "internal: member offsets missing for a stream-order decode");
    for (size_t b = 0; b < pl->big.size(); ++b)
      if (!pl->big[b].reaches) { int32_t rc = run_big(b); if (rc != AHIP_OK) return rc; }
    std::vector<size_t> order;
    for (size_t b = 0; b < pl->big.size(); ++b) if (pl->big[b].reaches) order.push_back(b);
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return pl->big[x].member < pl->big[y].member; });
    u32 first = 0;
    for (size_t b : order) {
      const u32 m = pl->big[b].member;  // (its own, neutralised, descriptor rides along with the range behind it)
      if (m > first) HIP_TRY(launch_inflate<true>(pl->d_in, pl->in_len, pl->members.as<MemberDesc>(), M, d_out, pl->results.as<MemberResult>(), st,
                                                  pl->host_out_off.data(), whole[1], nullptr, nullptr, first, m));
      int32_t rc = run_big(b);
      if (rc != AHIP_OK) return rc;
      first = m;
    }
    HIP_TRY(launch_inflate<true>(pl->d_in, pl->in_len, pl->members.as<MemberDesc>(), M, d_out, pl->results.as<MemberResult>(), st,
                                 pl->host_out_off.data(), whole[1], nullptr, nullptr, first, M));
  } else {
  for (size_t b = 0; b < pl->big.size() && !kept; ++b) {
    int32_t rc = run_big(b);
    if (rc != AHIP_OK) return rc;
  }
  if (kept) {
    // the sizing run's tokens are still in the scratch: resolve them, then tokenize + resolve the few members whose
    // tokens are no use (a full area -- a false candidate cut it short --, a reach into earlier members, an error).
    // (The long members use the same scratch: they follow, and a member that reaches into one of them is among the
    // few -- those are launched last.)
    HIP_TRY(launch_resolve_kept(pl->d_in, pl->in_len, pl->members.as<MemberDesc>(), M, d_out, pl->results.as<MemberResult>(),
                                pl->cand_pos.as<u64>(), pl->K, pl->sizing_results.as<MemberResult>(), st));
    for (size_t b = 0; b < pl->big.size(); ++b) {
      int32_t rc = run_big(b);
      if (rc != AHIP_OK) return rc;
    }
    HIP_TRY(launch_inflate_listed(pl->d_in, pl->in_len, pl->members.as<MemberDesc>(), pl->retok_ids.as<u32>(), pl->retok_rel.as<u64>(),
                                  pl->retok_n, pl->retok_span, d_out, pl->results.as<MemberResult>(), st));
  } else {
    HIP_TRY(launch_inflate<true>(pl->d_in, pl->in_len, pl->members.as<MemberDesc>(), M, d_out,
                                 pl->results.as<MemberResult>(), st,
                                 pl->host_out_off.empty() ? nullptr : pl->host_out_off.data(), whole[1]));
  }
  }
  for (size_t b = 0; b < pl->big.size(); ++b)
    HIP_TRY(hipMemcpyAsync(pl->results.as<MemberResult>() + pl->big[b].member, &big_res[b], sizeof(MemberResult), hipMemcpyHostToDevice, st));
  if (!pl->big.empty()) HIP_TRY(hipStreamSynchronize(st));  // (big_res lives on this stack frame)
  hipLaunchKernelGGL(gz_verify, dim3(cdiv(M, 256)), dim3(256), 0, st, pl->members.as<MemberDesc>(),
                     pl->expect_status.as<u32>(), pl->results.as<MemberResult>(), M, pl->drun.as<RunSummary>());
  HIP_TRY(hipGetLastError());
  return AHIP_OK;
}

// verdict of the member chain alone (the tail, if any, is the caller's business)
//  returns AHIP_* ; *needs_sizing set when the trusted index was wrong
int32_t plan_verdict(ahip_gzip_plan *pl, hipStream_t st, bool *needs_sizing) {
  *needs_sizing = false;
  RunSummary rs{};
  HIP_TRY(hipMemcpyAsync(&rs, pl->drun.p, sizeof rs, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (rs.mismatches) {
    if (getenv("AHIP_DEBUG") && rs.first_bad < pl->sum.members) {
      MemberResult r{}; MemberDesc d{}; u32 ex = 0;
      (void)copy_on(&r, pl->results.as<MemberResult>() + rs.first_bad, sizeof r, hipMemcpyDeviceToHost, st);
      (void)copy_on(&d, pl->members.as<MemberDesc>() + rs.first_bad, sizeof d, hipMemcpyDeviceToHost, st);
      (void)copy_on(&ex, pl->expect_status.as<u32>() + rs.first_bad, 4, hipMemcpyDeviceToHost, st);
      fprintf(stderr, "[ahip] verify: %u mismatches, first member %u: status %u (expected %u) out_len %llu (limit %llu) end %llu (expected %llu) blocks %x runs %llu hist %u sized=%d\n",
              rs.mismatches, rs.first_bad, r.status, ex, (unsigned long long)r.out_len, (unsigned long long)d.out_limit,
              (unsigned long long)r.end_pos, (unsigned long long)d.expect_end, r.blocks, (unsigned long long)r.tok_words, d.hist, (int)pl->sized);
    }
    if (!pl->sized) { *needs_sizing = true; return AHIP_OK; }
    return fail(AHIP_E_DEVICE, "internal: decode disagrees with its own sizing run");
  }
  if (rs.any_range) return AHIP_RANGE;  // incl. a back-reference to before the first byte of the whole output
  if (rs.any_hang) return AHIP_HANG;
  return AHIP_OK;
}

// Inflate one raw stream that starts at d_in[off]; output appended at d_out (device), window [0, cap).
// Runs a sizing pass first when `cap` may be too small.  Single wave: the serial path of the
// reference has no member parallelism to offer.
struct OneResult { MemberResult r; };
// ---- checksums of device-resident data (checksum_kernels.hpp) ----
static u32 g_ck_tables[CK_TAB_WORDS];
static std::once_flag g_ck_once;
static hipError_t ck_prepare(DevBuf &dtab, DevBuf &dacc) {
  std::call_once(g_ck_once, [] { ck_build_tables(g_ck_tables); });
  hipError_t e = dacc.reserve(64);
  if (e != hipSuccess) return e;
  if (!dtab.p) {
    e = dtab.reserve(sizeof(g_ck_tables));
    if (e != hipSuccess) return e;
    e = hipMemcpy(dtab.p, g_ck_tables, sizeof(g_ck_tables), hipMemcpyHostToDevice);
  }
  return e;
}
static u32 ck_grid(size_t n) {
  const size_t nseg = (n + CK_SEG - 1) / CK_SEG;
  const size_t waves = nseg < 8192 ? nseg : 8192;
  return (u32)((waves + 3) / 4 ? (waves + 3) / 4 : 1);
}
// getCrc32(d[0, n), crc0)  (util/crc32.dart:6-27)
static int32_t crc32_device_impl(const u8 *d, size_t n, u32 crc0, u32 *out, hipStream_t st) {
  static thread_local DevBuf dtab, dacc;
  HIP_TRY(ck_prepare(dtab, dacc));
  u32 raw = 0;
  if (n) {
    HIP_TRY(hipMemsetAsync(dacc.p, 0, 4, st));
    hipLaunchKernelGGL(crc32_kernel, dim3(ck_grid(n)), dim3(256), 0, st, d, (u64)n, dtab.as<u32>(), dacc.as<u32>());
    HIP_TRY(hipMemcpyAsync(&raw, dacc.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
  }
  // the register starts at ~crc0, runs over n bytes (x^(8n)) and is inverted at the end
  const u32 state = ck_mulmod(ck_xpow8(n, g_ck_tables + 1280), ~crc0) ^ raw;
  *out = ~state;
  return AHIP_OK;
}
// getAdler32(d[0, n), adler0)  (util/adler32.dart:29-52)
static int32_t adler32_device_impl(const u8 *d, size_t n, u32 adler0, u32 *out, hipStream_t st) {
  static thread_local DevBuf dtab, dacc;
  HIP_TRY(ck_prepare(dtab, dacc));
  unsigned long long acc[2] = {0, 0};
  if (n) {
    HIP_TRY(hipMemsetAsync(dacc.p, 0, 16, st));
    hipLaunchKernelGGL(adler32_kernel, dim3(ck_grid(n)), dim3(256), 0, st, d, (u64)n, dacc.as<unsigned long long>());
    HIP_TRY(hipMemcpyAsync(acc, dacc.p, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
  }
  const u64 M = 65521, a0 = adler0 & 0xffff, b0 = adler0 >> 16;
  const u64 A = acc[0] % M, T = acc[1] % M, nm = (u64)n % M;
  const u64 s1 = (a0 + A) % M;
  const u64 s2 = (b0 + nm * a0 % M + nm * A % M + M - T) % M;
  *out = (u32)((s2 << 16) | s1);
  return AHIP_OK;
}

// ---- one long stream on many waves (sm_inflate.hpp) ----
static u32 sm_translate_blocks() {  // workgroups per chunk of sm_translate_kernel (AHIP_SM_TBLOCKS: tuning)
  static u32 v = 0;
  if (!v) { const char *e = getenv("AHIP_SM_TBLOCKS"); v = e && atoi(e) > 0 ? (u32)atoi(e) : 4u; }
  return v;
}
u64 sm_min_bytes() {
  static u64 v = 0;
  if (!v) { const char *e = getenv("AHIP_SM_MIN"); v = e && atoll(e) > 0 ? (u64)atoll(e) : (2ull << 20); }
  return v;
}
static u64 sm_chunk_bytes() {
  static u64 v = 0;
  if (!v) { const char *e = getenv("AHIP_SM_CHUNK"); v = e && atoll(e) >= 4096 ? (u64)atoll(e) : (48ull << 10); }  // 48 KiB: measured best of 32..96 KiB on 256 MiB of text
  return v;
}
struct SmPlan {  // what the sizing pass learned, kept for the write pass of the same stream
  const u8 *d_in = nullptr;
  u64 n = 0, off = 0;
  std::vector<u64> cand;        // every candidate block start (bits), sorted
  std::vector<ChunkDesc> chain; // the chunks on the true path, exact offsets
  std::vector<MemberResult> sized;
  u64 total_out = 0, end_pos = 0;
  u32 blocks = 0;
  u32 hist0 = 0;               // the bytes of earlier output the stream was sized with (see sm_inflate)
  bool valid = false;
  u64 tok_gen = 0;             // the sizing pass kept its tokens (along the input) as scratch contents number tok_gen (0: it did not)
  std::vector<u32> chain_cand; // candidate index of every chunk of the chain
  SmBase lay{0, 0, 0, 0, 0, 0}; // how the kept tokens are laid out (sm_layout_in)
};
static thread_local SmPlan g_sm;
static thread_local int32_t g_last_chunks = 0;  // chunks of this thread's last long stream (ahip_debug_last_chunks)

// AHIP_SM_COUNTER=0: the chunk kernels take their chunks by the grid's stride instead of a device counter
static bool sm_use_counter() {
  static int v = -1;
  if (v < 0) { const char *e = getenv("AHIP_SM_COUNTER"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}
// a zeroed counter for the next chunk-kernel launch on `st` (nullptr: stride)
static thread_local DevBuf g_sm_ctr;
static hipError_t sm_counter(hipStream_t st, u32 slot, u32 **out) {
  *out = nullptr;
  if (!sm_use_counter()) return hipSuccess;
  hipError_t e = g_sm_ctr.reserve(256);
  if (e != hipSuccess) return e;
  *out = g_sm_ctr.as<u32>() + 16 * slot;  // (a slot per launch site: a launch never shares its counter with one still in flight)
  return hipMemsetAsync(*out, 0, 4, st);
}
static int sm_resident_waves() {
  static int v = 0;
  if (!v) {
    int dev = 0, cus = 0, a = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 1024;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, sm_tokenize_kernel, 64, 0) != hipSuccess || a < 1) a = 8;
    v = cus * (a > 1 ? a - 1 : a);
  }
  return v;
}

// *handled = false: not a case for this path (too short, no block found, chain broken, an error inside a chunk) --
// the caller decodes the stream on one wave, which restates the reference exactly.
int32_t sm_inflate(const u8 *d_in, u64 n, u64 off, u8 *d_out, u64 out_cap, bool write, MemberResult *res, bool *handled,
                   hipStream_t st, u32 hist0) {
  *handled = false;
  if (write) g_last_chunks = 0;
  if (getenv("AHIP_NO_SM") || n <= off || n - off < sm_min_bytes()) return AHIP_OK;
  const auto t_start = std::chrono::steady_clock::now();
  static thread_local DevBuf dcand, dchunks, dres, dsym, dwin;
  const bool dbg = getenv("AHIP_DEBUG") != nullptr;
  // the plan of a sizing call serves exactly one following write call on the same stream (device buffers are reused
  // between API calls, so a pointer match alone proves nothing)
  if (!write || !(g_sm.valid && g_sm.d_in == d_in && g_sm.n == n && g_sm.off == off)) {
    g_sm = SmPlan{};
    u64 cb = sm_chunk_bytes();
    // (48 KiB is the best cut for a member of a few hundred MiB -- one round of serial chunk times; a member of gigabytes is several
    //  rounds whatever the cut, and larger chunks save the finder's header checks and the window chain: 2 GiB 20.5 ms with 48 KiB
    //  cuts, 18.6 with 96.  About 8 000 chunks, between 48 and 128 KiB; AHIP_SM_CHUNK overrides)
    if (!getenv("AHIP_SM_CHUNK")) { const u64 want = ((n - off) / 8192 + 4095) & ~4095ull; cb = want < cb ? cb : (want > (128ull << 10) ? (128ull << 10) : want); }
    while ((n - off + cb - 1) / cb > 32768) cb *= 2;  // grid.y of the per-chunk kernels; 32 Ki chunks are plenty
    const u32 n_chunks = (u32)((n - off + cb - 1) / cb);
    if (n_chunks < 4) return AHIP_OK;
    static const u32 SPLIT = [] { const char *e = getenv("AHIP_SM_SPLIT"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? (u32)v : 4u; }();  // waves searching behind each cut
    HIP_TRY(dcand.reserve((size_t)n_chunks * SPLIT * 8 * 4 + (size_t)n_chunks * 4));  // (x 4: -DAHIP_PROFILE builds leave three cycle counts per workgroup behind the finds)
    // behind the finds: per cut, the lowest part that has found something (0xff..: none) -- later parts stop searching
    HIP_TRY(hipMemsetAsync(dcand.as<u64>() + (size_t)n_chunks * SPLIT * 4, 0xff, (size_t)n_chunks * 4, st));
    hipLaunchKernelGGL(sm_find_kernel, dim3((n_chunks - 1) * SPLIT), dim3(64), 0, st, d_in, n, off, cb, n_chunks, SPLIT, dcand.as<u64>());
#ifdef AHIP_PROFILE
    if (dbg) {
      std::vector<u64> pc((size_t)n_chunks * SPLIT * 4);
      HIP_TRY(hipMemcpy(pc.data(), dcand.p, pc.size() * 8, hipMemcpyDeviceToHost));
      const size_t nn = (size_t)n_chunks * SPLIT, wgs = (size_t)(n_chunks - 1) * SPLIT;
      double a = 0, b = 0, c = 0, d = 0;
      size_t busy = 0;
      for (size_t i = 0; i < wgs; ++i) { a += (double)pc[nn + i]; b += (double)pc[2 * nn + i]; c += (double)(pc[3 * nn + i] & 0xffffffffull); d += (double)(pc[3 * nn + i] >> 32); busy += pc[nn + i] != 0; }
      fprintf(stderr, "[ahip] sm find: per workgroup (%zu of %zu called the header check): calls %.2f  candidates %.1f  cycles in it %.0f of %.0f\n", busy, wgs, a / wgs, b / wgs, c / wgs, d / wgs);
    }
#endif
    std::vector<u64> found((size_t)n_chunks * SPLIT);
    HIP_TRY(hipMemcpyAsync(found.data(), dcand.p, (size_t)n_chunks * SPLIT * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    std::vector<u64> cand;
    cand.push_back(off * 8);
    for (u32 k = 1; k < n_chunks; ++k)
      for (u32 part = 0; part < SPLIT; ++part) {
        const u64 f = found[(size_t)k * SPLIT + part];
        if (f != ~0ull) { if (f > cand.back()) cand.push_back(f); break; }  // the first find behind the cut
      }
    if (dbg) fprintf(stderr, "[ahip] sm: %u cuts, %zu block starts found\n", n_chunks, cand.size());
    if (cand.size() < 4) return AHIP_OK;
    const u32 nc = (u32)cand.size();
    // sizing: every candidate decodes (no tokens stored) until a block starts on a later candidate
    std::vector<ChunkDesc> cd(nc);
    for (u32 i = 0; i < nc; ++i) cd[i] = ChunkDesc{cand[i], 0, 1ull << 62, i ? SM_WINDOW : hist0, 0};
    HIP_TRY(dcand.reserve((size_t)nc * 8));
    HIP_TRY(dchunks.reserve((size_t)nc * sizeof(ChunkDesc)));
    HIP_TRY(dres.reserve((size_t)nc * sizeof(MemberResult)));
    HIP_TRY(hipMemcpyAsync(dcand.p, cand.data(), (size_t)nc * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dchunks.p, cd.data(), (size_t)nc * sizeof(ChunkDesc), hipMemcpyHostToDevice, st));
    const u32 grid = nc < (u32)sm_resident_waves() ? nc : (u32)sm_resident_waves();
    // the sizing pass keeps its tokens, laid out along the input (12 B of scratch per compressed byte): the write pass
    // then only resolves them (AHIP_SM_TWO_PASS=1: tokenize again with exact offsets, as the first version did)
    void *ktp = nullptr, *ksp = nullptr;
    u64 kept_gen = 0;
    // Several buffers of areas where the memory is there (SmBase::ways, sm_layout_in: an area then reaches to the W-th candidate
    // behind its own): a chunk whose columns fill unevenly, or that decodes across a false block start, no longer runs out of room
    // -- one such chunk sends the WHOLE stream to "tokenize again" (a 1 GiB pigz-style member: 15.4 ms of which 2.6 are that).
    const u64 way_words = ((u64)n * IN_R + (u64)nc * IN_PAD + 64 + 15) & ~15ull, way_dirs = (u64)(n / 32) + (u64)nc * 64 + 64;
    u32 ways = 1;
    {
      static const int forced = [] { const char *e = getenv("AHIP_SM_WAYS"); return e ? atoi(e) : 0; }();  // (dev: 1, 2, 4)
      const u64 budget = 24ull << 30;
      for (u32 w : {4u, 2u}) if (ways == 1 && way_words * 4 * w <= budget) ways = w;
      if (forced == 1 || forced == 2 || forced == 4) ways = (u32)forced;
    }
    if (!getenv("AHIP_SM_TWO_PASS") && n <= (4ull << 30)) {
      for (;; ways >>= 1) {  // (a device short of memory: fewer buffers before none at all)
        if (tokens_reserve((size_t)way_words * ways * 4, &ktp) == hipSuccess &&
            scratch_reserve((size_t)way_dirs * ways * DIR_BYTES, &ksp) == hipSuccess) { kept_gen = g_tok_gen; break; }
        (void)hipGetLastError();
        ktp = nullptr; ksp = nullptr;
        if (ways == 1) break;
      }
    }
    g_sm.lay = SmBase{0, 0, 0, ways, way_words, way_dirs};
    u32 *ctr0 = nullptr;
    HIP_TRY(sm_counter(st, 0, &ctr0));
    hipLaunchKernelGGL(sm_tokenize_kernel, dim3(grid), dim3(64), 0, st, d_in, n, dchunks.as<ChunkDesc>(), nc, dcand.as<u64>(), nc,
                       (u32 *)ktp, (DirEnt *)ksp, dres.as<MemberResult>(), kept_gen ? 1u : 0u, ctr0, g_sm.lay);
    std::vector<MemberResult> rs(nc);
    HIP_TRY(hipMemcpyAsync(rs.data(), dres.p, (size_t)nc * sizeof(MemberResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    // the chain of chunks: each ends exactly where the next one starts
    u32 i = 0;
    u64 total = 0;
    for (;;) {
      const MemberResult &r = rs[i];
      // what a chunk may reach: the stream's own output in front of it, and behind that the hist0 bytes of earlier output
      g_sm.chain.push_back(ChunkDesc{cand[i], total, r.out_len, (u32)(total + hist0 < SM_WINDOW ? total + hist0 : SM_WINDOW), i});
      g_sm.sized.push_back(r);
      // the kept tokens serve only if the chunk was sized with the window it really has, and its token area held
      if (i && total + hist0 < SM_WINDOW) kept_gen = 0;
      if (r.blocks & MR_FAR) kept_gen = 0;
      total += r.out_len;
      g_sm.blocks += (r.blocks & 0x00ffffffu) | (i == 0 ? (r.blocks & MR_REACH) : 0u);
      if (r.status == MS_OK) { g_sm.end_pos = r.end_pos; break; }
      if (r.status != MS_CHUNK_END) { if (dbg) fprintf(stderr, "[ahip] sm: chunk %u ended with status %u: one-wave path\n", i, r.status); return AHIP_OK; }
      // (a chunk nearly always ends on the very next block start: look there first, search only when it does not)
      size_t j = (size_t)i + 1;
      if (j >= cand.size() || cand[j] != r.end_pos) j = (size_t)(std::lower_bound(cand.begin(), cand.end(), r.end_pos) - cand.begin());
      if (j >= cand.size() || cand[j] != r.end_pos || j <= i) return fail(AHIP_E_DEVICE, "internal: chunk chain broken");
      i = (u32)j;
    }
    // Only chunk 0 is WATCHED for references into what earlier gzip members wrote (q8).  A later chunk can reach there too
    // when chunk 0 made less than a window of output (a small AHIP_SM_CHUNK; 15-bit literal codes): taken as reaching --
    // the member is then decoded in stream order, after the members in front of it, which is always right.
    if (hist0 > 0 && g_sm.chain.size() > 1 && g_sm.chain[1].out_off < SM_WINDOW) g_sm.blocks |= MR_REACH;
    g_sm.cand = cand;
    g_sm.total_out = total;
    g_sm.d_in = d_in; g_sm.n = n; g_sm.off = off; g_sm.hist0 = hist0;
    g_sm.valid = true;
    g_sm.tok_gen = kept_gen;
    if (dbg) fprintf(stderr, "[ahip] sm: %zu chunks on the chain, %llu bytes out (find + sizing + chain: %.2f ms)\n", g_sm.chain.size(),
                     (unsigned long long)total, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  }
  *res = MemberResult{};
  res->status = MS_OK;
  res->out_len = g_sm.total_out;
  res->end_pos = g_sm.end_pos;
  res->blocks = g_sm.blocks;
  if (!write) { *handled = true; return AHIP_OK; }
  if (g_sm.hist0 != hist0) {
    // sized with another history than the output really has in front of this stream (the sizing run of a gzip member does
    // not know its output offset yet and allows the full 32 KiB): the chunks near the start get their true windows and are
    // tokenized again -- a reference that now reaches in front of the whole output shows up as a changed verdict below
    // and sends the stream to the one-wave path, which reports it exactly like the reference
    for (auto &c : g_sm.chain) c.hist = (u32)(c.out_off + hist0 < SM_WINDOW ? c.out_off + hist0 : SM_WINDOW);
    g_sm.tok_gen = 0;
    g_sm.hist0 = hist0;
  }
  if (g_sm.total_out > out_cap) { res->status = MS_CAP; *handled = true; return AHIP_OK; }
  // ---- tokens (exact offsets), symbols, windows, bytes ----
  const u32 nch = (u32)g_sm.chain.size(), nc = (u32)g_sm.cand.size();
  const u64 total = g_sm.total_out;
  void *tp = nullptr, *sp = nullptr;
  HIP_TRY(dsym.reserve((size_t)total * 2 + 64));
  HIP_TRY(dwin.reserve((size_t)nch * SM_WINDOW));
  HIP_TRY(dcand.reserve((size_t)nc * 8));
  HIP_TRY(dchunks.reserve((size_t)nch * sizeof(ChunkDesc)));
  HIP_TRY(dres.reserve((size_t)nch * sizeof(MemberResult)));
  HIP_TRY(hipMemcpyAsync(dcand.p, g_sm.cand.data(), (size_t)nc * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(dchunks.p, g_sm.chain.data(), (size_t)nch * sizeof(ChunkDesc), hipMemcpyHostToDevice, st));
  const u32 grid = nch < (u32)sm_resident_waves() ? nch : (u32)sm_resident_waves();
  const bool kept = g_sm.tok_gen && g_sm.tok_gen == g_tok_gen;  // the sizing pass's tokens are still in the scratch
  if (dbg) fprintf(stderr, "[ahip] sm: kept tokens %llu, scratch holds %llu -> %s\n", (unsigned long long)g_sm.tok_gen, (unsigned long long)g_tok_gen, kept ? "one pass" : "tokenize again");
  if (kept) {
    tp = g_tokens.p; sp = g_scratch.p;
    HIP_TRY(hipMemcpyAsync(dres.p, g_sm.sized.data(), (size_t)nch * sizeof(MemberResult), hipMemcpyHostToDevice, st));
  } else {
    HIP_TRY(tokens_reserve(((size_t)(total * 3 / 2) + (size_t)nch * 1024 + 64) * 4, &tp));
    HIP_TRY(scratch_reserve(((size_t)(total / 16) + (size_t)nch * 64 + 64) * DIR_BYTES, &sp));
    u32 *ctr1 = nullptr;
    HIP_TRY(sm_counter(st, 1, &ctr1));
    hipLaunchKernelGGL(sm_tokenize_kernel, dim3(grid), dim3(64), 0, st, d_in, n, dchunks.as<ChunkDesc>(), nch, dcand.as<u64>(), nc,
                       (u32 *)tp, (DirEnt *)sp, dres.as<MemberResult>(), 0u, ctr1, SmBase{0, 0, 0, 0, 0, 0});
    std::vector<MemberResult> rs(nch);
    HIP_TRY(hipMemcpyAsync(rs.data(), dres.p, (size_t)nch * sizeof(MemberResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (u32 i = 0; i < nch; ++i)
      if (rs[i].status != g_sm.sized[i].status || rs[i].out_len != g_sm.sized[i].out_len || rs[i].end_pos != g_sm.sized[i].end_pos) {
        // e.g. a back-reference into the void in front of the stream that the permissive sizing pass let through
        if (dbg) fprintf(stderr, "[ahip] sm: chunk %u differs from its sizing run (status %u vs %u): one-wave path\n", i, rs[i].status, g_sm.sized[i].status);
        g_sm.valid = false;
        return AHIP_OK;
      }
  }
  static thread_local DevBuf derr;  // chunks whose resolver gave up on a loop bound (cannot happen; if it does: AHIP_E_DEVICE, not wrong bytes)
  HIP_TRY(derr.reserve(16));
  HIP_TRY(hipMemsetAsync(derr.p, 0, 4, st));
  u32 *ctr2 = nullptr;
  HIP_TRY(sm_counter(st, 2, &ctr2));
  hipLaunchKernelGGL(sm_resolve_kernel, dim3(grid), dim3(64), 0, st, d_in, n, dchunks.as<ChunkDesc>(), nch, dsym.as<u16>(), (const u32 *)tp,
                     (const DirEnt *)sp, dres.as<MemberResult>(), dcand.as<u64>(), nc, kept ? 1u : 0u, derr.as<u32>(), ctr2, g_sm.lay);
  {
    static thread_local DevBuf dwsym, dgwin;
    u32 gs = 1;
    while ((u64)gs * gs < nch) ++gs;  // about sqrt(chunks) groups of sqrt(chunks) chunks: both serial parts equally short
    const u32 ng = (nch + gs - 1) / gs;
    HIP_TRY(dwsym.reserve((size_t)nch * SM_WINDOW * 2));
    HIP_TRY(dgwin.reserve((size_t)ng * SM_WINDOW));
    hipLaunchKernelGGL(sm_windows_group, dim3(ng), dim3(1024), 0, st, dchunks.as<ChunkDesc>(), dres.as<MemberResult>(), nch, gs,
                       dsym.as<u16>(), dwsym.as<u16>());
    // (the window in front of the stream: the hist0 bytes that end right in front of d_out -- already final, see plan_run)
    hipLaunchKernelGGL(sm_windows_link, dim3(1), dim3(1024), 0, st, nch, gs, dwsym.as<u16>(), dgwin.as<u8>(), (const u8 *)d_out - SM_WINDOW, hist0, SM_WINDOW);
    hipLaunchKernelGGL(sm_windows_apply, dim3(nch), dim3(1024), 0, st, gs, dwsym.as<u16>(), dgwin.as<u8>(), dwin.as<u8>(),
                       (const u8 *)d_out - SM_WINDOW, hist0, (const u8 *)nullptr);
  }
  hipLaunchKernelGGL(sm_translate_kernel, dim3(sm_translate_blocks(), nch), dim3(256), 0, st, dchunks.as<ChunkDesc>(), dres.as<MemberResult>(), dsym.as<u16>(),
                     dwin.as<u8>(), d_out, (const u8 *)d_out - SM_WINDOW, hist0, (const u8 *)nullptr);
  u32 res_err = 0;
  HIP_TRY(hipMemcpyAsync(&res_err, derr.p, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  g_sm.valid = false;
  if (res_err) return fail(AHIP_E_DEVICE, "internal: the chunk resolver ran into its loop bound");
  *handled = true;
  g_last_chunks = (int32_t)nch;
  if (dbg) fprintf(stderr, "[ahip] sm: write pass %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  return AHIP_OK;
}

int32_t inflate_one(const u8 *d_in, u64 n, u64 off, u8 *d_out, u64 out_cap, bool write, MemberResult *res,
                    hipStream_t st) {
  bool handled = false;
  int32_t rc = sm_inflate(d_in, n, off, d_out, out_cap, write, res, &handled, st);
  if (rc != AHIP_OK) return rc;
  if (handled) return AHIP_OK;
  return inflate_one_wave(d_in, n, off, d_out, out_cap, write, res, st);
}
int32_t inflate_one_wave(const u8 *d_in, u64 n, u64 off, u8 *d_out, u64 out_cap, bool write, MemberResult *res, hipStream_t st, u32 hist0) {
  static thread_local DevBuf dd, dr;
  HIP_TRY(dd.reserve(sizeof(MemberDesc)));
  HIP_TRY(dr.reserve(sizeof(MemberResult)));
  MemberDesc d{off, 0, out_cap, POS_UNKNOWN, 0, hist0, 0};  // (hist0 bytes of earlier output end right in front of d_out)
  HIP_TRY(hipMemcpyAsync(dd.p, &d, sizeof d, hipMemcpyHostToDevice, st));
  const u64 one_off[2] = {0, out_cap};
  if (write) HIP_TRY(launch_inflate<true>(d_in, n, dd.as<MemberDesc>(), 1u, d_out, dr.as<MemberResult>(), st, one_off));
  else HIP_TRY(launch_inflate<false>(d_in, n, dd.as<MemberDesc>(), 1u, (u8 *)nullptr, dr.as<MemberResult>(), st));
  HIP_TRY(hipMemcpyAsync(res, dr.p, sizeof *res, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return AHIP_OK;
}

// Where the reference's InputStream stands after the last decode call of this thread (ahip_last_consumed): decodeStream
// CONSUMES its input -- all of it when it returns true; on `false` it stops where the check that failed left the reader
// (_zlib_decoder_web.dart:53-99), and callers that go on reading the stream must find it there.
static thread_local u64 g_consumed = 0;

int32_t member_status_to_abi(u32 ms) {
  switch (ms) {
    case MS_OK: case MS_EOS: return AHIP_OK;
    case MS_FALSE: return AHIP_FALSE;
    case MS_RANGE: case MS_FARREF: return AHIP_RANGE;  // single stream: source before index 0
    case MS_HANG: return AHIP_HANG;
    default: g_err = "internal: unexpected member status"; return AHIP_E_DEVICE;
  }
}

// _ZLibDecoder.decodeStream on device memory (serial member loop, deferred flush, quirk q7).
//  d_out: device output, `committed` bytes already hold earlier output.  The member being
//  inflated is written right behind the committed bytes and only counted once the NEXT header
//  passed its checks (or the input ended).
//  host_in: host copy of the input (headers/trailers are read there).
int32_t zlib_stream_device(const u8 *host_in, const u8 *d_in, u64 n, u64 pos, bool big_endian, int verify, int raw,
                           DevBuf &outbuf, u64 *committed_io, hipStream_t st) {
#define ZRET(code) do { g_consumed = pos; return (code); } while (0)
  u64 committed = *committed_io;
  bool have = false;
  u64 buf_len = 0;
  std::vector<u8> tmp;
  while (pos < n) {
    if (!raw) {
      if (pos + 2 > n) return AHIP_RANGE;
      u32 cmf = host_in[pos], flg = host_in[pos + 1];
      pos += 2;
      if ((cmf & 8) != 8) { *committed_io = committed; ZRET(AHIP_FALSE); }
      if (((cmf * 256) + flg) % 31 != 0) { *committed_io = committed; ZRET(AHIP_FALSE); }
      if ((flg & 32) >> 5) {  // readUint32() of the dictionary id, then `false`
        *committed_io = committed;
        if (pos + 4 > n) return AHIP_RANGE;
        pos += 4;
        ZRET(AHIP_FALSE);
      }
    }
    if (have) committed += buf_len;
    // sizing pass, then the real one into a window of exactly that size
    MemberResult r{};
    int32_t rc = inflate_one(d_in, n, pos, nullptr, ~0ull, false, &r, st);
    if (rc != AHIP_OK) return rc;
    if (r.status == MS_RANGE || r.status == MS_FARREF) return AHIP_RANGE;
    if (r.status == MS_HANG) return AHIP_HANG;
    if (r.out_len) {
      // grow keeping the committed prefix
      if (committed + r.out_len > outbuf.cap) {
        DevBuf nb;
        HIP_TRY(nb.reserve((committed + r.out_len) * 2));
        if (committed) HIP_TRY(hipMemcpyAsync(nb.p, outbuf.p, committed, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        outbuf.release();
        outbuf.p = nb.p; outbuf.cap = nb.cap;  // (moved, not copied: a DevBuf owns its block)
        nb.p = nullptr; nb.cap = 0;
      }
      MemberResult r2{};
      rc = inflate_one(d_in, n, pos, outbuf.as<u8>() + committed, r.out_len, true, &r2, st);
      if (rc != AHIP_OK) return rc;
      if (r2.out_len != r.out_len || r2.end_pos != r.end_pos)
        return fail(AHIP_E_DEVICE, "internal: decode disagrees with its own sizing run");
    }
    have = true;
    buf_len = r.out_len;
    pos = r.end_pos;
    if (!raw) {
      if (pos + 4 > n) return AHIP_RANGE;
      u32 a = host_in[pos], b = host_in[pos + 1], c = host_in[pos + 2], d = host_in[pos + 3];
      u32 want = big_endian ? ((a << 24) | (b << 16) | (c << 8) | d) : ((d << 24) | (c << 16) | (b << 8) | a);
      pos += 4;
      if (verify) {  // Adler-32 of the member's bytes where they are: in HBM
        u32 got = 0;
        rc = adler32_device_impl(outbuf.as<u8>() + committed, buf_len, 1, &got, st);
        if (rc != AHIP_OK) return rc;
        if (got != want) { *committed_io = committed; ZRET(AHIP_FALSE); }
      }
    }
  }
  if (have) committed += buf_len;
  *committed_io = committed;
  ZRET(AHIP_OK);
#undef ZRET
}

// ---- ONE long stream decoded by several ranks (the member loop of _gzip_decoder_web.dart:29-55 has
// nothing to shard when there is one member; inflate.dart:104-156 is one block loop) ----
// Every rank holds the whole COMPRESSED stream and takes an equal range of its cuts: it finds the block starts behind its
// own cuts, sizes its own candidates (keeping the tokens), resolves the chunks of the chain that are its own to symbols and
// composes "the last 32 KiB of my range as a function of the 32 KiB in front of it".  Three things cross between ranks,
// each as one all-gather done by the caller (archive_amd/sharding.py::ShardedStreamDecoder): the candidate lists, the
// sizing results (32 B a candidate) and the window maps (64 KiB a rank).  What every rank decides from gathered data it
// decides alike (the chain); what it learns alone (a chunk that differs from its sizing run) travels in the map's status word.
constexpr u32 SPLIT_MAP_ELEMS = SM_WINDOW + 32;  // u16 elements of a rank's map in the exchange buffer: the map, then the status word
struct SplitState {
  const u8 *d_in = nullptr;
  u64 n = 0, off = 0, cb = 0;
  u32 rank = 0, world = 1, n_cuts = 0, k0 = 0, k1 = 0;
  hipStream_t st = nullptr;
  bool eligible = false;
  int phase = 0;               // 0 created, 1 candidates found, 2 sized, 3 chain known, 4 resolved, 5 finished
  std::vector<u64> own, cand;  // this rank's block starts; everybody's (rank order = stream order)
  u32 c0 = 0, c1 = 0;          // own == cand[c0 .. c1)
  u64 byte0 = 0;               // first input byte of the range (token areas are laid out from there)
  SmBase base_in{0, 0, 0, 0, 0, 0};  // ... and how (sm_layout_in)
  std::vector<MemberResult> own_res, sized;
  std::vector<ChunkDesc> chain;  // the chunks of the chain this rank owns; out_off counted from base_out
  std::vector<u8> retok;         // ... and which of them cannot use the tokens their sizing run kept
  u64 base_out = 0, out_len = 0, total_out = 0, end_pos = 0;
  bool kept = false, have_tokens = false;
  u32 gs = 1, ng = 0;
  DevBuf dfind, dcand, dchunks, dres, dsym, dwin, dwsym, dgsym, dgwin, dlink, dtok, ddir, derr, dctr, dmap, dmaps, dtok2, ddir2, dsubc, dsubr;  // (dmap / dmaps: the one-process form's exchange buffers; dtok2 ..: chunks tokenized again next to kept ones)
  void release() { for (DevBuf *b : {&dfind, &dcand, &dchunks, &dres, &dsym, &dwin, &dwsym, &dgsym, &dgwin, &dlink, &dtok, &ddir, &derr, &dctr, &dmap, &dmaps, &dtok2, &ddir2, &dsubc, &dsubr}) b->release(); }
};

hipError_t split_counter(SplitState *h, u32 slot, u32 **out) {
  *out = nullptr;
  if (!sm_use_counter()) return hipSuccess;
  hipError_t e = h->dctr.reserve(256);
  if (e != hipSuccess) return e;
  *out = h->dctr.as<u32>() + 16 * slot;
  return hipMemsetAsync(*out, 0, 4, h->st);
}

int32_t split_candidates(SplitState *h) {
  h->own.clear();
  if (h->rank == 0) h->own.push_back(h->off * 8);
  // the cuts K in [max(k0, 1), k1): sm_find_kernel searches cuts 1 .. n - 1 behind `data_start`
  const u32 K0 = h->k0 > 1 ? h->k0 : 1;
  if (h->k1 > K0) {
    static const u32 SPLIT = [] { const char *e = getenv("AHIP_SM_SPLIT"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? (u32)v : 4u; }();
    const u32 nloc = h->k1 - K0 + 1;
    const u64 start = h->off + (u64)(K0 - 1) * h->cb;
    HIP_TRY(h->dfind.reserve((size_t)nloc * SPLIT * 8 * 4 + (size_t)nloc * 4));
    HIP_TRY(hipMemsetAsync(h->dfind.as<u64>() + (size_t)nloc * SPLIT * 4, 0xff, (size_t)nloc * 4, h->st));
    hipLaunchKernelGGL(sm_find_kernel, dim3((nloc - 1) * SPLIT), dim3(64), 0, h->st, h->d_in, h->n, start, h->cb, nloc, SPLIT, h->dfind.as<u64>());
    std::vector<u64> found((size_t)nloc * SPLIT);
    HIP_TRY(hipMemcpyAsync(found.data(), h->dfind.p, found.size() * 8, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    HIP_TRY(hipGetLastError());
    for (u32 k = 1; k < nloc; ++k)
      for (u32 part = 0; part < SPLIT; ++part) {
        const u64 f = found[(size_t)k * SPLIT + part];
        if (f != ~0ull) { if (h->own.empty() || f > h->own.back()) h->own.push_back(f); break; }
      }
  }
  h->phase = 1;
  return AHIP_OK;
}

// everybody's candidates are known: size the own ones.  *handled = false: not a case for the chunked decode (too few block starts)
int32_t split_size(SplitState *h, const u64 *all, size_t n_all, bool *handled) {
  *handled = false;
  for (size_t i = 1; i < n_all; ++i) if (all[i] <= all[i - 1]) return fail(AHIP_E_ARG, "stream split: the gathered block starts are not in stream order");
  if (n_all == 0 || all[0] != h->off * 8) return fail(AHIP_E_ARG, "stream split: the gathered block starts do not begin with the stream's first bit");
  h->cand.assign(all, all + n_all);
  h->c0 = h->own.empty() ? 0 : (u32)(std::lower_bound(h->cand.begin(), h->cand.end(), h->own[0]) - h->cand.begin());
  h->c1 = h->c0 + (u32)h->own.size();
  if (h->c1 > n_all || !std::equal(h->own.begin(), h->own.end(), h->cand.begin() + h->c0)) return fail(AHIP_E_ARG, "stream split: this rank's block starts are not in the gathered list");
  h->own_res.assign(h->own.size(), MemberResult{});
  h->phase = 2;
  if (n_all < 4 || n_all > 0x7fffffffu) return AHIP_OK;
  *handled = true;
  const u32 nown = (u32)h->own.size(), nc = (u32)n_all;
  if (!nown) return AHIP_OK;
  std::vector<ChunkDesc> cd(nown);
  for (u32 i = 0; i < nown; ++i) cd[i] = ChunkDesc{h->own[i], 0, 1ull << 62, (h->c0 + i) ? SM_WINDOW : 0u, 0};
  HIP_TRY(h->dcand.reserve((size_t)nc * 8));
  HIP_TRY(h->dchunks.reserve((size_t)nown * sizeof(ChunkDesc)));
  HIP_TRY(h->dres.reserve((size_t)nown * sizeof(MemberResult)));
  HIP_TRY(hipMemcpyAsync(h->dcand.p, h->cand.data(), (size_t)nc * 8, hipMemcpyHostToDevice, h->st));
  HIP_TRY(hipMemcpyAsync(h->dchunks.p, cd.data(), (size_t)nown * sizeof(ChunkDesc), hipMemcpyHostToDevice, h->st));
  // the tokens are kept, laid out along the range's input bytes (tok_layout_in, like the sizing pass of sm_inflate)
  h->byte0 = h->own[0] >> 3;
  // four buffers of areas (sm_layout_in), an area reaches to the fourth candidate behind its own
  u32 ways = 4;
  while (ways > 1 && (u64)(h->n - h->byte0) * IN_R * 4 * ways > (24ull << 30)) ways >>= 1;  // (a crude bound: the exact size follows)
  const u64 byte1 = h->c1 - 1 + ways < nc ? (h->cand[h->c1 - 1 + ways] >> 3) + 1 : h->n;
  const u64 span = byte1 > h->byte0 ? byte1 - h->byte0 : 0;
  const u64 way_words = (span * IN_R + (u64)nown * IN_PAD + 64 + 15) & ~15ull, way_dirs = span / 32 + (u64)nown * 64 + 64;
  h->base_in = SmBase{h->byte0, h->c0, h->c0, ways, way_words, way_dirs};
  h->have_tokens = false;
  if (!getenv("AHIP_SM_TWO_PASS") && span <= (4ull << 30) &&
      h->dtok.reserve((size_t)way_words * ways * 4) == hipSuccess &&
      h->ddir.reserve((size_t)way_dirs * ways * DIR_BYTES) == hipSuccess) h->have_tokens = true;
  else (void)hipGetLastError();  // (no room to keep the tokens: sized without, tokenized again later)
  const u32 grid = nown < (u32)sm_resident_waves() ? nown : (u32)sm_resident_waves();
  u32 *ctr = nullptr;
  HIP_TRY(split_counter(h, 0, &ctr));
  hipLaunchKernelGGL(sm_tokenize_kernel, dim3(grid), dim3(64), 0, h->st, h->d_in, h->n, h->dchunks.as<ChunkDesc>(), nown, h->dcand.as<u64>(), nc,
                     h->have_tokens ? h->dtok.as<u32>() : (u32 *)nullptr, h->have_tokens ? h->ddir.as<DirEnt>() : (DirEnt *)nullptr,
                     h->dres.as<MemberResult>(), h->have_tokens ? 1u : 0u, ctr, h->base_in);
  HIP_TRY(hipMemcpyAsync(h->own_res.data(), h->dres.p, (size_t)nown * sizeof(MemberResult), hipMemcpyDeviceToHost, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  HIP_TRY(hipGetLastError());
  return AHIP_OK;
}

// everybody's sizing results are known (4 words a candidate: status, bytes, end position, blocks): follow the chain of
// ends == starts from the stream's first bit exactly like sm_inflate, keep what is this rank's
int32_t split_chain(SplitState *h, const u64 *res, size_t n_all, bool *handled) {
  *handled = false;
  if (h->phase != 2 || n_all != h->cand.size()) return fail(AHIP_E_ARG, "stream split: results for another candidate list");
  h->phase = 3;
  h->chain.clear(); h->sized.clear(); h->retok.clear();
  h->base_out = 0; h->out_len = 0; h->total_out = 0; h->end_pos = 0;
  if (n_all < 4) return AHIP_OK;
  h->kept = h->have_tokens;
  const bool dbg = getenv("AHIP_DEBUG") != nullptr;
  u64 total = 0;
  bool based = false;
  size_t i = 0;
  for (;;) {
    const u64 status = res[4 * i], out_len = res[4 * i + 1], end_pos = res[4 * i + 2], blocks = res[4 * i + 3];
    const bool mine = i >= h->c0 && i < h->c1;
    if (!based && i >= h->c0) { h->base_out = total; based = true; }
    if (mine) {
      const MemberResult &r = h->own_res[i - h->c0];
      if (r.status != status || r.out_len != out_len || r.end_pos != end_pos) return fail(AHIP_E_ARG, "stream split: the gathered results differ from this rank's own");
      h->chain.push_back(ChunkDesc{h->cand[i], total - h->base_out, out_len, (u32)(total < SM_WINDOW ? total : SM_WINDOW), (u32)i});
      h->sized.push_back(r);
      // sized with a full window in front but has less (see sm_inflate), or its token area did not hold: THIS chunk is tokenized again
      h->retok.push_back((i && total < SM_WINDOW) || (blocks & MR_FAR) ? 1 : 0);
      if (dbg && h->retok.back()) fprintf(stderr, "[ahip] stream split: rank %u chunk %zu (candidate %zu, %llu bytes out) is tokenized again\n", h->rank, h->chain.size() - 1, i, (unsigned long long)out_len);
      h->out_len += out_len;
    }
    total += out_len;
    if (status == MS_OK) { h->end_pos = end_pos; break; }
    if (status != MS_CHUNK_END) { if (dbg) fprintf(stderr, "[ahip] stream split: chunk %zu ended with status %llu: not for this path\n", i, (unsigned long long)status); return AHIP_OK; }
    // (a chunk nearly always ends on the very next block start: look there first, search only when it does not)
    size_t j = i + 1;
    if (j >= n_all || h->cand[j] != end_pos) j = (size_t)(std::lower_bound(h->cand.begin(), h->cand.end(), end_pos) - h->cand.begin());
    if (j >= n_all || h->cand[j] != end_pos || j <= i) return fail(AHIP_E_DEVICE, "internal: chunk chain broken");
    i = j;
  }
  if (!based) h->base_out = total;
  h->total_out = total;
  *handled = true;
  return AHIP_OK;
}

// the own chunks -> symbols; the range's window map (+ status word: 1 = fine) -> d_map
int32_t split_resolve(SplitState *h, u16 *d_map) {
  if (h->phase != 3) return fail(AHIP_E_ARG, "stream split: resolve before the chain is known");
  h->phase = 4;
  const u32 nch = (u32)h->chain.size(), nc = (u32)h->cand.size();
  u32 ok = 1;
  h->gs = 1; h->ng = 0;
  if (nch) {
    HIP_TRY(h->dsym.reserve((size_t)h->out_len * 2 + 64));
    HIP_TRY(h->dchunks.reserve((size_t)nch * sizeof(ChunkDesc)));
    HIP_TRY(h->dres.reserve((size_t)nch * sizeof(MemberResult)));
    HIP_TRY(hipMemcpyAsync(h->dchunks.p, h->chain.data(), (size_t)nch * sizeof(ChunkDesc), hipMemcpyHostToDevice, h->st));
    const u32 resident = (u32)sm_resident_waves();
    // the chunks that use the tokens their sizing run kept (K) and those that are tokenized again with exact offsets and
    // windows (R: one whose token area did not hold, one near the stream's start that was sized with a full window)
    std::vector<u32> K, R;
    // AHIP_SPLIT_TEST_RETOK=n (tests): every n-th chunk is tokenized again whatever its sizing run said, and the two kinds are kept apart
    static const u32 test_every = [] { const char *e = getenv("AHIP_SPLIT_TEST_RETOK"); return e && atoi(e) > 0 ? (u32)atoi(e) : 0u; }();
    for (u32 k = 0; k < nch; ++k) (h->have_tokens && !h->retok[k] && !(test_every && k % test_every == test_every - 1) ? K : R).push_back(k);
    // (The two kernels of the R chunks cost a chunk's serial time each, behind one another: worth it only where tokenizing
    //  everything again is several rounds of the resident waves.  Otherwise: all of them again, side by side.)
    if (!R.empty() && !K.empty() && nch <= 2 * resident && !test_every) { K.clear(); R.clear(); for (u32 k = 0; k < nch; ++k) R.push_back(k); }
    h->kept = R.empty();
    std::vector<MemberResult> full = h->sized;
    HIP_TRY(h->derr.reserve(16));
    HIP_TRY(hipMemsetAsync(h->derr.p, 0, 4, h->st));
    const u32 nK = (u32)K.size(), nR = (u32)R.size();
    // sub-lists (device): [0, nK) the kept chunks, [nK, nch) the others -- the whole list itself when there is only one kind
    const ChunkDesc *dK = h->dchunks.as<ChunkDesc>(), *dR = h->dchunks.as<ChunkDesc>();
    MemberResult *rK = h->dres.as<MemberResult>(), *rR = h->dres.as<MemberResult>();
    if (nK && nR) {
      std::vector<ChunkDesc> sc(nch);
      std::vector<MemberResult> sr(nch);
      for (u32 a = 0; a < nK; ++a) { sc[a] = h->chain[K[a]]; sr[a] = h->sized[K[a]]; }
      for (u32 a = 0; a < nR; ++a) { sc[nK + a] = h->chain[R[a]]; sr[nK + a] = h->sized[R[a]]; }
      HIP_TRY(h->dsubc.reserve((size_t)nch * sizeof(ChunkDesc)));
      HIP_TRY(h->dsubr.reserve((size_t)nch * sizeof(MemberResult)));
      HIP_TRY(hipMemcpyAsync(h->dsubc.p, sc.data(), (size_t)nch * sizeof(ChunkDesc), hipMemcpyHostToDevice, h->st));
      HIP_TRY(hipMemcpyAsync(h->dsubr.p, sr.data(), (size_t)nch * sizeof(MemberResult), hipMemcpyHostToDevice, h->st));
      HIP_TRY(hipStreamSynchronize(h->st));  // (sc / sr are locals)
      dK = h->dsubc.as<ChunkDesc>(); dR = dK + nK;
      rK = h->dsubr.as<MemberResult>(); rR = rK + nK;
    }
    const u32 *tpR = nullptr;
    const DirEnt *spR = nullptr;
    if (nR) {
      // exact offsets, exact windows: tokenize these again (tok_layout on offsets counted from the range's first byte; token areas of
      // their own, the kept tokens stay where they are)
      HIP_TRY(h->dtok2.reserve(((size_t)(h->out_len * 3 / 2) + (size_t)nch * 1024 + 64) * 4));
      HIP_TRY(h->ddir2.reserve(((size_t)(h->out_len / 16) + (size_t)nch * 64 + 64) * DIR_BYTES));
      tpR = h->dtok2.as<u32>(); spR = h->ddir2.as<DirEnt>();
      u32 *ctr1 = nullptr;
      HIP_TRY(split_counter(h, 1, &ctr1));
      hipLaunchKernelGGL(sm_tokenize_kernel, dim3(nR < resident ? nR : resident), dim3(64), 0, h->st, h->d_in, h->n, dR, nR, h->dcand.as<u64>(), nc,
                         h->dtok2.as<u32>(), h->ddir2.as<DirEnt>(), rR, 0u, ctr1, SmBase{0, 0, h->chain[R[0]].pad, 0, 0, 0});
      std::vector<MemberResult> rs(nR);
      HIP_TRY(hipMemcpyAsync(rs.data(), rR, (size_t)nR * sizeof(MemberResult), hipMemcpyDeviceToHost, h->st));
      HIP_TRY(hipStreamSynchronize(h->st));
      HIP_TRY(hipGetLastError());
      for (u32 a = 0; a < nR; ++a) {
        const MemberResult &z = h->sized[R[a]];
        if (rs[a].status != z.status || rs[a].out_len != z.out_len || rs[a].end_pos != z.end_pos) ok = 0;
        full[R[a]] = rs[a];
      }
    }
    // (the windows and the last pass read sizes from the list in chain order)
    if (!(nR && !nK)) HIP_TRY(hipMemcpyAsync(h->dres.p, full.data(), (size_t)nch * sizeof(MemberResult), hipMemcpyHostToDevice, h->st));
    if (ok) {
      if (nK) {
        u32 *ctr2 = nullptr;
        HIP_TRY(split_counter(h, 2, &ctr2));
        hipLaunchKernelGGL(sm_resolve_kernel, dim3(nK < resident ? nK : resident), dim3(64), 0, h->st, h->d_in, h->n, dK, nK, h->dsym.as<u16>(), h->dtok.as<u32>(),
                           h->ddir.as<DirEnt>(), rK, h->dcand.as<u64>(), nc, 1u, h->derr.as<u32>(), ctr2, h->base_in);
      }
      if (nR) {
        u32 *ctr3 = nullptr;
        HIP_TRY(split_counter(h, 3, &ctr3));
        hipLaunchKernelGGL(sm_resolve_kernel, dim3(nR < resident ? nR : resident), dim3(64), 0, h->st, h->d_in, h->n, dR, nR, h->dsym.as<u16>(), tpR, spR,
                           rR, h->dcand.as<u64>(), nc, 0u, h->derr.as<u32>(), ctr3, SmBase{0, 0, 0, 0, 0, 0});
      }
      while ((u64)h->gs * h->gs < nch) ++h->gs;
      h->ng = (nch + h->gs - 1) / h->gs;
      HIP_TRY(h->dwsym.reserve((size_t)nch * SM_WINDOW * 2));
      hipLaunchKernelGGL(sm_windows_group, dim3(h->ng), dim3(1024), 0, h->st, h->dchunks.as<ChunkDesc>(), h->dres.as<MemberResult>(), nch, h->gs,
                         h->dsym.as<u16>(), h->dwsym.as<u16>());
    }
  }
  HIP_TRY(h->dgsym.reserve((size_t)(h->ng ? h->ng : 1) * SM_WINDOW * 2));
  hipLaunchKernelGGL(sm_windows_link_sym, dim3(1), dim3(1024), 0, h->st, ok ? nch : 0u, h->gs, h->dwsym.as<u16>(), h->dgsym.as<u16>());
  HIP_TRY(hipMemcpyAsync(d_map, h->dgsym.as<u16>() + (size_t)(h->ng && ok ? h->ng - 1 : 0) * SM_WINDOW, (size_t)SM_WINDOW * 2, hipMemcpyDeviceToDevice, h->st));
  u32 res_err = 0;
  if (nch) HIP_TRY(hipMemcpyAsync(&res_err, h->derr.p, 4, hipMemcpyDeviceToHost, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  HIP_TRY(hipGetLastError());
  if (res_err) return fail(AHIP_E_DEVICE, "internal: the chunk resolver ran into its loop bound");
  const u32 word[2] = {ok, 0};
  HIP_TRY(hipMemcpyAsync(d_map + SM_WINDOW, word, 8, hipMemcpyHostToDevice, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  if (!ok) h->phase = 6;  // nothing left to do here
  return AHIP_OK;
}

// everybody's maps are known: the bytes in front of the range, the chunks' windows, the range's bytes
int32_t split_finish(SplitState *h, const u16 *d_maps, u8 *d_out, size_t out_cap, size_t *out_len, bool *handled) {
  *handled = false;
  if (out_len) *out_len = 0;
  if (h->phase != 4 && h->phase != 6) return fail(AHIP_E_ARG, "stream split: finish before resolve");
  std::vector<u32> ok(h->world, 0);
  // (the status words of all ranks in ONE strided copy: a copy per rank was 24 us each)
  HIP_TRY(hipMemcpy2DAsync(ok.data(), 4, d_maps + SM_WINDOW, (size_t)SPLIT_MAP_ELEMS * 2, 4, h->world, hipMemcpyDeviceToHost, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  for (u32 r = 0; r < h->world; ++r) if (ok[r] != 1) { h->phase = 5; return AHIP_OK; }  // some rank's chunk differs from its sizing run: the caller's exact path
  if (out_len) *out_len = h->out_len;
  if (h->out_len > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
  const u32 nch = (u32)h->chain.size();
  h->phase = 5;
  *handled = true;
  if (!nch) return AHIP_OK;
  const u8 *entry = nullptr;
  if (h->rank) {  // the ranks in front, linked in rank order: what their ranges leave behind
    HIP_TRY(h->dlink.reserve((size_t)h->rank * SM_WINDOW));
    hipLaunchKernelGGL(sm_windows_link, dim3(1), dim3(1024), 0, h->st, h->rank, 1u, d_maps, h->dlink.as<u8>(), (const u8 *)nullptr, 0u, SPLIT_MAP_ELEMS);
    entry = h->dlink.as<u8>() + (size_t)(h->rank - 1) * SM_WINDOW;
  }
  HIP_TRY(h->dgwin.reserve((size_t)h->ng * SM_WINDOW));
  HIP_TRY(h->dwin.reserve((size_t)nch * SM_WINDOW));
  // the groups' last windows (every one a function of the entry window), then every chunk's
  hipLaunchKernelGGL(sm_windows_apply, dim3(h->ng), dim3(1024), 0, h->st, 0x7fffffffu, h->dgsym.as<u16>(), (const u8 *)nullptr, h->dgwin.as<u8>(),
                     (const u8 *)nullptr, 0u, entry);
  hipLaunchKernelGGL(sm_windows_apply, dim3(nch), dim3(1024), 0, h->st, h->gs, h->dwsym.as<u16>(), h->dgwin.as<u8>(), h->dwin.as<u8>(),
                     (const u8 *)nullptr, 0u, entry);
  hipLaunchKernelGGL(sm_translate_kernel, dim3(sm_translate_blocks(), nch), dim3(256), 0, h->st, h->dchunks.as<ChunkDesc>(), h->dres.as<MemberResult>(),
                     h->dsym.as<u16>(), h->dwin.as<u8>(), d_out, (const u8 *)nullptr, 0u, entry);
  HIP_TRY(hipStreamSynchronize(h->st));
  HIP_TRY(hipGetLastError());
  return AHIP_OK;
}

}  // namespace
struct ahip_stream_split { SplitState s; };

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
extern "C" {

uint32_t ahip_abi_version(void) { return (2u << 16) | 4u; }

const char *ahip_last_error(void) { return g_err.c_str(); }

size_t ahip_last_consumed(void) { return (size_t)g_consumed; }

int32_t ahip_init(int32_t device) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  if (device >= 0) HIP_TRY(hipSetDevice(device));
  return AHIP_OK;
}

void rccl_drop();
void ahip_shutdown(void) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  stop_workers();
  rccl_drop();
  for (auto &b : g_pool) (void)hipFree(b.p);
  g_pool.clear();
  g_inited = false;
}

uint32_t ahip_crc32(const uint8_t *data, size_t len, uint32_t crc) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      table[i] = c;
    }
    ready = true;
  }
  crc ^= 0xffffffffu;
  for (size_t i = 0; i < len; ++i) crc = table[(crc ^ data[i]) & 0xff] ^ (crc >> 8);
  return crc ^ 0xffffffffu;
}

uint32_t ahip_adler32(const uint8_t *data, size_t len, uint32_t adler) {
  uint64_t s1 = adler & 0xffff, s2 = adler >> 16;
  size_t i = 0;
  while (len > 0) {
    size_t k = len < 3800 ? len : 3800;
    len -= k;
    while (k--) { s1 += data[i++]; s2 += s1; }
    s1 %= 65521; s2 %= 65521;
  }
  return (uint32_t)((s2 << 16) | s1);
}

static bool bz_parallel_huffman() { return getenv("AHIP_BZ_SERIAL_HUFFMAN") == nullptr; }  // (read per call: the tests switch it)
// bzip2 on device memory.  `in` = the first bytes of the stream on the host (header checks), d_in = the whole
// stream on the device, d_out = device output of out_cap bytes.
// sh (sharded decode, ahip_bzip2_decode_shards): this call decodes only the blocks among candidates
// [ncand * index / count, ncand * (index + 1) / count) of the stream -- every device scans the whole (small) compressed
// stream for block magics, so all agree on the list -- and reports what the caller needs to merge the shards in order:
// how its part of the chain ended, and the fold of its block CRCs (the stream CRC is rotl-xor over the blocks: linear).
struct BzShard {
  u32 index = 0, count = 1;
  u64 nblocks = 0;       // block CRCs folded
  u32 fold = 0;          // combined CRC of those blocks, starting from 0
  bool saw_eos = false;  // met the end-of-stream block
  u32 eos_stored = 0;
  bool stopped = false;  // the stream ended inside this shard (end-of-stream block, clean end of input, or a verdict)
  u64 from = ~0ull;      // in: start the chain at this candidate instead of the range's first (the merge's second try)
  u64 first = 0, next = 0;  // out: the candidate this shard's chain started at / expects next (>= the range's end unless stopped)
};
static int32_t bzip2_device_impl(const u8 *in, const u8 *d_in, size_t in_len, int32_t verify, u8 *d_out, size_t out_cap,
                                 size_t *out_len, BzShard *sh = nullptr, hipStream_t st = nullptr) {
  // (every launch, fill and copy below is ordered on `st` -- the caller's stream, a worker context's own, or the default one;
  //  the verdicts the host chain needs are read back with copy_on: a copy on `st` that the host waits for)
  if (out_len) *out_len = 0;
  // BZh + level, read through the bit reader: fewer than 4 bytes is a RangeError in the reference
  // (ahip_last_consumed: the reference's reader has pulled the bytes it compared, `||` stops at the first that differs --
  //  bzip2_decoder.dart:29-41; never the position a previous call of this thread left)
  g_consumed = 0;
  if (in_len < 4) {
    for (size_t i = 0; i < in_len && i < 3; ++i) if (in[i] != "BZh"[i]) { g_consumed = i + 1; return AHIP_FALSE; }
    return AHIP_RANGE;
  }
  for (size_t i = 0; i < 3; ++i) if (in[i] != "BZh"[i]) { g_consumed = i + 1; return AHIP_FALSE; }
  const int level = (int)in[3] - 0x30;
  g_consumed = 4;
  if (level < 0 || level > 9) return AHIP_FALSE;
  if (in_len == 4) return AHIP_OK;  // while (!input.isEOS) never runs
  if (level == 0) return AHIP_FALSE;  // zero-sized tt: the first symbol already fails nblock >= nblockMAX
  static thread_local DevBuf dcand, dcount, dtt, dsel, dres, dcrc, doff;
  // B0: block / end-of-stream magics at any bit offset
  const u32 cap_c = (u32)(in_len / 32 + 64);
  HIP_TRY(dcand.reserve((size_t)cap_c * sizeof(BzCand)));
  HIP_TRY(dcount.reserve(16));
  HIP_TRY(hipMemsetAsync(dcount.p, 0, 4, st));
  hipLaunchKernelGGL(bz_scan_magic, dim3(cdiv(in_len, 256)), dim3(256), 0, st, d_in, (u64)in_len,
                     dcand.as<BzCand>(), dcount.as<u32>(), cap_c);
  u32 ncand = 0;
  HIP_TRY(copy_on(&ncand, dcount.p, 4, hipMemcpyDeviceToHost, st));
  if (ncand > cap_c) return fail(AHIP_E_UNSUPPORTED, "too many bzip2 block-magic candidates");
  std::vector<BzCand> cands(ncand);
  if (ncand) HIP_TRY(copy_on(cands.data(), dcand.p, (size_t)ncand * sizeof(BzCand), hipMemcpyDeviceToHost, st));
  std::sort(cands.begin(), cands.end(), [](const BzCand &a, const BzCand &b) { return a.bit < b.bit; });
  // the first block type is read at bit 32; anything else there is "Invalid Block Signature"
  // the stream's seven bytes from bit >> 3 on (zeros beyond the end): what _readBlockType reads where no magic starts
  auto peek = [&](u64 bit, u8 *b7) {
    const u64 p = bit >> 3;
    if (p < in_len) (void)copy_on(b7, d_in + p, (size_t)std::min<u64>(7, in_len - p), hipMemcpyDeviceToHost, st);
  };
  if (ncand == 0 || cands[0].bit != 32) {
    if (sh) sh->stopped = true;
    u8 b7[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    peek(32, b7);
    u64 sb = 32;
    const int32_t v0 = bz_no_magic_verdict(32, in_len, b7, &sb);
    g_consumed = std::min<u64>((u64)in_len, (sb + 7) / 8);
    return v0;
  }
  size_t c_lo = sh ? (size_t)((u64)ncand * sh->index / sh->count) : 0;
  const size_t c_hi = sh ? (size_t)((u64)ncand * (sh->index + 1) / sh->count) : ncand;
  if (sh) {
    // (the chain of the shards in front may end somewhere else than at this shard's first candidate -- a false magic
    //  inside a block's data on the boundary: the merge then runs this shard again from where the chain really stands)
    if (sh->from != ~0ull) c_lo = (size_t)sh->from;
    sh->first = c_lo; sh->next = c_lo;
  }
  if (c_lo >= c_hi) return AHIP_OK;  // (more shards than blocks, or the chain has stepped over this shard's range)
  HIP_TRY(copy_on(dcand.p, cands.data(), (size_t)ncand * sizeof(BzCand), hipMemcpyHostToDevice, st));
  const u64 nblock_max = 100000ull * (u64)level;
  const u64 wstride = nblock_max / BZ_G + 2;
  static thread_local DevBuf dpre, dwalk, drank, dspans;
  // Candidates are taken in bounded batches in stream order (work memory O(batch x block size), about 5.4 MB per
  // block at level 9), following the chain of blocks exactly like decodeStream between them: a block ends where
  // the next magic starts.  A batch's blocks are placed and expanded before the next batch is decoded; nothing
  // behind the point where the chain stops is ever touched.
  const u64 per_block = nblock_max * 5 + (u64)BZ_SYM_CAP * 2 + BZ_CHUNKS * 530 + (u64)BZ_PARTS * 260 + (u64)BZ_TINV_WAVES * 1024 + sizeof(BzTables) + BZ_MAX_SELECTORS * 4 + wstride * (sizeof(BzWalk) + 4) + BZ_SPANS * sizeof(BzSpan) + BZ_MAX_SELECTORS + sizeof(BzResult) + 64;
  u64 batch_mem = 24ull << 30;  // (of 288 GB: every batch costs one latency-bound header + walk + rank round)
  {
    // ... but never more than half of what the device has free right now (other contexts, a caller's own tensors)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) batch_mem = std::min<u64>(batch_mem, std::max<u64>(free_b / 2, 256ull << 20));
    else (void)hipGetLastError();
  }
  if (const char *e = getenv("AHIP_BZ_BATCH_BYTES")) { if (atoll(e) > 0) batch_mem = (u64)atoll(e); }
  const u64 j50_per_block = bz_parallel_huffman() ? (u64)in_len * 8 / std::max<u32>(1, ncand) * 12 : 0;  // 6 tables x 2 bytes per bit
  u32 batch = (u32)std::min<u64>(c_hi - c_lo, std::max<u64>(4, batch_mem / (per_block + j50_per_block)));
  static thread_local DevBuf dsyms, dlist0, dchunks, dperms, dlists, dchoff, dtab, dj50, dgstart, dgcount, dpperms, dpcounts, dthist;
  // the work memory of one batch; a device that cannot spare it gets smaller batches, not an error
  auto reserve_batch = [&](u32 b) -> hipError_t {
    hipError_t e;
#define BZ_RES(buf, bytes) do { e = (buf).reserve((size_t)(bytes)); if (e != hipSuccess) return e; } while (0)
    BZ_RES(dtt, (size_t)b * nblock_max * 4);
    BZ_RES(dsel, (size_t)b * BZ_MAX_SELECTORS);
    BZ_RES(dtab, (size_t)b * sizeof(BzTables));
    BZ_RES(dgstart, (size_t)b * BZ_MAX_SELECTORS * 4);
    BZ_RES(dgcount, (size_t)b * 4);
    BZ_RES(dsyms, (size_t)b * BZ_SYM_CAP * 2);
    BZ_RES(dlist0, (size_t)b * 256);
    BZ_RES(dchunks, (size_t)b * BZ_CHUNKS * sizeof(BzChunk));
    BZ_RES(dperms, (size_t)b * BZ_CHUNKS * 256);
    BZ_RES(dlists, (size_t)b * BZ_CHUNKS * 256);
    BZ_RES(dchoff, (size_t)b * BZ_CHUNKS * 4);
    BZ_RES(dpperms, (size_t)b * BZ_PARTS * 256);
    BZ_RES(dpcounts, (size_t)b * BZ_PARTS * 4);
    BZ_RES(dthist, (size_t)b * BZ_TINV_WAVES * 256 * 4);
    BZ_RES(dpre, (size_t)b * nblock_max);
    BZ_RES(dwalk, (size_t)b * wstride * sizeof(BzWalk));
    BZ_RES(drank, (size_t)b * wstride * 4);
    BZ_RES(dspans, (size_t)b * BZ_SPANS * sizeof(BzSpan));
    BZ_RES(dres, (size_t)b * sizeof(BzResult));
    BZ_RES(doff, (size_t)b * 8);
#undef BZ_RES
    return hipSuccess;
  };
  for (;;) {
    const hipError_t e = reserve_batch(batch);
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    if (batch <= 1) return fail(AHIP_E_DEVICE, std::string("bzip2 work memory: ") + hipGetErrorString(e));
    batch = (batch + 1) / 2;
  }
  static thread_local bool crc_tables_up = false;  // (this context's copy of the MSB-first CRC tables: uploaded once)
  if (!crc_tables_up || !dcrc.p) {
    u32 table[256 + 64];
    for (u32 i = 0; i < 256; ++i) {
      u32 c = i << 24;
      for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? ((c << 1) ^ 0x04c11db7u) : (c << 1);
      table[i] = c;
    }
    auto mulmod = [](u32 a, u32 b) {
      u32 r = 0;
      for (int i = 31; i >= 0; --i) { r = (r << 1) ^ ((r >> 31) ? 0x04c11db7u : 0u); if ((b >> i) & 1) r ^= a; }
      return r;
    };
    u32 pw = 0x100;  // x^8
    for (int k = 0; k < 64; ++k) { table[256 + k] = pw; pw = mulmod(pw, pw); }
    HIP_TRY(dcrc.reserve(sizeof(table)));
    HIP_TRY(copy_on(dcrc.p, table, sizeof(table), hipMemcpyHostToDevice, st));
    crc_tables_up = true;
  }
  static thread_local DevBuf dcktab, dckacc;  // the reflected CRC's tables (bz_block_crc mirrors it)
  HIP_TRY(ck_prepare(dcktab, dckacc));
  const u32 wgrid = (u32)cdiv(wstride, 256);
  // workgroups of 256 threads per XCD (bz_walk): more hide latency, but threads beyond a block's ~7 000 sublists work on the
  // next blocks' vectors and the XCD's L2 holds about one
  static const u32 wpx = [] { const char *e = getenv("AHIP_BZ_WALK_WGS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 4096 ? (u32)v : 80u; }();
  (void)wgrid;
  static thread_local DevBuf dwq;  // the XCDs' work queues, one set for each of the two walks
  HIP_TRY(dwq.reserve(64));
  static thread_local DevBuf ddir;
  // chain state (decodeStream's loop: bzip2_chain.hpp drives the two phases below batch by batch)
  BzChain ch;
  bool over_cap = false;
  // the counting passes of the candidates [c0, c0 + nb): everything but the placement of the bytes
  auto decode_batch = [&](size_t c0, u32 nb, std::vector<BzResult> &res) -> int32_t {
    const BzCand *dc = dcand.as<BzCand>() + c0;
    // phase 1: the Huffman side of every block (one wave each) -> symbol streams; what the symbols mean by chunks
    const u64 bit0 = cands[c0].bit, bit1 = c0 + nb < ncand ? cands[c0 + nb].bit : (u64)in_len * 8;
    const u64 tstride = bit1 - bit0 + 64;
    // (12 bytes of jumps per bit of the batch: a batch whose few candidates lie gigabytes apart -- not a bzip2 stream --
    //  is left to the serial wave, which needs none)
    // (12 bytes of jumps per bit; when even that allocation fails the serial wave, which needs none, does the batch)
    bool jumps = bz_parallel_huffman() && tstride * 12 <= std::max<u64>(batch_mem, 4ull << 30);
    if (jumps && dj50.reserve((size_t)tstride * 6 * 2) != hipSuccess) { (void)hipGetLastError(); jumps = false; }
    // Everything up to the vector of the inverse transform, for the blocks [b0, b0 + n) of the batch on stream s.
    // (Five of these kernels are one wave or one workgroup per block walking something serially -- header and selectors,
    //  the group starts, the list scan, the counting sort, 4.9 of the 11.6 ms of a 64-block decode -- and the batch was
    //  tried in 2 / 4 / 8 parts on streams of their own, one part's serial kernels under another's wide ones: 12.1 / 16.3 /
    //  21.1 ms.  A serial kernel that shares its SIMD with a wide one is slowed by as much as it overlaps.)
    auto front = [&](hipStream_t s, u32 b0, u32 n) {
      const BzCand *dcs = dc + b0;
      BzTables *tabs = dtab.as<BzTables>() + b0;
      u8 *list0 = dlist0.as<u8>() + (size_t)b0 * 256, *sels = dsel.as<u8>() + (size_t)b0 * BZ_MAX_SELECTORS;
      BzResult *ress = dres.as<BzResult>() + b0;
      u16 *syms = dsyms.as<u16>() + (size_t)b0 * BZ_SYM_CAP;
      BzChunk *chunks = dchunks.as<BzChunk>() + (size_t)b0 * BZ_CHUNKS;
      u8 *perms = dperms.as<u8>() + (size_t)b0 * BZ_CHUNKS * 256, *lists = dlists.as<u8>() + (size_t)b0 * BZ_CHUNKS * 256;
      u32 *choff = dchoff.as<u32>() + (size_t)b0 * BZ_CHUNKS;
      u8 *b8 = dpre.as<u8>() + (size_t)b0 * nblock_max, *pperms = dpperms.as<u8>() + (size_t)b0 * BZ_PARTS * 256;
      u32 *pcounts = dpcounts.as<u32>() + (size_t)b0 * BZ_PARTS, *tts = dtt.as<u32>() + (size_t)b0 * nblock_max;
      if (jumps) {
        // headers and tables (one wave per block), the 50-code jump from every bit position (tiles, any number of workgroups),
        // the walk over the groups (one workgroup per block), the groups (one thread each); what is irregular, serially
        u64 widest = 0;
        for (u32 i = b0; i < b0 + n; ++i) widest = std::max<u64>(widest, (c0 + i + 1 < ncand ? cands[c0 + i + 1].bit : (u64)in_len * 8) - cands[c0 + i].bit);
        u32 *gstart = dgstart.as<u32>() + (size_t)b0 * BZ_MAX_SELECTORS, *gcount = dgcount.as<u32>() + b0;
        hipLaunchKernelGGL(bz_header, dim3(n), dim3(64), 0, s, d_in, (u64)in_len, dcs, n, tabs, list0, sels, ress);
        hipLaunchKernelGGL(bz_jump_tiles, dim3((u32)cdiv(widest, BZ_TW) + 1, n), dim3(512), 0, s, d_in, (u64)in_len, dcand.as<BzCand>(), ncand,
                           (u32)c0 + b0, tabs, dj50.as<u16>(), bit0, tstride);
        hipLaunchKernelGGL(bz_group_starts, dim3(n), dim3(512), 0, s, d_in, (u64)in_len, dcand.as<BzCand>(), ncand, (u32)c0 + b0, tabs, sels,
                           dj50.as<u16>(), bit0, tstride, gstart, gcount, ress);
        hipLaunchKernelGGL(bz_decode_groups, dim3(cdiv(BZ_MAX_SELECTORS, 256), n), dim3(256), 0, s, d_in, (u64)in_len, tabs, sels, gstart, gcount,
                           syms, ress);
        hipLaunchKernelGGL(bz_decode_block, dim3(n), dim3(64), 0, s, d_in, (u64)in_len, dcs, n, syms, list0, sels, ress, 1u, tabs);
      } else {
        hipLaunchKernelGGL(bz_decode_block, dim3(n), dim3(64), 0, s, d_in, (u64)in_len, dcs, n, syms, list0, sels, ress, 0u, tabs);
      }
      hipLaunchKernelGGL(bz_mtf_lanes<false>, dim3(BZ_CHUNKS, n), dim3(64), 0, s, syms, ress, (u32)level, chunks, perms, lists, choff, b8, pperms, pcounts);
      hipLaunchKernelGGL(bz_mtf_scan, dim3(n), dim3(64), 0, s, ress, dcs, n, (u32)level, chunks, perms, list0, lists, choff);
      hipLaunchKernelGGL(bz_mtf_lanes<true>, dim3(BZ_CHUNKS, n), dim3(64), 0, s, syms, ress, (u32)level, chunks, perms, lists, choff, b8, pperms, pcounts);
      // damaged blocks in which _getMtfVal fails: the reference goes on with -1 as a symbol -- its own loop, one lane each
      hipLaunchKernelGGL(bz_block_exact, dim3(n), dim3(64), 0, s, d_in, (u64)in_len, n, tabs, list0, sels, ress, b8, (u32)level);
      u32 *thist = dthist.as<u32>() + (size_t)b0 * BZ_TINV_WAVES * 256;
      hipLaunchKernelGGL(bz_tinv_hist, dim3(BZ_TINV_PARTS, n), dim3(1024), 0, s, tts, b8, (u32)level, dcs, ress, thist);  // (dpre: the bytes before the walk, the walk's output after it)
      hipLaunchKernelGGL(bz_tinv_cursors, dim3(n), dim3(256), 0, s, dcs, ress, thist);
      hipLaunchKernelGGL(bz_tinv_scatter, dim3(BZ_TINV_PARTS, n), dim3(1024), 0, s, tts, b8, (u32)level, dcs, ress, thist);
    };
    front(st, 0, nb);
    HIP_TRY(hipMemsetAsync(dwq.p, 0, 64, st));
    hipLaunchKernelGGL(bz_walk<false>, dim3(8 * wpx), dim3(256), 0, st, dtt.as<u32>(), (u32)level, dc, dres.as<BzResult>(),
                       dwalk.as<BzWalk>(), drank.as<u32>(), dpre.as<u8>(), nb, dwq.as<u32>() + 0);
    HIP_TRY(hipMemsetAsync(drank.p, 0xff, (size_t)nb * wstride * 4, st));  // ~0: not on the head's cycle
    hipLaunchKernelGGL(bz_rank, dim3(nb), dim3(1024), 0, st, (u32)level, dc, dres.as<BzResult>(), dwalk.as<BzWalk>(), drank.as<u32>());
    hipLaunchKernelGGL(bz_walk<true>, dim3(8 * wpx), dim3(256), 0, st, dtt.as<u32>(), (u32)level, dc, dres.as<BzResult>(),
                       dwalk.as<BzWalk>(), drank.as<u32>(), dpre.as<u8>(), nb, dwq.as<u32>() + 8);
    hipLaunchKernelGGL(bz_rle_scan, dim3(nb), dim3(1024), 0, st, (u32)level, dc, dres.as<BzResult>(), dpre.as<u8>(), dspans.as<BzSpan>());
    // blocks the parallel path handed back (BZ_ST_SERIAL): the reference loop, counting only (no slab)
    hipLaunchKernelGGL(bz_unbwt, dim3(cdiv(nb, 64)), dim3(64), 0, st, dtt.as<u32>(), (u32)level, nb, dc, (u8 *)nullptr, (u64)0,
                       dres.as<BzResult>(), dcrc.as<u32>(), (const u64 *)nullptr, (u8 *)nullptr);
    HIP_TRY(copy_on(res.data(), dres.p, (size_t)nb * sizeof(BzResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    return AHIP_OK;
  };
  // the bytes of the blocks the chain placed, and the CRCs of the parallel ones
  auto place_batch = [&](size_t c0, u32 nb, const std::vector<BzPlaced> &placed, const std::vector<BzResult> &, std::vector<BzResult> &res2) -> int32_t {
    const BzCand *dc = dcand.as<BzCand>() + c0;
    std::vector<u64> par_off(nb, ~0ull), ser_off(nb, ~0ull);
    bool any_serial = false;
    for (const BzPlaced &pl : placed) {
      if (pl.how == BZ_PL_PARALLEL) par_off[pl.cand] = pl.off;
      else { ser_off[pl.cand] = pl.off; any_serial = true; }
    }
    HIP_TRY(copy_on(doff.p, par_off.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bz_rle_expand, dim3(BZ_SPANS / 256, nb), dim3(256), 0, st, (u32)level, dc, dres.as<BzResult>(), dpre.as<u8>(),
                       dspans.as<BzSpan>(), doff.as<u64>(), d_out);
    hipLaunchKernelGGL(bz_block_crc, dim3((u32)cdiv(cdiv(nblock_max * 2, CK_SEG), 4), nb), dim3(256), 0, st, dc, dres.as<BzResult>(), doff.as<u64>(),
                       d_out, dcktab.as<u32>(), dcrc.as<u32>());
    if (any_serial) {  // what the serial inverse transform wrote -- also before it failed (BZ_PL_PARTIAL)
      HIP_TRY(ddir.reserve((size_t)nb * 8));
      HIP_TRY(copy_on(ddir.p, ser_off.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(bz_unbwt, dim3(cdiv(nb, 64)), dim3(64), 0, st, dtt.as<u32>(), (u32)level, nb, dc, (u8 *)nullptr, (u64)0,
                         dres.as<BzResult>(), dcrc.as<u32>(), ddir.as<u64>(), d_out);
    }
    HIP_TRY(copy_on(res2.data(), dres.p, (size_t)nb * sizeof(BzResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    return AHIP_OK;
  };
  {
    const int32_t rc = bz_chain_run(ch, cands.data(), ncand, c_lo, c_hi, batch, (u64)in_len, verify, (u64)out_cap, &over_cap, decode_batch, place_batch, peek);
    if (rc < 0) return rc;
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (ch.verdict == 1 && !ch.crc_stop && ch.fail_cand != ~(size_t)0) {
    // decodeStream returned false inside a block: where did its reader stand?  (bz_fail_cursor; the batch's tables are still there)
    static thread_local DevBuf dcur;
    HIP_TRY(dcur.reserve(8));
    hipLaunchKernelGGL(bz_fail_cursor, dim3(1), dim3(64), 0, st, d_in, (u64)in_len, (u32)(ch.fail_cand - ch.fail_c0), dtab.as<BzTables>(),
                       dlist0.as<u8>(), dsel.as<u8>(), dres.as<BzResult>(), (u32)level, dcur.as<u64>());
    u64 cur = 0;
    HIP_TRY(copy_on(&cur, dcur.p, 8, hipMemcpyDeviceToHost, st));
    if (cur) ch.fail_bit = cur;
  }
#ifdef AHIP_BZ_PROFILE
  {
    unsigned long long pr[8] = {0}, zero[8] = {0};
    HIP_TRY(hipMemcpyFromSymbol(pr, HIP_SYMBOL(bz_prof), sizeof(pr)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(bz_prof), zero, sizeof(zero)));
    fprintf(stderr, "[ahip] bzip2 Huffman pass, cycles summed over blocks: header %llu tables %llu | window setup %llu chain %llu cuts+stores %llu | windows %llu symbols %llu\n",
            pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6]);
  }
#endif
  if (over_cap) { if (out_len) *out_len = ch.total; return fail(AHIP_E_CAP, "output buffer too small"); }
  if (ch.verdict == AHIP_E_UNSUPPORTED) return fail(AHIP_E_UNSUPPORTED, "randomised bzip2 block");
  if (sh) {  // the caller merges the shards: the end-of-stream CRC is checked there
    sh->nblocks = ch.folded; sh->fold = ch.combined; sh->saw_eos = ch.saw_eos; sh->eos_stored = ch.eos_stored;
    sh->stopped = ch.stopped || ch.verdict != AHIP_OK;
    sh->next = ch.next;
    BzChain part = ch;
    part.saw_eos = false;
    u64 n = 0;
    const int32_t v = bz_chain_finish(part, verify, &n);
    if (out_len) *out_len = (size_t)n;
    return v;
  }
  u64 n = 0, sb = 0;
  const int32_t v = bz_chain_finish(ch, verify, &n, &sb);
  if (out_len) *out_len = (size_t)n;
  // (what the bit reader has pulled from the InputStream when decodeStream returns -- true: behind the last block or marker;
  //  false: where the failing check stood -- whole bytes, bz2_bit_reader.dart:12-44)
  g_consumed = std::min<u64>((u64)in_len, (sb + 7) / 8);
  return v;
}

int32_t ahip_bzip2_decode_device(const void *d_in, size_t in_len, int32_t verify, void *d_out, size_t out_cap,
                                 size_t *out_len, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  // (everything is ordered on the caller's stream; the call still returns with the verdict, i.e. after the stream has caught
  //  up: the chain of blocks is followed on the host between the kernel phases)
  hipStream_t st = (hipStream_t)stream;
  if (out_len) *out_len = 0;
  u8 hdr[4] = {0, 0, 0, 0};
  if (in_len >= 4 || in_len > 0) {
    int32_t rc = ensure_init();
    if (rc != AHIP_OK) return rc;
    HIP_TRY(copy_on(hdr, d_in, in_len < 4 ? in_len : 4, hipMemcpyDeviceToHost, st));
  }
  return bzip2_device_impl(hdr, (const u8 *)d_in, in_len, verify, (u8 *)d_out, out_cap, out_len, nullptr, st);
}

// One RANK's part of a bzip2 stream (one process per GPU; the one-process form is ahip_bzip2_decode_shards, which does the
// same per shard and merges): the blocks among the block-magic candidates [K rank / world, K (rank + 1) / world) -- or, from
// != ~0, from candidate `from` on (the merge's second try: the ranks in front ended somewhere else, on a false magic inside a
// block's data).  info[8] = {blocks folded, CRC fold of those blocks, met the end-of-stream block, its stored CRC, the stream
// ended inside this range, first candidate the chain started at, candidate it expects next, 0}: what the caller gathers and
// merges in rank order exactly like decodeStream (archive_amd/sharding.py::merge_bzip2_ranks; ref bzip2_decoder.dart:20-88).
int32_t ahip_bzip2_decode_range_device(const void *d_in, size_t in_len, int32_t verify, uint32_t rank, uint32_t world, uint64_t from,
                                       void *d_out, size_t out_cap, size_t *out_len, uint64_t *info, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (!info || !world || rank >= world) return fail(AHIP_E_ARG, "bzip2 range: rank / world / info");
  hipStream_t st = (hipStream_t)stream;
  if (out_len) *out_len = 0;
  for (int i = 0; i < 8; ++i) info[i] = 0;
  u8 hdr[4] = {0, 0, 0, 0};
  if (in_len > 0) {
    int32_t rc = ensure_init();
    if (rc != AHIP_OK) return rc;
    HIP_TRY(copy_on(hdr, d_in, in_len < 4 ? in_len : 4, hipMemcpyDeviceToHost, st));
  }
  BzShard sh;
  sh.index = rank; sh.count = world; sh.from = from;
  const int32_t rc = bzip2_device_impl(hdr, (const u8 *)d_in, in_len, verify, (u8 *)d_out, out_cap, out_len, &sh, st);
  info[0] = sh.nblocks; info[1] = sh.fold; info[2] = sh.saw_eos ? 1 : 0; info[3] = sh.eos_stored; info[4] = sh.stopped ? 1 : 0;
  info[5] = sh.first; info[6] = sh.next;
  return rc;
}

int32_t ahip_bzip2_decode(const uint8_t *in, size_t in_len, int32_t verify, uint8_t *out, size_t out_cap,
                          size_t *out_len) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (out_len) *out_len = 0;
  static thread_local DevBuf din, dout;
  // the header-only outcomes need no device
  if (in_len <= 4 || in[0] != 'B' || in[1] != 'Z' || in[2] != 'h' || in[3] < '1' || in[3] > '9')
    return bzip2_device_impl(in, nullptr, in_len, verify, nullptr, 0, out_len);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  HIP_TRY(din.reserve(in_len + 16));
  HIP_TRY(hipMemcpy(din.p, in, in_len, hipMemcpyHostToDevice));
  HIP_TRY(dout.reserve(out_cap + 16));
  size_t n = 0;
  rc = bzip2_device_impl(in, din.as<u8>(), in_len, verify, dout.as<u8>(), out_cap, &n);
  if (out_len) *out_len = n;
  if ((rc == AHIP_OK || rc == AHIP_FALSE) && n && n <= out_cap) HIP_TRY(hipMemcpy(out, dout.p, n, hipMemcpyDeviceToHost));
  return rc;
}

int32_t ahip_bzip2_decode_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, size_t in_len, int32_t verify,
                                 void *const *d_out, const size_t *out_cap, size_t *out_len, uint64_t *offsets, int32_t *status);

size_t ahip_decode_bound(const uint8_t *in, size_t in_len) {
  // walk the members through their BC subfields (FEXTRA {'B','C',2,0,BSIZE-1}); anything else: unknown
  size_t pos = 0, total = 0, members = 0;
  while (pos < in_len) {
    if (in_len - pos < 18 || in[pos] != 0x1f || in[pos + 1] != 0x8b || in[pos + 2] != 8) break;
    size_t next = 0;
    if ((in[pos + 3] & 4) && pos + 12 <= in_len) {
      const size_t xlen = in[pos + 10] | ((size_t)in[pos + 11] << 8);
      for (size_t q = pos + 12; q + 4 <= pos + 12 + xlen && q + 4 <= in_len;) {
        const size_t slen = in[q + 2] | ((size_t)in[q + 3] << 8);
        if (in[q] == 'B' && in[q + 1] == 'C' && slen == 2 && q + 6 <= in_len) { next = pos + (in[q + 4] | ((size_t)in[q + 5] << 8)) + 1; break; }
        q += 4 + slen;
      }
    }
    if (!next) {  // no BC: fine only if this is the one and only member and small enough for ISIZE to be the size
      if (members == 0 && in_len < (1ull << 30)) { next = in_len; } else return 0;
    }
    if (next > in_len || next < pos + 18) return 0;
    total += (size_t)in[next - 4] | ((size_t)in[next - 3] << 8) | ((size_t)in[next - 2] << 16) | ((size_t)in[next - 1] << 24);
    ++members;
    pos = next;
  }
  return pos == in_len ? total : 0;
}

size_t ahip_deflate_bound(size_t in_len) {
  const size_t chunks = (in_len + DF_CHUNK - 1) / DF_CHUNK;
  return in_len + chunks * 16 + 64;
}

// Deflate on device memory.  Returns the compressed size through *out_len.
// open: the input is a shard of a longer one and not its last (DeflateParams::open): no final block, ends on a byte boundary
static int32_t deflate_device_impl(const u8 *d_in, size_t n, int level, int window_bits, u8 *d_out, size_t cap, size_t *out_len,
                                   hipStream_t st, bool open = false) {
  static thread_local DevBuf b_match, b_tok, b_ntok, b_slabs, b_csize, b_coff;
  if (out_len) *out_len = 0;
  if (level < 0 || level > 9 || window_bits < 9 || window_bits > 15) return AHIP_OK;  // the reference's _init fails silently: no output (deflate.dart:105-115)
  if (n == 0 && open) return AHIP_OK;  // an empty shard in the middle adds nothing
  if (n == 0) {  // reference: one fixed-Huffman block holding only the end-of-block code (level >= 1), or an empty stored block
    const u8 fixed_empty[2] = {0x03, 0x00}, stored_empty[5] = {0x01, 0x00, 0x00, 0xff, 0xff};
    const u8 *src = level == 0 ? stored_empty : fixed_empty;
    const size_t len = level == 0 ? 5 : 2;
    if (cap < len) { if (out_len) *out_len = len; return fail(AHIP_E_CAP, "output buffer too small"); }
    HIP_TRY(hipMemcpyAsync(d_out, src, len, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (out_len) *out_len = len;
    return AHIP_OK;
  }
  DeflateParams P;
  P.n = n;
  P.chunks = (u32)((n + DF_CHUNK - 1) / DF_CHUNK);
  P.lazy = level >= 4 ? 1u : 0u;
  P.store = level == 0 ? 1u : 0u;
  P.max_cmp = 258;
  P.max_dist = (1u << window_bits) - 262;
  P.nice = level <= 3 ? DF_CAP : (level <= 6 ? 128u : 258u);
  P.open = open ? 1u : 0u;
  HIP_TRY(b_match.reserve(n * 4 + 64 + (size_t)P.chunks * 32 + 64));
  HIP_TRY(b_tok.reserve(n * 4 + 64));
  HIP_TRY(b_ntok.reserve((size_t)P.chunks * 4));
  HIP_TRY(b_slabs.reserve((size_t)P.chunks * DF_SLAB));
  HIP_TRY(b_csize.reserve((size_t)P.chunks * 4));
  HIP_TRY(b_coff.reserve((size_t)P.chunks * 8 + 8));
  if (!P.store)
  {
    // levels 1-3: 4096 x 2 entries, 256 threads, three workgroups per CU.  Levels 4-9: ONE workgroup of 512 threads per
    // CU (the same 8 waves as two workgroups of 256) owns the whole LDS: 8192 x 4 entries of 4-byte strings (every window
    // position indexed), 16384 of 8-byte and 8192 of 16-byte strings; the levels differ in nice_length (P.nice).
    // A position only sees the strings of EARLIER steps (plus one of its own step), so the step has to stay well
    // below the window: small windows (windowBits 9..12) take smaller workgroups.
    // The grid: as many workgroups as are resident at once, each taking a run of consecutive chunks -- from its second chunk on
    // a workgroup finds the window's tables already there (deflate_match_kernel).  AHIP_DF_RUNS=0 (tests): one chunk a
    // workgroup, every chunk inserts its history; =N: N workgroups -- the output must be the same byte for byte.
    auto launch = [&](auto kernel, u32 threads) -> hipError_t {
      static thread_local int cus = 0;
      if (!cus) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256; }
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
      const char *e = getenv("AHIP_DF_RUNS");  // tests: 0 = one chunk a workgroup, N = N workgroups (long runs on a small input)
      const u32 grid = !e ? std::min<u32>(P.chunks, (u32)cus * (u32)per_cu) : (atoi(e) <= 0 ? P.chunks : std::min<u32>(P.chunks, (u32)atoi(e)));
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, st, d_in, P, b_match.as<u32>());
      return hipGetLastError();
    };
    if (window_bits <= 10) HIP_TRY(launch(deflate_match_kernel<12, 4, 0, 0, 64>, 64));
    else if (window_bits <= 12) HIP_TRY(launch(deflate_match_kernel<12, 4, 12, 0, 256>, 256));
    else if (level <= 3) HIP_TRY(launch(deflate_match_kernel<12, 2>, 256));
    // levels 4-7: 1 024 positions a step (16 waves hide the compare rounds' LDS latency: 1 GiB 32.9 -> 23 ms; a position sees
    // fewer candidates closer than a step: log text +4.1 % instead of +3.4 % over the reference); 8-9 keep 512
    else if (level <= 7 && !getenv("AHIP_DF_SUB512")) HIP_TRY(launch(deflate_match_kernel<13, 4, 14, 13, 1024>, 1024));
    else HIP_TRY(launch(deflate_match_kernel<13, 4, 14, 13, 512>, 512));
  }
#ifdef AHIP_PROFILE
  if (!P.store && getenv("AHIP_DEBUG")) {
    std::vector<u32> pc((size_t)P.chunks * 8);
    (void)hipMemcpy(pc.data(), b_match.as<u32>() + n + 16, pc.size() * 4, hipMemcpyDeviceToHost);
    double s[8] = {0};
    for (u32 c = 0; c < P.chunks; ++c) for (int k = 0; k < 8; ++k) s[k] += pc[(size_t)c * 8 + k] * 16.0;
    fprintf(stderr, "[ahip] match kernel cycles per chunk (wave 0): pre-sync %.0f  barriers+insert %.0f  verify %.0f  compare %.0f  "
            "post-sync total %.0f  search loop %.0f  (compare, first pass alone %.0f)\n", s[0] / P.chunks, s[1] / P.chunks, s[2] / P.chunks, s[3] / P.chunks,
            s[4] / P.chunks, s[5] / P.chunks, s[6] / P.chunks);
  }
#endif
  hipLaunchKernelGGL(deflate_parse_kernel, dim3(P.chunks), dim3(64), 0, st, d_in, P, b_match.as<u32>(), b_tok.as<u32>(),
                     b_ntok.as<u32>());
  hipLaunchKernelGGL(deflate_encode_kernel, dim3(P.chunks), dim3(256), 0, st, d_in, P, b_tok.as<u32>(), b_ntok.as<u32>(),
                     b_slabs.as<u8>(), b_csize.as<u32>());
#ifdef AHIP_PROFILE
  if (!P.store && getenv("AHIP_DEBUG")) {
    (void)hipStreamSynchronize(st);
    std::vector<u8> sl((size_t)P.chunks * DF_SLAB);
    (void)hipMemcpy(sl.data(), b_slabs.p, sl.size(), hipMemcpyDeviceToHost);
    double s4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 c = 0; c < P.chunks; ++c) { const u32 *pc = (const u32 *)(sl.data() + (size_t)c * DF_SLAB + DF_SLAB - 32); for (int k = 0; k < 8; ++k) s4[k] += pc[k] * 16.0; }
    fprintf(stderr, "[ahip] encode kernel cycles per chunk: zero+histogram %.0f  trees+header %.0f (literal/length tree %.0f, run-length scan of the lengths %.0f, "
            "code-length tree %.0f, header emission %.0f)  size pass %.0f  token rounds %.0f\n",
            s4[0] / P.chunks, s4[1] / P.chunks, s4[4] / P.chunks, s4[5] / P.chunks, s4[6] / P.chunks, s4[7] / P.chunks, s4[2] / P.chunks, s4[3] / P.chunks);
  }
#endif
  // sizes -> offsets -> gather, all on the stream; the host reads the total (8 bytes) once at the end
  hipLaunchKernelGGL(deflate_offsets_kernel, dim3(1), dim3(1024), 0, st, b_csize.as<u32>(), P.chunks, b_coff.as<u64>(),
                     b_coff.as<u64>() + P.chunks);
  hipLaunchKernelGGL(deflate_concat_kernel, dim3(P.chunks), dim3(256), 0, st, b_slabs.as<u8>(), b_csize.as<u32>(),
                     b_coff.as<u64>(), d_out, (u64)cap);
  u64 total = 0;
  HIP_TRY(hipMemcpyAsync(&total, b_coff.as<u64>() + P.chunks, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  if (out_len) *out_len = total;
  if (total > cap) return fail(AHIP_E_CAP, "output buffer too small");
  return AHIP_OK;
}

int32_t ahip_deflate_raw_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, void *d_out, size_t out_cap,
                                size_t *out_len, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  return deflate_device_impl((const u8 *)d_in, in_len, level, window_bits, (u8 *)d_out, out_cap, out_len, (hipStream_t)stream);
}

// One RANK's piece of a sharded Deflate (one process per GPU; the one-process form is ahip_deflate_shards): the piece is
// compressed on its own; every piece but the last (`last` = 0) ends with the byte-aligning empty stored block the reference
// itself writes as a flush marker (deflate.dart:219) instead of a final block, so the pieces laid end to end in rank order
// are ONE raw DEFLATE stream of the whole input.  crc32 (may be NULL): CRC-32 of the piece's INPUT, for a gzip trailer.
int32_t ahip_deflate_piece_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, int32_t last, void *d_out,
                                  size_t out_cap, size_t *out_len, uint32_t *crc32, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  rc = deflate_device_impl((const u8 *)d_in, in_len, level, window_bits, (u8 *)d_out, out_cap, out_len, (hipStream_t)stream, !last);
  if (rc == AHIP_OK && crc32) rc = crc32_device_impl((const u8 *)d_in, in_len, 0, crc32, (hipStream_t)stream);
  return rc;
}

// host-pointer Deflate; the checksums of the input are taken from its device copy
static int32_t deflate_host_impl(const uint8_t *in, size_t in_len, int32_t level, int32_t window_bits, uint8_t *out,
                                 size_t out_cap, size_t *out_len, uint32_t *crc32, uint32_t *adler32) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (out_len) *out_len = 0;
  if (crc32) *crc32 = 0;
  if (adler32) *adler32 = 1;
  // the reference's Deflate fails silently on invalid parameters (deflate.dart:105-115): no output, its crc32 stays 0.
  // The zlib encoder computes the Adler-32 of the input BEFORE it calls Deflate.stream (_zlib_encoder_web.dart:62-72),
  // so that trailer is the real checksum even then.
  const bool noop = window_bits < 9 || window_bits > 15 || level < 0 || level > 9;
  if (noop && !adler32) return AHIP_OK;
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  static thread_local DevBuf din, dout;
  HIP_TRY(din.reserve(in_len + 16));
  if (in_len) HIP_TRY(hipMemcpy(din.p, in, in_len, hipMemcpyHostToDevice));
  size_t produced = 0;
  if (!noop) {
    const size_t bound = ahip_deflate_bound(in_len);
    HIP_TRY(dout.reserve(bound));
    rc = deflate_device_impl(din.as<u8>(), in_len, level, window_bits, dout.as<u8>(), bound, &produced, nullptr);
    if (rc != AHIP_OK) return rc;
    if (out_len) *out_len = produced;
    if (crc32) { rc = crc32_device_impl(din.as<u8>(), in_len, 0, crc32, nullptr); if (rc != AHIP_OK) return rc; }
  }
  if (adler32) { rc = adler32_device_impl(din.as<u8>(), in_len, 1, adler32, nullptr); if (rc != AHIP_OK) return rc; }
  if (produced > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
  if (produced) HIP_TRY(hipMemcpy(out, dout.p, produced, hipMemcpyDeviceToHost));
  return AHIP_OK;
}

int32_t ahip_deflate_raw(const uint8_t *in, size_t in_len, int32_t level, int32_t window_bits, uint8_t *out,
                         size_t out_cap, size_t *out_len, uint32_t *crc32) {
  return deflate_host_impl(in, in_len, level, window_bits, out, out_cap, out_len, crc32, nullptr);
}

int32_t ahip_crc32_device(const void *d_data, size_t len, uint32_t crc, uint32_t *out, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (!out) return fail(AHIP_E_ARG, "out == NULL");
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  return crc32_device_impl((const u8 *)d_data, len, crc, out, (hipStream_t)stream);
}

int32_t ahip_adler32_device(const void *d_data, size_t len, uint32_t adler, uint32_t *out, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (!out) return fail(AHIP_E_ARG, "out == NULL");
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  return adler32_device_impl((const u8 *)d_data, len, adler, out, (hipStream_t)stream);
}

int32_t ahip_gzip_encode(const uint8_t *in, size_t in_len, int32_t level, int32_t window_bits, uint32_t mtime, uint8_t *out,
                         size_t out_cap, size_t *out_len) {
  if (out_len) *out_len = 0;
  if (out_cap < 18) { if (out_len) *out_len = ahip_deflate_bound(in_len) + 18; return fail(AHIP_E_CAP, "output buffer too small"); }
  size_t clen = 0;
  uint32_t crc = 0;
  int32_t rc = deflate_host_impl(in, in_len, level, window_bits, out + 10, out_cap - 18, &clen, &crc, nullptr);
  if (out_len) *out_len = clen + 18;
  if (rc != AHIP_OK) return rc;
  const uint8_t h[10] = {0x1f, 0x8b, 8, 0, (uint8_t)mtime, (uint8_t)(mtime >> 8), (uint8_t)(mtime >> 16), (uint8_t)(mtime >> 24), 0, 0xff};
  memcpy(out, h, 10);
  uint8_t *t = out + 10 + clen;
  for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(((uint32_t)in_len) >> (8 * k)); }
  return AHIP_OK;
}

int32_t ahip_zlib_encode(const uint8_t *in, size_t in_len, int32_t level, int32_t window_bits, uint8_t *out, size_t out_cap,
                         size_t *out_len) {
  if (out_len) *out_len = 0;
  if (out_cap < 6) { if (out_len) *out_len = ahip_deflate_bound(in_len) + 6; return fail(AHIP_E_CAP, "output buffer too small"); }
  size_t clen = 0;
  uint32_t a = 1;
  int32_t rc = deflate_host_impl(in, in_len, level, window_bits, out + 2, out_cap - 6, &clen, nullptr, &a);
  if (out_len) *out_len = clen + 6;
  if (rc != AHIP_OK) return rc;
  // CMF / FLG exactly as the reference builds them (_zlib_encoder_web.dart:42-63): CINFO = windowBits - 8,
  // FLEVEL = FDICT = 0, FCHECK the smallest value that makes the pair a multiple of 31 (`78 01` for 15)
  const int wb = window_bits < 0 ? 0 : (window_bits > 15 ? 15 : window_bits);
  const uint32_t cmf = (uint32_t)(((wb - 8) << 4) | 8) & 0xff;
  uint32_t flg = 0;
  while ((cmf * 256 + flg) % 31 != 0) ++flg;
  out[0] = (uint8_t)cmf; out[1] = (uint8_t)flg;
  uint8_t *t = out + 2 + clen;
  t[0] = (uint8_t)(a >> 24); t[1] = (uint8_t)(a >> 16); t[2] = (uint8_t)(a >> 8); t[3] = (uint8_t)a;
  return AHIP_OK;
}

// Device-resident encoders: the DEFLATE stream is written straight behind the header bytes in d_out, the checksum of
// the trailer comes from the device copy of the input; header and trailer are a few bytes copied from the host.
static int32_t framed_encode_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, const uint8_t *head, size_t head_len,
                                    bool gzip, void *d_out, size_t out_cap, size_t *out_len, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const size_t tail_len = gzip ? 8 : 4;
  if (out_len) *out_len = 0;
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  if (out_cap < head_len + tail_len) { if (out_len) *out_len = ahip_deflate_bound(in_len) + head_len + tail_len; return fail(AHIP_E_CAP, "output buffer too small"); }
  hipStream_t st = (hipStream_t)stream;
  size_t clen = 0;
  u8 *o = (u8 *)d_out;
  if (!(window_bits < 9 || window_bits > 15 || level < 0 || level > 9)) {  // (the reference's silent no-op otherwise)
    rc = deflate_device_impl((const u8 *)d_in, in_len, level, window_bits, o + head_len, out_cap - head_len - tail_len, &clen, st);
    if (out_len) *out_len = clen + head_len + tail_len;
    if (rc != AHIP_OK) return rc;
  }
  // gzip: the CRC-32 is the Deflate object's (0 when it refused its parameters); zlib: the Adler-32 is computed over the
  // input before Deflate.stream is called, whatever Deflate then does (_zlib_encoder_web.dart:62-72)
  uint32_t ck = gzip ? 0u : 1u;
  if (!gzip || !(window_bits < 9 || window_bits > 15 || level < 0 || level > 9)) {
    rc = gzip ? crc32_device_impl((const u8 *)d_in, in_len, 0, &ck, st) : adler32_device_impl((const u8 *)d_in, in_len, 1, &ck, st);
    if (rc != AHIP_OK) return rc;
  }
  uint8_t t[8];
  if (gzip) for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(ck >> (8 * k)); t[4 + k] = (uint8_t)(((uint32_t)in_len) >> (8 * k)); }
  else { t[0] = (uint8_t)(ck >> 24); t[1] = (uint8_t)(ck >> 16); t[2] = (uint8_t)(ck >> 8); t[3] = (uint8_t)ck; }
  HIP_TRY(hipMemcpyAsync(o, head, head_len, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(o + head_len + clen, t, tail_len, hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (out_len) *out_len = clen + head_len + tail_len;
  return AHIP_OK;
}

int32_t ahip_gzip_encode_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, uint32_t mtime, void *d_out,
                                size_t out_cap, size_t *out_len, void *stream) {
  const uint8_t h[10] = {0x1f, 0x8b, 8, 0, (uint8_t)mtime, (uint8_t)(mtime >> 8), (uint8_t)(mtime >> 16), (uint8_t)(mtime >> 24), 0, 0xff};
  return framed_encode_device(d_in, in_len, level, window_bits, h, 10, true, d_out, out_cap, out_len, stream);
}

int32_t ahip_zlib_encode_device(const void *d_in, size_t in_len, int32_t level, int32_t window_bits, void *d_out, size_t out_cap,
                                size_t *out_len, void *stream) {
  const int wb = window_bits < 0 ? 0 : (window_bits > 15 ? 15 : window_bits);
  const uint32_t cmf = (uint32_t)(((wb - 8) << 4) | 8) & 0xff;
  uint32_t flg = 0;
  while ((cmf * 256 + flg) % 31 != 0) ++flg;
  const uint8_t h[2] = {(uint8_t)cmf, (uint8_t)flg};
  return framed_encode_device(d_in, in_len, level, window_bits, h, 2, false, d_out, out_cap, out_len, stream);
}

int32_t ahip_gzip_plan_create(const void *d_in, size_t in_len, void *stream, ahip_gzip_plan **plan) {
  if (!plan) return fail(AHIP_E_ARG, "plan == NULL");
  *plan = nullptr;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  ahip_gzip_plan *pl = new ahip_gzip_plan();
  pl->d_in = (const u8 *)d_in;
  pl->in_len = in_len;
  rc = plan_build(pl, false, (hipStream_t)stream);
  if (rc != AHIP_OK) { delete pl; return rc; }
  *plan = pl;
  return AHIP_OK;
}

int32_t ahip_gzip_plan_info(const ahip_gzip_plan *plan, uint64_t *members, uint64_t *out_bytes,
                            uint64_t *payload_bytes) {
  if (!plan) return fail(AHIP_E_ARG, "plan == NULL");
  if (members) *members = plan->sum.members;
  if (out_bytes) *out_bytes = plan->sum.total_out;
  if (payload_bytes) *payload_bytes = plan->sum.payload_bytes;
  return AHIP_OK;
}

int32_t ahip_gzip_plan_run(ahip_gzip_plan *plan, void *d_out, size_t out_cap, void *stream) {
  if (!plan) return fail(AHIP_E_ARG, "plan == NULL");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  return plan_run(plan, (u8 *)d_out, out_cap, (hipStream_t)stream);
}

int32_t ahip_gzip_plan_status(ahip_gzip_plan *plan, size_t *out_len) {
  if (!plan) return fail(AHIP_E_ARG, "plan == NULL");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (out_len) *out_len = plan->sum.total_out;
  if (!plan->ran) return fail(AHIP_E_ARG, "plan has not been run");
  bool needs = false;
  int32_t rc = plan_verdict(plan, plan->run_stream, &needs);  // waits for the run on the stream it was enqueued on
  if (rc != AHIP_OK) return rc;
  if (needs) return fail(AHIP_E_UNSUPPORTED, "member index (BC/ISIZE) disagrees with the data; use ahip_gzip_decode_device");
  if (plan->sum.range_error) return AHIP_RANGE;
  if (plan->sum.tail_pos != plan->in_len) return AHIP_FALSE;
  return AHIP_OK;
}

// Diagnostics (not part of the drop-in surface): per-member results of the last run, as
// 20 u32 words each {end_pos lo/hi, out_len lo/hi, status, blocks, windows, rounds, fallbacks, partial, cyc[8], tok_words lo/hi}.
int32_t ahip_debug_plan_results(ahip_gzip_plan *plan, uint32_t *host_words, size_t max_members, size_t *n_members) {
  if (!plan || !plan->ran) return fail(AHIP_E_ARG, "plan has not been run");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  size_t M = plan->sum.members < max_members ? plan->sum.members : max_members;
  if (n_members) *n_members = M;
  static_assert(sizeof(MemberResult) == 80, "MemberResult layout");
  if (M) HIP_TRY(hipMemcpy(host_words, plan->results.p, M * sizeof(MemberResult), hipMemcpyDeviceToHost));
  return AHIP_OK;
}

void ahip_gzip_plan_destroy(ahip_gzip_plan *plan) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  delete plan;
}

// ---- one long stream, several ranks (SplitState above; include/archive_hip.h has the protocol) ----
// what ahip_stream_split_create decides (host only): the cuts of the stream and this rank's range of them
static void split_setup(SplitState &h, const void *d_in, size_t in_len, size_t data_off, uint32_t rank, uint32_t world, hipStream_t st) {
  h.d_in = (const u8 *)d_in; h.n = in_len; h.off = data_off; h.rank = rank; h.world = world; h.st = st;
  // Several ranks: fewer chunks per GPU, and a chunk is one wave's serial work -- 16 KiB cuts make about every block of a
  // zlib stream a chunk of its own (1 GiB of text on 8 ranks: 3.9 ms against 6.0 with 32 KiB cuts, profiles/r06_stream_split.md);
  // one rank keeps the single-device path's rule.  AHIP_SM_CHUNK overrides.
  h.cb = sm_chunk_bytes();
  const u64 len = in_len - data_off;
  if (!getenv("AHIP_SM_CHUNK")) {
    // about 8 000 chunks a rank (sm_inflate's rule) between 48 and 128 KiB; a rank with less than that to do: 16 KiB with several
    // ranks (measured: 24 and 32 KiB are worse than both 16 and 48 -- the same block starts, longer finder parts)
    const u64 want = (len / world / 8192 + 4095) & ~4095ull;
    h.cb = want >= (40ull << 10) ? (want > (128ull << 10) ? (128ull << 10) : (want < h.cb ? h.cb : want)) : (world > 1 ? (16ull << 10) : h.cb);
  }
  while (((len + h.cb - 1) / h.cb + world - 1) / world > 32768) h.cb *= 2;  // (a RANK's chunks are the grid.y of its per-chunk kernels)
  h.n_cuts = (u32)((len + h.cb - 1) / h.cb);
  h.k0 = (u32)((u64)h.n_cuts * rank / world);
  h.k1 = (u32)((u64)h.n_cuts * (rank + 1) / world);
  h.eligible = !getenv("AHIP_NO_SM") && len >= sm_min_bytes() && h.n_cuts >= 4;
}

int32_t ahip_stream_split_create(const void *d_in, size_t in_len, size_t data_off, uint32_t rank, uint32_t world, void *stream,
                                 ahip_stream_split **split) {
  if (!split) return fail(AHIP_E_ARG, "split == NULL");
  *split = nullptr;
  if (!world || rank >= world || world > 4096) return fail(AHIP_E_ARG, "stream split: rank / world");
  if (data_off > in_len) return fail(AHIP_E_ARG, "stream split: data_off behind the input");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  ahip_stream_split *sp = new ahip_stream_split();
  split_setup(sp->s, d_in, in_len, data_off, rank, world, (hipStream_t)stream);
  *split = sp;
  return AHIP_OK;
}

int32_t ahip_stream_split_candidates(ahip_stream_split *split, uint64_t *cand, size_t cap, size_t *n) {
  if (!split || !n) return fail(AHIP_E_ARG, "stream split: NULL argument");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  SplitState &h = split->s;
  if (h.phase < 1) {
    if (h.eligible) {
      int32_t rc = ensure_init();
      if (rc != AHIP_OK) return rc;
      rc = split_candidates(&h);
      if (rc != AHIP_OK) return rc;
    } else {
      h.own.clear();
      if (h.rank == 0) h.own.push_back(h.off * 8);
      h.phase = 1;
    }
  }
  *n = h.own.size();
  if (h.own.size() > cap) return fail(AHIP_E_CAP, "stream split: candidate buffer too small");
  if (cand) std::copy(h.own.begin(), h.own.end(), cand);
  return AHIP_OK;
}

int32_t ahip_stream_split_size(ahip_stream_split *split, const uint64_t *all_cand, size_t n_all, uint64_t *results, size_t cap_words,
                               int32_t *handled) {
  if (!split || !all_cand || !handled) return fail(AHIP_E_ARG, "stream split: NULL argument");
  *handled = 0;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  SplitState &h = split->s;
  if (h.phase != 1) return fail(AHIP_E_ARG, "stream split: size before candidates (or twice)");
  if (cap_words < 4 * h.own.size() || (!results && !h.own.empty())) return fail(AHIP_E_CAP, "stream split: result buffer too small");
  bool ok = false;
  int32_t rc = AHIP_OK;
  if (h.eligible) {
    rc = ensure_init();
    if (rc == AHIP_OK) rc = split_size(&h, all_cand, n_all, &ok);
  } else {
    h.cand.assign(all_cand, all_cand + n_all);
    h.own_res.assign(h.own.size(), MemberResult{});
    h.phase = 2;
  }
  if (rc != AHIP_OK) return rc;
  for (size_t i = 0; i < h.own.size(); ++i) {
    const MemberResult &r = h.own_res[i];
    results[4 * i] = r.status; results[4 * i + 1] = r.out_len; results[4 * i + 2] = r.end_pos; results[4 * i + 3] = r.blocks;
  }
  *handled = ok ? 1 : 0;
  return AHIP_OK;
}

int32_t ahip_stream_split_chain(ahip_stream_split *split, const uint64_t *all_results, size_t n_all, int32_t *handled, uint64_t *rank_off,
                                uint64_t *rank_len, uint64_t *total_out, uint64_t *end_pos) {
  if (!split || !all_results || !handled) return fail(AHIP_E_ARG, "stream split: NULL argument");
  *handled = 0;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  SplitState &h = split->s;
  bool ok = false;
  int32_t rc = split_chain(&h, all_results, n_all, &ok);
  if (rc != AHIP_OK) return rc;
  ok = ok && h.eligible;
  if (rank_off) *rank_off = ok ? h.base_out : 0;
  if (rank_len) *rank_len = ok ? h.out_len : 0;
  if (total_out) *total_out = ok ? h.total_out : 0;
  if (end_pos) *end_pos = ok ? h.end_pos : 0;
  *handled = ok ? 1 : 0;
  return AHIP_OK;
}

size_t ahip_stream_split_map_bytes(void) { return (size_t)SPLIT_MAP_ELEMS * 2; }

// Diagnostics / tests: the chain walk of ahip_stream_split_chain on plain arrays, no device and no handle (a rank that owns
// the candidates [c0, c1) of `n`).  out[0..5] = handled, slice offset, slice length, total, end position, chunks of the chain
// that are the rank's.
int32_t ahip_debug_stream_split_chain(const uint64_t *cand, const uint64_t *results, size_t n, uint32_t c0, uint32_t c1, uint64_t *out) {
  if (!cand || !results || !out || c0 > c1 || c1 > n) return fail(AHIP_E_ARG, "stream split: bad arguments");
  SplitState h;
  h.cand.assign(cand, cand + n);
  h.c0 = c0; h.c1 = c1;
  h.own.assign(cand + c0, cand + c1);
  h.own_res.resize(c1 - c0);
  for (uint32_t i = c0; i < c1; ++i) { h.own_res[i - c0].status = (u32)results[4 * i]; h.own_res[i - c0].out_len = results[4 * i + 1]; h.own_res[i - c0].end_pos = results[4 * i + 2]; }
  h.phase = 2;
  bool ok = false;
  const int32_t rc = split_chain(&h, results, n, &ok);
  out[0] = ok ? 1 : 0; out[1] = h.base_out; out[2] = h.out_len; out[3] = h.total_out; out[4] = h.end_pos; out[5] = h.chain.size();
  return rc;
}

int32_t ahip_stream_split_resolve(ahip_stream_split *split, void *d_map) {
  if (!split || !d_map) return fail(AHIP_E_ARG, "stream split: NULL argument");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  return split_resolve(&split->s, (u16 *)d_map);
}

int32_t ahip_stream_split_finish(ahip_stream_split *split, const void *d_maps, void *d_out, size_t out_cap, size_t *out_len, int32_t *handled) {
  if (!split || !d_maps || !handled) return fail(AHIP_E_ARG, "stream split: NULL argument");
  *handled = 0;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  bool ok = false;
  rc = split_finish(&split->s, (const u16 *)d_maps, (u8 *)d_out, out_cap, out_len, &ok);
  *handled = ok ? 1 : 0;
  return rc;
}

void ahip_stream_split_destroy(ahip_stream_split *split) {
  if (!split) return;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  split->s.release();
  delete split;
}

// Shared body of the gzip entry points.  host_in may be NULL (device API): then a tail that is
// not a gzip member is inspected through a small D2H copy.
static int32_t gzip_decode_impl(const u8 *host_in, const u8 *d_in, size_t in_len, int verify, int raw, u8 *d_out,
                                size_t out_cap, bool out_is_growable, DevBuf *grow, size_t *out_len, hipStream_t st) {
  ahip_gzip_plan pl;
  pl.d_in = d_in;
  pl.in_len = in_len;
  int32_t rc = plan_build(&pl, false, st);
  if (rc != AHIP_OK) return rc;
  for (int attempt = 0;; ++attempt) {
    // a header or trailer on the member chain runs past the end of the input: the reference
    // throws RangeError out of decodeBytes, whatever was decoded before is lost
    if (pl.sum.range_error) return AHIP_RANGE;
    if (out_is_growable) {
      HIP_TRY(grow->reserve(pl.sum.total_out + 16));
      d_out = grow->as<u8>();
      out_cap = grow->cap;
    }
    if (out_len) *out_len = pl.sum.total_out;
    if (pl.sum.total_out > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
    rc = plan_run(&pl, d_out, out_cap, st);
    if (rc != AHIP_OK) return rc;
    bool needs = false;
    rc = plan_verdict(&pl, st, &needs);
    if (rc != AHIP_OK) return rc;
    if (!needs) break;
    if (attempt) return fail(AHIP_E_DEVICE, "internal: sizing run did not settle the member index");
    rc = plan_build(&pl, true, st);  // BC/ISIZE lied: take sizes from the data
    if (rc != AHIP_OK) return rc;
  }
  if (pl.sum.range_error) return AHIP_RANGE;
  u64 committed = pl.sum.total_out;
  if (out_len) *out_len = committed;
  g_consumed = in_len;  // (the member loop runs until isEOS)
  if (pl.sum.tail_pos >= in_len) return AHIP_OK;
  // The bytes at tail_pos are not a gzip header: the reference hands the rest of the stream to
  // the zlib decoder (little-endian stream, so its Adler-32 is read byte-swapped).
  std::vector<u8> tail_host;
  const u8 *h = host_in;
  if (!h) {
    tail_host.resize(in_len - pl.sum.tail_pos);
    HIP_TRY(copy_on(tail_host.data(), d_in + pl.sum.tail_pos, tail_host.size(), hipMemcpyDeviceToHost, st));
    h = tail_host.data() - pl.sum.tail_pos;
  }
  // `1f 8b` and then the end of the input: _readHeader has matched the 16-bit signature and its next readByte()
  // throws (_gzip_decoder_web.dart:99-107) -- the zlib decoder is never asked
  if (in_len - pl.sum.tail_pos == 2 && h[pl.sum.tail_pos] == 0x1f && h[pl.sum.tail_pos + 1] == 0x8b) return AHIP_RANGE;
  // cheap pre-check of the first header so the common "trailing garbage" case needs no kernels
  if (!raw) {
    u64 p = pl.sum.tail_pos;
    if (p + 2 > in_len) return AHIP_RANGE;
    u32 cmf = h[p], flg = h[p + 1];
    if ((cmf & 8) != 8 || ((cmf * 256) + flg) % 31 != 0) { g_consumed = p + 2; return AHIP_FALSE; }
  }
  if (!out_is_growable) {
    // fixed caller buffer: decode the tail into scratch, then append what fits
    DevBuf scratch;
    u64 tail_committed = 0;
    rc = zlib_stream_device(h, d_in, in_len, pl.sum.tail_pos, false, verify, raw, scratch, &tail_committed, st);
    if (rc == AHIP_OK || rc == AHIP_FALSE) {
      if (out_len) *out_len = committed + tail_committed;
      if (committed + tail_committed > out_cap) { scratch.release(); return fail(AHIP_E_CAP, "output buffer too small"); }
      if (tail_committed) HIP_TRY(copy_on(d_out + committed, scratch.p, tail_committed, hipMemcpyDeviceToDevice, st));
    }
    scratch.release();
    return rc;
  }
  rc = zlib_stream_device(h, d_in, in_len, pl.sum.tail_pos, false, verify, raw, *grow, &committed, st);
  if (out_len) *out_len = committed;
  return rc;
}

int32_t ahip_gzip_decode_device(const void *d_in, size_t in_len, void *d_out, size_t out_cap, size_t *out_len,
                                void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  return gzip_decode_impl(nullptr, (const u8 *)d_in, in_len, 0, 0, (u8 *)d_out, out_cap, false, nullptr, out_len,
                          (hipStream_t)stream);
}

// host buffers -> this thread's device -> host buffers (the whole stream, or one shard of it)
// the stream of this thread's device context (a worker's own non-blocking stream; the default stream otherwise)
static thread_local hipStream_t g_ctx_stream = nullptr;
static int32_t gzip_decode_host_impl(const uint8_t *in, size_t in_len, int32_t verify, int32_t raw, uint8_t *out, size_t out_cap,
                                     size_t *out_len) {
  static thread_local DevBuf din, dout;
  hipStream_t st = g_ctx_stream;
  HIP_TRY(din.reserve(in_len + 16));
  if (in_len) HIP_TRY(hipMemcpyAsync(din.p, in, in_len, hipMemcpyHostToDevice, st));
  size_t produced = 0;
  int32_t rc = gzip_decode_impl(in, din.as<u8>(), in_len, verify, raw, nullptr, 0, true, &dout, &produced, st);
  if (out_len) *out_len = produced;
  if (rc == AHIP_OK || rc == AHIP_FALSE) {
    if (produced > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
    if (produced) HIP_TRY(hipMemcpyAsync(out, dout.p, produced, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  return rc;
}

// ---- one process, several GPUs ----
// ahip_init_devices() starts one worker thread per device; all library state is thread_local, so a worker IS a
// device context (its own scratch pools, streams, plans).  A multi-member stream whose members all carry the BGZF
// `BC` subfield is partitioned on the host -- contiguous member ranges balanced on compressed bytes, exactly
// archive_amd.sharding.partition_members -- and every worker uploads, decodes and downloads only its slice; the
// slices' output offsets are the prefix sums of the members' ISIZE trailers, checked against what was produced.
// Anything else (no BC, one device, a shard that does not end cleanly) takes the single-device path, which has the
// reference's exact semantics for every input.
namespace {
struct Worker {
  int device = 0;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, quit = false;
  // every job has a number; wait() is for the job the CALLING thread submitted last (a second submitter -- nothing in the
  // library does that today: callers hold g_mu or g_shards_mu -- neither resets the first one's completion nor is woken by it)
  u64 submitted = 0, completed = 0;
  const u64 id = next_worker_id();  // (never reused: a worker that takes a dead one's address does not inherit its tickets)
  static u64 next_worker_id() { static std::atomic<u64> n{0}; return ++n; }
  void loop() {
    (void)hipSetDevice(device);
    tl_free_bufs_at_exit = true;  // din / dout / token scratch / ... of this context go back to the device when the thread ends
    // its own non-blocking stream: copies and kernels of different contexts overlap (H2D of one slice, decode of
    // another, D2H of a third) instead of queueing on the legacy default stream
    hipStream_t own = nullptr;
    if (hipStreamCreateWithFlags(&own, hipStreamNonBlocking) == hipSuccess) g_ctx_stream = own;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) break;
      lk.unlock();
      job();
      lk.lock();
      has_job = false; ++completed;
      cv.notify_all();
    }
    for (auto &b : g_pool) (void)hipFree(b.p);  // this thread's pool
    g_pool.clear();
    if (own) (void)hipStreamDestroy(own);
    g_ctx_stream = nullptr;
  }
  void submit(std::function<void()> f) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !has_job; });  // (one job at a time: never replace a closure the worker is still running)
    job = std::move(f); has_job = true;
    my_ticket() = ++submitted;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const u64 t = my_ticket();
    cv.wait(lk, [&] { return completed >= t; });
  }
  u64 &my_ticket() {  // (per calling thread and worker)
    static thread_local std::vector<std::pair<u64, u64>> mine;
    for (auto &e : mine) if (e.first == id) return e.second;
    mine.push_back({id, 0});
    return mine.back().second;
  }
};
std::vector<std::unique_ptr<Worker>> g_workers;  // guarded by g_mu; the set does not change while g_shards_mu is held
// A shards call owns the workers while its shards decode -- with g_mu RELEASED, so that the rest of the library stays
// usable from other threads meanwhile.  Lock order: g_mu, then g_shards_mu (never the other way round: the shards call
// lets go of g_shards_mu before it takes g_mu again).
std::mutex g_shards_mu;
// One device, host pointers: contexts on the SAME device that only exist to overlap PCIe traffic with the decode
// (started on first use by a large stream when ahip_init_devices() selected nothing; AHIP_HOST_PIPE=0 turns it off,
// =k asks for k contexts)
std::vector<std::unique_ptr<Worker>> g_pipe;  // guarded by g_mu
}  // namespace
namespace {
void stop_set(std::vector<std::unique_ptr<Worker>> &set) {
  for (auto &w : set) {
    { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_all(); }
    if (w->th.joinable()) w->th.join();
  }
  set.clear();
}
void stop_workers() {
  std::lock_guard<std::mutex> use(g_shards_mu);  // (a shards call in flight on another thread finishes first)
  stop_set(g_workers);
  stop_set(g_pipe);
}
// Worker threads must be gone before the process tears its statics down (a joinable std::thread in a static is
// std::terminate, and a thread blocked inside the HIP runtime at exit hangs the process): stop them at exit.
void stop_at_exit() {
  static bool registered = false;
  if (registered) return;
  registered = true;
  atexit([] { std::lock_guard<std::recursive_mutex> lk(g_mu); stop_workers(); });  // (RCCL communicators are left to the process teardown)
}

struct GzMember { size_t begin, end; uint32_t isize; };
// members of a stream in which EVERY member has a BC subfield and the chain ends exactly at the end of the input
bool walk_bc_members(const uint8_t *in, size_t n, std::vector<GzMember> &ms) {
  size_t pos = 0;
  while (pos < n) {
    if (n - pos < 26 || in[pos] != 0x1f || in[pos + 1] != 0x8b || in[pos + 2] != 8 || !(in[pos + 3] & 4)) return false;
    const size_t xlen = in[pos + 10] | ((size_t)in[pos + 11] << 8);
    size_t next = 0;
    for (size_t q = pos + 12; q + 4 <= pos + 12 + xlen && q + 6 <= n;) {
      const size_t slen = in[q + 2] | ((size_t)in[q + 3] << 8);
      if (in[q] == 'B' && in[q + 1] == 'C' && slen == 2) { next = pos + (in[q + 4] | ((size_t)in[q + 5] << 8)) + 1; break; }
      q += 4 + slen;
    }
    if (!next || next > n || next < pos + 26) return false;
    ms.push_back({pos, next, (uint32_t)in[next - 4] | ((uint32_t)in[next - 3] << 8) | ((uint32_t)in[next - 2] << 16) | ((uint32_t)in[next - 1] << 24)});
    pos = next;
  }
  return !ms.empty();
}

// Members WITHOUT size hints (an ordinary `cat a.gz b.gz`: the reference's header parser skips FEXTRA anyway,
// _gzip_decoder_web.dart:119-122): where they begin and end is only known after inflating them.  One device does that
// sizing pass for the whole stream (member index + a store-less run of the tokenizer: plan_build), the table it yields
// is what the contexts then partition -- each decodes its slice from its own upload.  Only a clean chain qualifies
// (every member complete and sound, nothing behind the last one, no member long enough for the chunked path); anything
// else is left to the exact single-context path.
bool size_members_on_device(const uint8_t *in, size_t in_len, std::vector<GzMember> &ms) {
  static thread_local DevBuf din;
  if (din.reserve(in_len + 16) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemcpy(din.p, in, in_len, hipMemcpyHostToDevice) != hipSuccess) return false;
  ahip_gzip_plan pl;
  pl.d_in = din.as<u8>();
  pl.in_len = in_len;
  if (plan_build(&pl, false, nullptr) != AHIP_OK) return false;
  const u64 M = pl.sum.members;
  if (!pl.sum.first_is_gzip || M < 2 || pl.sum.range_error || pl.sum.tail_pos != in_len || !pl.big.empty() || !pl.sized) return false;
  std::vector<MemberDesc> md(M);
  std::vector<u64> cp(pl.K);
  std::vector<u32> ex(M);
  if (copy_on(md.data(), pl.members.p, M * sizeof(MemberDesc), hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
      copy_on(cp.data(), pl.cand_pos.p, (size_t)pl.K * 8, hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
      copy_on(ex.data(), pl.expect_status.p, M * 4, hipMemcpyDeviceToHost, nullptr) != hipSuccess) return false;
  ms.clear();
  for (u64 m = 0; m < M; ++m) {
    if (ex[m] != MS_OK || md[m].pad >= pl.K || md[m].expect_end == POS_UNKNOWN || md[m].out_limit > 0xffffffffull) return false;
    const u64 begin = cp[md[m].pad], end = md[m].expect_end + 8;
    if (end > in_len || end <= begin || (m && begin != ms.back().end)) return false;
    ms.push_back({(size_t)begin, (size_t)end, (uint32_t)md[m].out_limit});
  }
  return !ms.empty() && ms.front().begin == 0 && ms.back().end == in_len;
}

// returns true when the sharded path produced the final answer in *rc_out.  The stream is cut into `per` slices per
// context (contiguous member ranges balanced on compressed bytes, sharding.partition_members); context w takes slices
// w, w + W, ... one after the other, so that while it downloads one slice the others upload and decode theirs.
bool gzip_decode_sharded(std::vector<std::unique_ptr<Worker>> &set, size_t per, const uint8_t *in, size_t in_len, int32_t verify, uint8_t *out,
                         size_t out_cap, size_t *out_len, int32_t *rc_out, bool may_size) {
  const size_t W = set.size();
  if (W < 2 || in_len < (4u << 20)) return false;
  std::vector<GzMember> ms;
  if (!walk_bc_members(in, in_len, ms)) {
    ms.clear();
    if (!may_size || !size_members_on_device(in, in_len, ms)) return false;
  }
  if (ms.size() < 2 * W) return false;
  size_t S = W * per;
  while (S > W && ms.size() < 2 * S) S -= W;
  std::vector<size_t> bounds{0};
  size_t acc = 0, r = 1;
  for (size_t i = 0; i < ms.size(); ++i) {
    acc += ms[i].end - ms[i].begin;
    while (r < S && (unsigned __int128)acc * S >= (unsigned __int128)in_len * r && bounds.size() < S) { bounds.push_back(i + 1); ++r; }
  }
  while (bounds.size() < S) bounds.push_back(ms.size());
  bounds.push_back(ms.size());
  std::vector<size_t> o_off(S + 1, 0);
  for (size_t w = 0; w < S; ++w) {
    size_t sum = 0;
    for (size_t i = bounds[w]; i < bounds[w + 1]; ++i) sum += ms[i].isize;
    o_off[w + 1] = o_off[w] + sum;
  }
  // (ISIZE is not trusted: a total that does not fit the caller's buffer sends the call to the exact path, which reports
  // the real size)
  if (o_off[S] > out_cap) return false;
  std::vector<int32_t> rcs(S, AHIP_OK);
  std::vector<size_t> got(S, 0);
  for (size_t w = 0; w < W; ++w) {
    std::vector<size_t> mine;
    for (size_t q = w; q < S; q += W) if (bounds[q] < bounds[q + 1]) mine.push_back(q);
    set[w]->submit([&, mine] {
      for (size_t q : mine) {
        const size_t lo = bounds[q], hi = bounds[q + 1];
        rcs[q] = gzip_decode_host_impl(in + ms[lo].begin, ms[hi - 1].end - ms[lo].begin, verify, 0, out + o_off[q], o_off[q + 1] - o_off[q], &got[q]);
        if (rcs[q] != AHIP_OK) break;
      }
    });
  }
  for (size_t w = 0; w < W; ++w) set[w]->wait();
  for (size_t q = 0; q < S; ++q)
    if (rcs[q] != AHIP_OK || got[q] != o_off[q + 1] - o_off[q]) return false;  // lying ISIZE, damaged member, a reference into another slice: exact path
  if (out_len) *out_len = o_off[S];
  *rc_out = AHIP_OK;
  g_consumed = in_len;
  return true;
}
// contexts on the current device for the host-pointer pipeline (lazily, once)
bool ensure_pipe() {
  if (!g_pipe.empty()) return true;
  stop_at_exit();
  int k = 4;
  if (const char *e = getenv("AHIP_HOST_PIPE")) k = atoi(e);
  if (k < 2) return false;
  if (k > 16) k = 16;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  for (int i = 0; i < k; ++i) {
    g_pipe.emplace_back(new Worker());
    Worker *w = g_pipe.back().get();
    w->device = dev;
    w->th = std::thread([w] { w->loop(); });
  }
  return true;
}
}  // namespace

int32_t ahip_init_devices(uint64_t device_mask) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  stop_workers();
  std::vector<int> devs;
  for (int d = 0; d < n && d < 64; ++d)
    if ((device_mask >> d) & 1) devs.push_back(d);
  if (device_mask == 0) return fail(AHIP_E_ARG, "empty device mask");
  if (devs.empty()) return fail(AHIP_E_DEVICE, "no device of the mask is present");
  if (const char *e = getenv("AHIP_FAKE_DEVICES")) {  // tests on a one-GPU box: several contexts on the same device
    const int k = atoi(e);
    while ((int)devs.size() < k && devs.size() < 16) devs.push_back(devs[0]);
  }
  HIP_TRY(hipSetDevice(devs[0]));  // the calling thread keeps working on the first one
  stop_at_exit();
  if (devs.size() > 1)
    for (int d : devs) {
      g_workers.emplace_back(new Worker());
      Worker *w = g_workers.back().get();
      w->device = d;
      w->th = std::thread([w] { w->loop(); });
    }
  return AHIP_OK;
}

static int32_t g_last_shards = 1;
static std::atomic<int32_t> g_bz_reruns{0};  // shards of ahip_bzip2_decode_shards run a second time (tests)
int32_t ahip_debug_bz_reruns(void) { return g_bz_reruns.load(); }
int32_t ahip_debug_last_shards(void) { return g_last_shards; }

// ---- device-resident shards on several GPUs + the size exchange ----
// The one exchange step of the path: every device contributes the number of bytes its shard produced, all of them
// end up with all the sizes, the exclusive prefix sum is a shard's offset in the logical output.  Between distinct
// devices that is an all-gather of one uint64 per device over RCCL (ncclAllGather on the devices' own buffers, one
// communicator per device made by ncclCommInitAll; librccl is loaded at run time so that the library does not depend
// on it); several contexts on ONE device (AHIP_FAKE_DEVICES), a missing librccl or AHIP_NO_RCCL=1 take host sums.
#include <dlfcn.h>
namespace {
struct Rccl {
  void *h = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool tried = false, ok = false;
  std::vector<int> devs;      // the devices the communicators below were made for
  std::vector<void *> comms;
  std::vector<u64 *> send, recv;
  std::vector<hipStream_t> streams;  // one non-blocking stream per device: the exchange waits for ITS work only
  bool load() {
    if (tried) return ok;
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return false;
    CommInitAll = (int (*)(void **, int, const int *))dlsym(h, "ncclCommInitAll");
    CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    ok = CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
    return ok;
  }
  void drop() {
    for (size_t i = 0; i < comms.size(); ++i) {
      (void)hipSetDevice(devs[i]);
      if (comms[i]) CommDestroy(comms[i]);
      if (send[i]) (void)hipFree(send[i]);
      if (recv[i]) (void)hipFree(recv[i]);
      if (i < streams.size() && streams[i]) (void)hipStreamDestroy(streams[i]);
    }
    comms.clear(); send.clear(); recv.clear(); devs.clear(); streams.clear();
  }
  bool prepare(const std::vector<int> &d) {
    if (!load()) return false;
    if (d == devs) return true;
    drop();
    comms.assign(d.size(), nullptr); send.assign(d.size(), nullptr); recv.assign(d.size(), nullptr); streams.assign(d.size(), nullptr);
    devs = d;
    if (CommInitAll(comms.data(), (int)d.size(), d.data()) != 0) { comms.assign(d.size(), nullptr); drop(); return false; }
    for (size_t i = 0; i < d.size(); ++i) {
      if (hipSetDevice(d[i]) != hipSuccess || hipMalloc((void **)&send[i], 8) != hipSuccess ||
          hipMalloc((void **)&recv[i], 8 * d.size()) != hipSuccess ||
          hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) { drop(); return false; }
    }
    return true;
  }
  // sizes[i] of device devs[i] -> every device holds all of them; `all` = what device 0 ended up with
  bool all_gather(const std::vector<u64> &sizes, std::vector<u64> &all) {
    const size_t n = devs.size();
    // everything on the devices' own streams: the size goes up, the all-gather follows it in stream order, device 0's
    // copy of the result comes down behind it -- no device-wide barrier, only these streams are waited for
    for (size_t i = 0; i < n; ++i) {
      if (hipSetDevice(devs[i]) != hipSuccess) return false;
      if (hipMemcpyAsync(send[i], &sizes[i], 8, hipMemcpyHostToDevice, streams[i]) != hipSuccess) return false;
    }
    constexpr int kNcclUint64 = 5;
    if (GroupStart() != 0) return false;
    bool good = true;
    for (size_t i = 0; i < n; ++i) {
      (void)hipSetDevice(devs[i]);
      good = good && AllGather(send[i], recv[i], 1, kNcclUint64, comms[i], streams[i]) == 0;
    }
    if (GroupEnd() != 0 || !good) return false;
    all.resize(n);
    (void)hipSetDevice(devs[0]);
    if (hipMemcpyAsync(all.data(), recv[0], 8 * n, hipMemcpyDeviceToHost, streams[0]) != hipSuccess) return false;
    for (size_t i = 0; i < n; ++i) {
      if (hipSetDevice(devs[i]) != hipSuccess || hipStreamSynchronize(streams[i]) != hipSuccess) return false;
    }
    return true;
  }
};
Rccl g_rccl;      // guarded by g_mu
int32_t g_last_exchange = 0;
}  // namespace

void rccl_drop() { if (g_rccl.ok) g_rccl.drop(); }
int32_t ahip_debug_last_exchange(void) { return g_last_exchange; }
int32_t ahip_debug_last_chunks(void) { return g_last_chunks; }

// ---- the one exchange of the sharded calls: sizes -> offsets (exclusive prefix sum), offsets[n] = the total ----
// an all-gather of one uint64 per device over RCCL when the shards sit on distinct devices, host sums otherwise
static void exchange_sizes(uint32_t n_shards, const int32_t *devices, const size_t *got, uint64_t *offsets, int cur_device) {
  std::vector<u64> sizes(n_shards), all;
  for (u32 s = 0; s < n_shards; ++s) sizes[s] = got[s];
  g_last_exchange = 0;
  bool distinct = true;
  std::vector<int> devs(devices, devices + n_shards);
  for (u32 a = 0; a < n_shards; ++a) for (u32 b = a + 1; b < n_shards; ++b) if (devs[a] == devs[b]) distinct = false;
  const char *no = getenv("AHIP_NO_RCCL");
  if (distinct && !(no && no[0] == '1') && g_rccl.prepare(devs) && g_rccl.all_gather(sizes, all)) g_last_exchange = 1;
  else all = sizes;
  (void)hipSetDevice(cur_device);
  u64 acc = 0;
  for (u32 s = 0; s < n_shards; ++s) { offsets[s] = acc; acc += all[s]; }
  offsets[n_shards] = acc;
}
// which worker context takes shard s (-1: the calling thread); false = a device ahip_init_devices() did not select
static bool shard_workers(uint32_t n_shards, const int32_t *devices, int cur, std::vector<int> &wk) {
  wk.assign(n_shards, -1);
  for (u32 s = 0; s < n_shards; ++s) {
    if (g_workers.empty()) {
      if (devices[s] != cur) return false;
      continue;
    }
    // contexts of the same device (AHIP_FAKE_DEVICES) are dealt out round robin
    int pick = -1, seen = 0;
    for (size_t w = 0; w < g_workers.size(); ++w)
      if (g_workers[w]->device == devices[s]) { if (pick < 0 || seen <= (int)(s % g_workers.size())) pick = (int)w; ++seen; }
    if (pick < 0) return false;
    wk[s] = pick;
  }
  return true;
}
// run_shard(s) for every shard: on the calling thread when no workers exist, else every worker takes its shards in order
// and the workers run side by side.  The workers are this call's until they are done (g_shards_mu); g_mu is let go
// meanwhile -- the shards run in the workers' own thread-local contexts, nothing of the process-wide state g_mu guards
// is touched -- and taken again afterwards.
// `only` != ~0u: just that shard (a sharded call that runs one of its shards a second time).
static void run_shards(uint32_t n_shards, const std::vector<int> &wk, const std::function<void(u32)> &run_shard,
                       std::unique_lock<std::recursive_mutex> &lk, u32 only = ~0u) {
  if (g_workers.empty()) {
    for (u32 s = 0; s < n_shards; ++s) if (only == ~0u || s == only) run_shard(s);
    return;
  }
  std::unique_lock<std::mutex> use(g_shards_mu);
  std::vector<Worker *> ws;
  for (auto &w : g_workers) ws.push_back(w.get());
  lk.unlock();
  std::vector<Worker *> busy;
  for (size_t w = 0; w < ws.size(); ++w) {
    std::vector<u32> mine;
    for (u32 s = 0; s < n_shards; ++s) if (wk[s] == (int)w && (only == ~0u || s == only)) mine.push_back(s);
    if (mine.empty()) continue;
    ws[w]->submit([mine, &run_shard] { for (u32 s : mine) run_shard(s); });
    busy.push_back(ws[w]);
  }
  for (Worker *w : busy) w->wait();
  use.unlock();
  lk.lock();
}

// Deflate of ONE input cut into shards (ref: deflate.dart:219 -- the byte-aligning empty stored block
// `_trStoredBlock(0, 0, false)` the reference itself emits as a flush marker is what lets independently compressed pieces
// be spliced): shard s (any length; the caller cuts, multiples of 32 KiB lose nothing) is compressed by the context of
// its device into d_out[s]; every shard but the last ends with that marker instead of a final block, so the
// concatenation of the shards' outputs at offsets[] is one raw DEFLATE stream of the concatenated input.  A match
// never reaches into another shard (its 32 KiB of history start afresh).  crc32s (may be NULL): CRC-32 of every
// shard's INPUT, taken on its device -- what a gzip trailer is combined from.  The exchange is the size all-gather.
int32_t ahip_deflate_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, const size_t *in_len, int32_t level,
                            int32_t window_bits, void *const *d_out, const size_t *out_cap, size_t *out_len, uint64_t *offsets,
                            uint32_t *crc32s) {
  std::unique_lock<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  if (n_shards == 0 || !devices || !d_in || !in_len || !d_out || !out_cap || !out_len || !offsets) return fail(AHIP_E_ARG, "NULL shard table");
  int cur = 0;
  HIP_TRY(hipGetDevice(&cur));
  std::vector<int> wk;
  if (!shard_workers(n_shards, devices, cur, wk)) return fail(AHIP_E_ARG, "shard on a device that ahip_init_devices() did not select");
  std::vector<int32_t> rcs(n_shards, AHIP_OK);
  std::vector<std::string> errs(n_shards);
  std::vector<size_t> got(n_shards, 0);
  // (the reference's silent no-op on invalid parameters: every shard produces nothing)
  std::function<void(u32)> run_shard = [&](u32 s) {
    rcs[s] = deflate_device_impl((const u8 *)d_in[s], in_len[s], level, window_bits, (u8 *)d_out[s], out_cap[s], &got[s], g_ctx_stream, s + 1 < n_shards);
    if (rcs[s] == AHIP_OK && crc32s) rcs[s] = crc32_device_impl((const u8 *)d_in[s], in_len[s], 0, &crc32s[s], g_ctx_stream);
    if (rcs[s] < 0) errs[s] = g_err;
  };
  run_shards(n_shards, wk, run_shard, lk);
  int32_t worst = AHIP_OK;
  for (u32 s = 0; s < n_shards; ++s) {
    out_len[s] = got[s];
    if (rcs[s] < 0 && worst >= 0) { worst = rcs[s]; g_err = "shard " + std::to_string(s) + ": " + errs[s]; }
  }
  exchange_sizes(n_shards, devices, got.data(), offsets, cur);
  return worst;
}

// BZip2 blocks are independent once their bit positions are known (ref: bzip2_decoder.dart:20-88 walks them one after the
// other): shard s decodes the blocks among candidates [K s / n, K (s + 1) / n) of the stream -- d_in[s] is a copy of the
// WHOLE compressed stream on its device (blocks start at arbitrary bit positions; the compressed stream is the small side)
// -- into d_out[s].  The shards are merged in stream order exactly like decodeStream: the first verdict that is not OK,
// or the end-of-stream block, ends the stream (shards behind it count for nothing: out_len 0); block CRCs fold into the
// stream CRC linearly, so every shard reports its own fold.  offsets[] from the size exchange.
int32_t ahip_bzip2_decode_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, size_t in_len, int32_t verify,
                                 void *const *d_out, const size_t *out_cap, size_t *out_len, uint64_t *offsets, int32_t *status) {
  std::unique_lock<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  if (n_shards == 0 || !devices || !d_in || !d_out || !out_cap || !out_len || !offsets) return fail(AHIP_E_ARG, "NULL shard table");
  int cur = 0;
  HIP_TRY(hipGetDevice(&cur));
  std::vector<int> wk;
  if (!shard_workers(n_shards, devices, cur, wk)) return fail(AHIP_E_ARG, "shard on a device that ahip_init_devices() did not select");
  std::vector<int32_t> rcs(n_shards, AHIP_OK);
  std::vector<std::string> errs(n_shards);
  std::vector<size_t> got(n_shards, 0);
  std::vector<BzShard> shs(n_shards);
  std::vector<u64> from(n_shards, ~0ull);
  std::function<void(u32)> run_shard = [&](u32 s) {
    u8 hdr[4] = {0, 0, 0, 0};
    if (in_len && copy_on(hdr, d_in[s], in_len < 4 ? in_len : 4, hipMemcpyDeviceToHost, g_ctx_stream) != hipSuccess) { rcs[s] = AHIP_E_DEVICE; errs[s] = "header read-back"; return; }
    shs[s] = BzShard{};
    shs[s].index = s; shs[s].count = n_shards; shs[s].from = from[s];
    rcs[s] = bzip2_device_impl(hdr, (const u8 *)d_in[s], in_len, verify, (u8 *)d_out[s], out_cap[s], &got[s], &shs[s], g_ctx_stream);
    if (rcs[s] < 0) errs[s] = g_err;
  };
  run_shards(n_shards, wk, run_shard, lk);
  // merge in stream order
  int32_t worst = AHIP_OK;
  bool ended = false, saw_eos = false;
  u32 combined = 0, eos_stored = 0;
  u64 stands = ~0ull;  // the candidate the chain of the shards so far expects next
  for (u32 s = 0; s < n_shards; ++s) {
    if (ended) { got[s] = 0; if (status) status[s] = AHIP_OK; out_len[s] = 0; continue; }
    // Shard s started blindly at candidate K s / n.  The scan also lists false magics inside blocks' data, which the
    // chain steps over: when the shards in front end somewhere else (a false magic on the boundary), what this shard
    // decoded began at a non-block -- it is run again from where the chain really stands, like decodeStream would.
    if (s > 0 && stands != ~0ull && rcs[s] != AHIP_E_DEVICE && shs[s].first != stands) {
      from[s] = stands;
      got[s] = 0;
      run_shards(n_shards, wk, run_shard, lk, s);
      g_bz_reruns++;
    }
    stands = shs[s].next;
    out_len[s] = got[s];
    if (status) status[s] = rcs[s];
    const u32 r = (u32)(shs[s].nblocks & 31);
    combined = (r ? ((combined << r) | (combined >> (32 - r))) : combined) ^ shs[s].fold;
    if (rcs[s] != AHIP_OK) {
      if (rcs[s] < 0) g_err = "shard " + std::to_string(s) + ": " + errs[s];
      worst = rcs[s];
      ended = true;
    } else if (shs[s].stopped) {
      ended = true;
      saw_eos = shs[s].saw_eos; eos_stored = shs[s].eos_stored;
    }
  }
  if (saw_eos && verify && eos_stored != combined && worst == AHIP_OK) worst = AHIP_FALSE;
  exchange_sizes(n_shards, devices, got.data(), offsets, cur);
  return worst;
}

int32_t ahip_gzip_decode_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, const size_t *in_len,
                                void *const *d_out, const size_t *out_cap, size_t *out_len, uint64_t *offsets, int32_t *status) {
  std::unique_lock<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  if (n_shards == 0 || !devices || !d_in || !in_len || !d_out || !out_cap || !out_len || !offsets) return fail(AHIP_E_ARG, "NULL shard table");
  int cur = 0;
  HIP_TRY(hipGetDevice(&cur));
  // a shard runs in the context of its device: worker k of ahip_init_devices (several shards of one device take turns),
  // or this thread when no workers exist and the shard sits on the current device
  std::vector<int> wk;
  if (!shard_workers(n_shards, devices, cur, wk)) return fail(AHIP_E_ARG, "shard on a device that ahip_init_devices() did not select");
  std::vector<int32_t> rcs(n_shards, AHIP_OK);
  std::vector<std::string> errs(n_shards);
  std::vector<size_t> got(n_shards, 0);
  std::function<void(u32)> run_shard = [&](u32 s) {  // (on a worker: its own non-blocking stream; on the caller's thread: the default stream)
    rcs[s] = gzip_decode_impl(nullptr, (const u8 *)d_in[s], in_len[s], 0, 0, (u8 *)d_out[s], out_cap[s], false, nullptr, &got[s], g_ctx_stream);
    if (rcs[s] < 0) errs[s] = g_err;
  };
  run_shards(n_shards, wk, run_shard, lk);
  int32_t worst = AHIP_OK;
  for (u32 s = 0; s < n_shards; ++s) {
    out_len[s] = got[s];
    if (status) status[s] = rcs[s];
    if (rcs[s] < 0 && worst >= 0) { worst = rcs[s]; g_err = "shard " + std::to_string(s) + ": " + errs[s]; }
    else if (worst >= 0 && rcs[s] > worst) worst = rcs[s];
  }
  exchange_sizes(n_shards, devices, got.data(), offsets, cur);
  return worst;
}

// ONE long DEFLATE stream decoded by the device contexts of THIS process (the one-process form of ahip_stream_split_*: the same
// phases, shard s in the context of its device; what the ranks of a job all-gather is host memory here -- the block starts and the
// sizing results are concatenated on the calling thread, the 64 KiB window maps go through the host, 64 KiB x n each way).
// ref: zlib/inflate.dart:104-156, the block loop of one stream.  d_in[s]: the WHOLE compressed stream on shard s's device; the
// DEFLATE data starts at data_off.  d_out[s] / out_cap[s]: room for shard s's slice (slices are balanced on compressed bytes:
// allow for more than total / n); out_len[s], offsets[s] (offsets[n] = the total): what it wrote and where that lies in the
// stream's output; *end_pos: the reference's stream position behind the stream.  *handled = 0: not a case for the chunked
// decode (see ahip_stream_split_*): nothing was written, use ahip_inflate_raw / ahip_gzip_decode_device.
int32_t ahip_inflate_stream_shards(uint32_t n_shards, const int32_t *devices, const void *const *d_in, size_t in_len, size_t data_off,
                                   void *const *d_out, const size_t *out_cap, size_t *out_len, uint64_t *offsets, uint64_t *end_pos,
                                   int32_t *handled) {
  std::unique_lock<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  if (n_shards == 0 || n_shards > 4096 || !devices || !d_in || !d_out || !out_cap || !out_len || !offsets || !handled) return fail(AHIP_E_ARG, "NULL shard table");
  if (data_off > in_len) return fail(AHIP_E_ARG, "data_off behind the input");
  *handled = 0;
  if (end_pos) *end_pos = 0;
  for (u32 s = 0; s <= n_shards; ++s) { offsets[s] = 0; if (s < n_shards) out_len[s] = 0; }
  int cur = 0;
  HIP_TRY(hipGetDevice(&cur));
  std::vector<int> wk;
  if (!shard_workers(n_shards, devices, cur, wk)) return fail(AHIP_E_ARG, "shard on a device that ahip_init_devices() did not select");
  std::vector<std::unique_ptr<SplitState>> hs(n_shards);
  std::vector<int32_t> rcs(n_shards, AHIP_OK);
  std::vector<std::string> errs(n_shards);
  std::vector<u8> oks(n_shards, 0);
  // a phase on every shard, each in its own context (a state's buffers belong to the thread that runs its phases); true = all well
  auto phase = [&](const std::function<int32_t(u32, SplitState &)> &f) -> bool {
    std::function<void(u32)> run_shard = [&](u32 s) {
      if (rcs[s] < 0) return;
      if (!hs[s]) { hs[s].reset(new SplitState()); split_setup(*hs[s], d_in[s], in_len, data_off, s, n_shards, g_ctx_stream); }
      rcs[s] = f(s, *hs[s]);
      if (rcs[s] < 0) errs[s] = g_err;
    };
    run_shards(n_shards, wk, run_shard, lk);
    for (u32 s = 0; s < n_shards; ++s) if (rcs[s] < 0) return false;
    return true;
  };
  auto done = [&](int32_t ret) -> int32_t {  // the states' buffers go back to their own threads' pools
    std::function<void(u32)> drop = [&](u32 s) { if (hs[s]) hs[s]->release(); };
    run_shards(n_shards, wk, drop, lk);
    for (u32 s = 0; s < n_shards; ++s) if (rcs[s] < 0) { g_err = "shard " + std::to_string(s) + ": " + errs[s]; return rcs[s]; }
    return ret;
  };
  // 1. block starts
  if (!phase([&](u32, SplitState &h) -> int32_t {
        if (h.eligible) return split_candidates(&h);
        h.own.clear();
        if (h.rank == 0) h.own.push_back(h.off * 8);
        h.phase = 1;
        return AHIP_OK;
      })) return done(AHIP_OK);
  if (!hs[0]->eligible) return done(AHIP_OK);
  std::vector<u64> all;
  for (u32 s = 0; s < n_shards; ++s) all.insert(all.end(), hs[s]->own.begin(), hs[s]->own.end());
  // 2. sizes
  if (!phase([&](u32 s, SplitState &h) -> int32_t { bool ok = false; const int32_t r = split_size(&h, all.data(), all.size(), &ok); oks[s] = ok; return r; })) return done(AHIP_OK);
  for (u32 s = 0; s < n_shards; ++s) if (!oks[s]) return done(AHIP_OK);
  std::vector<u64> res;
  for (u32 s = 0; s < n_shards; ++s)
    for (const MemberResult &r : hs[s]->own_res) { res.push_back(r.status); res.push_back(r.out_len); res.push_back(r.end_pos); res.push_back(r.blocks); }
  // 3. the chain, symbols, window maps (through the host)
  std::vector<u16> maps((size_t)n_shards * SPLIT_MAP_ELEMS);
  if (!phase([&](u32 s, SplitState &h) -> int32_t {
        bool ok = false;
        int32_t r = split_chain(&h, res.data(), all.size(), &ok);
        oks[s] = ok;
        if (r != AHIP_OK || !ok) return r;
        if (h.dmap.reserve((size_t)SPLIT_MAP_ELEMS * 2) != hipSuccess) return fail(AHIP_E_DEVICE, "out of device memory");
        r = split_resolve(&h, h.dmap.as<u16>());
        if (r != AHIP_OK) return r;
        if (hipMemcpyAsync(maps.data() + (size_t)s * SPLIT_MAP_ELEMS, h.dmap.p, (size_t)SPLIT_MAP_ELEMS * 2, hipMemcpyDeviceToHost, h.st) != hipSuccess ||
            hipStreamSynchronize(h.st) != hipSuccess) return fail(AHIP_E_DEVICE, "window map read-back failed");
        return AHIP_OK;
      })) return done(AHIP_OK);
  for (u32 s = 0; s < n_shards; ++s) if (!oks[s]) return done(AHIP_OK);
  // 4. the bytes in front of every range, the slices
  std::vector<size_t> got(n_shards, 0);
  if (!phase([&](u32 s, SplitState &h) -> int32_t {
        if (h.dmaps.reserve(maps.size() * 2) != hipSuccess) return fail(AHIP_E_DEVICE, "out of device memory");
        if (hipMemcpyAsync(h.dmaps.p, maps.data(), maps.size() * 2, hipMemcpyHostToDevice, h.st) != hipSuccess) return fail(AHIP_E_DEVICE, "window map upload failed");
        bool ok = false;
        const int32_t r = split_finish(&h, h.dmaps.as<u16>(), (u8 *)d_out[s], out_cap[s], &got[s], &ok);
        oks[s] = ok;
        return r;
      })) {
    for (u32 s = 0; s < n_shards; ++s) out_len[s] = got[s];  // (AHIP_E_CAP: the sizes that are needed)
    return done(AHIP_OK);
  }
  for (u32 s = 0; s < n_shards; ++s) if (!oks[s]) return done(AHIP_OK);  // (some chunk differs from its sizing run: what was written is void)
  for (u32 s = 0; s < n_shards; ++s) { out_len[s] = got[s]; offsets[s] = hs[s]->base_out; }
  offsets[n_shards] = hs[0]->total_out;
  if (end_pos) *end_pos = hs[0]->end_pos;
  *handled = 1;
  g_last_shards = (int32_t)n_shards;
  return done(AHIP_OK);
}

int32_t ahip_device_count(void) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  return g_workers.empty() ? 1 : (int32_t)g_workers.size();
}

int32_t ahip_gzip_decode(const uint8_t *in, size_t in_len, int32_t verify, int32_t raw, uint8_t *out, size_t out_cap,
                         size_t *out_len) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  g_last_shards = 1;
  // (several devices: streams without size hints are partitioned too, after a sizing pass on this thread's device)
  // (the device workers may be a *_shards call's right now -- that call has let go of g_mu while its shards run, see
  //  run_shards -- and a Worker holds ONE job: whoever uses g_workers owns g_shards_mu.  Busy: this stream is decoded on
  //  the calling thread's device, same bytes.)
  if (!raw && !g_workers.empty()) {
    std::unique_lock<std::mutex> use(g_shards_mu, std::try_to_lock);
    if (use.owns_lock() && gzip_decode_sharded(g_workers, 1, in, in_len, verify, out, out_cap, out_len, &rc, true)) { g_last_shards = (int32_t)g_workers.size(); return rc; }
  }
  // one device: a large stream of BGZF members is cut into slices whose upload / decode / download overlap
  if (!raw && g_workers.empty() && in_len >= (32u << 20) && ensure_pipe() &&
      gzip_decode_sharded(g_pipe, 4, in, in_len, verify, out, out_cap, out_len, &rc, false)) { g_last_shards = (int32_t)g_pipe.size(); return rc; }
  return gzip_decode_host_impl(in, in_len, verify, raw, out, out_cap, out_len);
}

int32_t ahip_zlib_decode(const uint8_t *in, size_t in_len, int32_t verify, int32_t raw, uint8_t *out, size_t out_cap,
                         size_t *out_len) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  static thread_local DevBuf din, dout;
  HIP_TRY(din.reserve(in_len + 16));
  if (in_len) HIP_TRY(hipMemcpy(din.p, in, in_len, hipMemcpyHostToDevice));
  u64 committed = 0;
  rc = zlib_stream_device(in, din.as<u8>(), in_len, 0, true, verify, raw, dout, &committed, nullptr);
  if (out_len) *out_len = committed;
  if (rc == AHIP_OK || rc == AHIP_FALSE) {
    if (committed > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
    if (committed) HIP_TRY(hipMemcpy(out, dout.p, committed, hipMemcpyDeviceToHost));
  }
  return rc;
}

int32_t ahip_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len,
                         size_t *consumed) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  static thread_local DevBuf din, dout;
  HIP_TRY(din.reserve(in_len + 16));
  if (in_len) HIP_TRY(hipMemcpy(din.p, in, in_len, hipMemcpyHostToDevice));
  MemberResult r{};
  rc = inflate_one(din.as<u8>(), in_len, 0, nullptr, ~0ull, false, &r, nullptr);
  if (rc != AHIP_OK) return rc;
  if (out_len) *out_len = r.out_len;
  if (consumed) *consumed = r.end_pos;
  g_consumed = r.end_pos;
  int32_t st = member_status_to_abi(r.status);
  if (st < 0 || st == AHIP_RANGE || st == AHIP_HANG) return st;
  if (r.out_len > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
  if (r.out_len) {
    HIP_TRY(dout.reserve(r.out_len));
    MemberResult r2{};
    rc = inflate_one(din.as<u8>(), in_len, 0, dout.as<u8>(), r.out_len, true, &r2, nullptr);
    if (rc != AHIP_OK) return rc;
    if (r2.out_len != r.out_len) return fail(AHIP_E_DEVICE, "internal: decode disagrees with its own sizing run");
    HIP_TRY(hipMemcpy(out, dout.p, r.out_len, hipMemcpyDeviceToHost));
  }
  return st;
}

// Many independent raw DEFLATE streams in one call (the ZIP entry path).  Device-level worker.
static int32_t inflate_batch_impl(const u8 *d_in, size_t in_len, u32 n, const uint64_t *in_off, const uint64_t *in_size,
                                  const uint64_t *size_hint, u8 *d_out, size_t out_cap, uint64_t *out_off, uint64_t *out_len,
                                  int32_t *status, size_t *out_total, hipStream_t st, DevBuf *own_out) {
  if (out_total) *out_total = 0;
  if (n == 0) return AHIP_OK;
  if (!in_off || !in_size || !out_off || !out_len || !status) return fail(AHIP_E_ARG, "NULL entry table");
  for (u32 i = 0; i < n; ++i)
    if (in_off[i] > in_len || in_size[i] > in_len - in_off[i]) return fail(AHIP_E_ARG, "entry outside the input");
  static thread_local DevBuf ddesc, dres;
  std::vector<MemberDesc> md(n);
  std::vector<MemberResult> res(n);
  HIP_TRY(ddesc.reserve((size_t)n * sizeof(MemberDesc)));
  HIP_TRY(dres.reserve((size_t)n * sizeof(MemberResult)));
  // in_end = 0 means "unbounded", so an empty slice is expressed as a stream starting at the end of the input
  auto entry_desc = [&](u32 i, u64 ooff, u64 olim) {
    return in_size[i] ? MemberDesc{in_off[i], ooff, olim, POS_UNKNOWN, in_off[i] + in_size[i]}
                      : MemberDesc{(u64)in_len, ooff, olim, POS_UNKNOWN, 0};
  };
  // long entries are decoded by many waves each (sm_inflate), outside the batch launch
  auto big = [&](u32 i) { return in_size[i] >= sm_min_bytes() && !getenv("AHIP_NO_SM"); };
  auto skip_desc = [&](u64 ooff, u64 olim) { return MemberDesc{(u64)in_len, ooff, olim, POS_UNKNOWN, 0}; };
  std::vector<u64> size(n);
  if (size_hint) {
    // Directory sizes are untrusted: DEFLATE cannot expand a slice beyond 1032 x its length (258 bytes per 2 bits),
    // so anything larger is clamped -- the entry then reports AHIP_E_CAP like any other too-small window -- and the
    // sums below cannot wrap.
    for (u32 i = 0; i < n; ++i) {
      const u64 bound = in_size[i] > (~0ull >> 12) ? ~0ull >> 1 : in_size[i] * 1032 + 64;
      size[i] = size_hint[i] < bound ? size_hint[i] : bound;
    }
  } else {  // sizing run: every entry's true length
    for (u32 i = 0; i < n; ++i) md[i] = big(i) ? skip_desc(0, 0) : entry_desc(i, 0, ~0ull);
    HIP_TRY(hipMemcpyAsync(ddesc.p, md.data(), (size_t)n * sizeof(MemberDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(launch_inflate<false>(d_in, in_len, ddesc.as<MemberDesc>(), n, (u8 *)nullptr, dres.as<MemberResult>(), st));
    HIP_TRY(hipMemcpyAsync(res.data(), dres.p, (size_t)n * sizeof(MemberResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (u32 i = 0; i < n; ++i) {
      if (big(i)) { int32_t rc = inflate_one(d_in, in_off[i] + in_size[i], in_off[i], nullptr, ~0ull, false, &res[i], st); if (rc != AHIP_OK) return rc; }
      size[i] = res[i].out_len;
    }
  }
  std::vector<u64> off(n + 1);
  u64 total = 0;
  for (u32 i = 0; i < n; ++i) {
    off[i] = total;
    if (size[i] > (1ull << 46) || total + size[i] > (1ull << 46)) return fail(AHIP_E_ARG, "entry sizes add up to more than any device holds");
    total += size[i];
  }
  off[n] = total;
  if (out_total) *out_total = total;
  if (total > out_cap) return fail(AHIP_E_CAP, "output buffer too small");
  if (own_out) { HIP_TRY(own_out->reserve(total + 16)); d_out = own_out->as<u8>(); }
  for (u32 i = 0; i < n; ++i) md[i] = big(i) ? skip_desc(off[i], 0) : entry_desc(i, off[i], size[i]);
  HIP_TRY(hipMemcpyAsync(ddesc.p, md.data(), (size_t)n * sizeof(MemberDesc), hipMemcpyHostToDevice, st));
  HIP_TRY(launch_inflate<true>(d_in, in_len, ddesc.as<MemberDesc>(), n, d_out, dres.as<MemberResult>(), st, off.data()));
  HIP_TRY(hipMemcpyAsync(res.data(), dres.p, (size_t)n * sizeof(MemberResult), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  for (u32 i = 0; i < n; ++i)
    if (big(i)) {  // the slice end is this stream's end of input
      int32_t rc = inflate_one(d_in, in_off[i] + in_size[i], in_off[i], d_out + off[i], size[i], true, &res[i], st);
      if (rc != AHIP_OK) return rc;
    }
  for (u32 i = 0; i < n; ++i) {
    out_off[i] = off[i];
    out_len[i] = res[i].out_len;
    status[i] = res[i].status == MS_CAP ? AHIP_E_CAP : member_status_to_abi(res[i].status);
  }
  return AHIP_OK;
}

int32_t ahip_inflate_batch_device(const void *d_in, size_t in_len, uint32_t n_entries, const uint64_t *in_off,
                                  const uint64_t *in_size, const uint64_t *size_hint, void *d_out, size_t out_cap,
                                  uint64_t *out_off, uint64_t *out_len, int32_t *status, size_t *out_total, void *stream) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  return inflate_batch_impl((const u8 *)d_in, in_len, n_entries, in_off, in_size, size_hint, (u8 *)d_out, out_cap, out_off,
                            out_len, status, out_total, (hipStream_t)stream, nullptr);
}

int32_t ahip_inflate_batch(const uint8_t *in, size_t in_len, uint32_t n_entries, const uint64_t *in_off,
                           const uint64_t *in_size, const uint64_t *size_hint, uint8_t *out, size_t out_cap,
                           uint64_t *out_off, uint64_t *out_len, int32_t *status, size_t *out_total) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int32_t rc = ensure_init();
  if (rc != AHIP_OK) return rc;
  static thread_local DevBuf din, dout;
  HIP_TRY(din.reserve(in_len + 16));
  if (in_len) HIP_TRY(hipMemcpy(din.p, in, in_len, hipMemcpyHostToDevice));
  size_t total = 0;
  rc = inflate_batch_impl(din.as<u8>(), in_len, n_entries, in_off, in_size, size_hint, nullptr, out_cap, out_off, out_len,
                          status, &total, nullptr, &dout);
  if (out_total) *out_total = total;
  if (rc != AHIP_OK) return rc;
  if (total) HIP_TRY(hipMemcpy(out, dout.p, total, hipMemcpyDeviceToHost));
  return AHIP_OK;
}

}  // extern "C"
