// common.hpp -- shared device/host types for libarchive_hip (gfx950 only).
#pragma once
#ifdef AHIP_HOST_EMU  // CPU emulation of one wave for tests (tests/emu/wave_emu.hpp); never defined in the product build
#include "../../tests/emu/wave_emu.hpp"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace ahip {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

constexpr int kWave = 64;  // CDNA4 wavefront

// Per-member verdicts written by the inflate kernels.  0..3 mirror the C-ABI status codes.
enum : u32 {
  MS_OK = 0,        // final block decoded
  MS_FALSE = 1,     // reference `return -1` (bad symbol, bad stored header, EOS inside a block, BTYPE 3)
  MS_RANGE = 2,     // reference would throw RangeError
  MS_HANG = 3,      // reference would not terminate (zero-length litlen entry)
  MS_EOS = 4,       // input ended between blocks before a final block: the reference just stops
  MS_FALSE_EOS = 5, // internal: MS_FALSE caused by running out of input (stream position = end)
  MS_CAP = 16,      // member wanted to write past its output window
  MS_FARREF = 17,   // back-reference reaches before the first byte of the OUTPUT STREAM (Dart RangeError)
  MS_OVERSUB = 18,  // internal: over-subscribed Huffman code lengths met by the fast kernels; the late kernel decodes
                    // the member with the reference's own overwritten single-level table
  MS_CHUNK_END = 19, // chunked single-stream decode: stopped at the next chunk's first block (end_pos is a BIT position)
  MS_TOKFULL = 20,  // internal: the member's token area / run directory overflowed; the byte-writing serial kernel redoes it
  MS_INTERNAL = 21, // a kernel's own invariant did not hold (a loop bound that "cannot" be reached was reached): never silent --
                    // the host reports AHIP_E_DEVICE
};

struct MemberDesc {
  u64 in_off;     // byte offset of the first DEFLATE byte
  u64 out_off;    // byte offset of this member's output window in the output buffer
  u64 out_limit;  // size of that window
  u64 expect_end; // expected reference stream position after the deflate data (~0 = unknown)
  u64 in_end;     // end of the bytes this stream may read (0 = the end of the whole input): a ZIP entry's slice
  u32 hist;       // bytes of earlier output (in front of out_off) a back-reference may reach: gzip members share one
                  // OutputStream in the reference (quirk q8); 0 for streams with an output of their own
  u32 pad;
};
constexpr u32 MR_FAR = 0x80000000u;  // MemberResult::blocks: some back-reference reaches into earlier output (resolved late, in order)
constexpr u32 MR_REACH = 0x40000000u;  // MemberResult::blocks of a watched CHUNK (the first one of a long member): some back-reference
                                      // reaches in front of the chunk's own output, i.e. into what earlier gzip members produced (q8)

// Chunked decode of ONE long stream (sm_inflate): a chunk starts at a block header found by the block finder
// (any bit offset), may refer to `hist` bytes of output before its own, and stops in front of the first block
// that starts at one of the candidate positions.
struct ChunkDesc {
  u64 start_bit;  // absolute bit position of the chunk's first block header
  u64 out_off;    // token-stream / output offset (in elements) of the chunk
  u64 out_limit;  // output window size
  u32 hist;       // bytes of earlier output a back-reference may reach into (<= 32768)
  u32 pad;
};
struct ChunkCtx {
  const u64 *cand_bits;  // sorted candidate block-start bit positions
  u32 n_cand;
  u32 start_bit;         // 0..7 within the first byte
  u32 hist;
  u32 watch;             // report (MR_REACH) whether a back-reference reaches in front of the chunk's own output
};

struct MemberResult {
  u64 end_pos;  // reference InputStream position after Inflate returned
  u64 out_len;  // bytes produced
  u32 status;   // MS_*
  u32 blocks;   // DEFLATE blocks parsed
  u32 windows;  // parallel-decode windows committed        (diagnostics)
  u32 rounds;   // pass-B rounds over all windows           (diagnostics)
  u32 fallbacks;// windows handed to the serial decoder     (diagnostics)
  u32 partial;  // windows cut short by the token/byte caps (diagnostics)
  u32 cyc[8];   // -DAHIP_PROFILE builds: shader-clock cycles / 16 per phase
                //  0 header+tables 1 stage 2 pass A 3 pass B 4 emit 5 resolve 6 flush 7 serial decode
  u64 tok_words; // runs the tokenizer entered into this member's directory
};

#define AHIP_DEVINL __device__ __forceinline__

// A place where the code relies on a wave executing its LDS instructions in lock step (all lanes' earlier store before
// any lane's later one) without needing a compiler fence.  Nothing on the device; a barrier in the CPU emulation.
#ifdef AHIP_HOST_EMU
#define AHIP_LOCKSTEP() wave_emu::barrier()
#else
#define AHIP_LOCKSTEP() do { } while (0)
#endif
// A wave vote under a PER-LANE condition that only decides whether a branch can be skipped by everybody (the branch
// body tests the lane's own predicate again): the active lanes' vote on the device, the lane's own predicate in the
// CPU emulation (where lanes that are not in the branch cannot take part in an exchange).
// AHIP_ASM_NOTE: a comment in the ISA that also keeps the compiler from if-converting the branch around it.
#ifdef AHIP_HOST_EMU
#define AHIP_ANY_HINT(p) (p)
#define AHIP_ASM_NOTE(text) do { } while (0)
#else
#define AHIP_ANY_HINT(p) __any(p)
#define AHIP_ASM_NOTE(text) asm volatile("; " text ::: "memory")
#endif

// The value has to be in its register here: the compiler's wait for the load that makes it is placed at this point
// (inside a rarely taken branch) instead of at the next use on the common path.  Nothing in the CPU emulation.
#ifdef AHIP_HOST_EMU
#define AHIP_PIN(x) do { } while (0)
#else
#define AHIP_PIN(x) asm volatile("" : "+v"(x))
#endif

#ifdef AHIP_PROFILE
#define AHIP_TICK(var) const u64 var = __builtin_amdgcn_s_memtime()
#define AHIP_ACC(slot, t0, t1) (slot) += (u32)(((t1) - (t0)) >> 4)
#else
#define AHIP_TICK(var) do { } while (0)
#define AHIP_ACC(slot, t0, t1) do { } while (0)
#endif

// A word in LDS that one wave of a workgroup writes and the others poll (the frontier of the workgroup-per-member
// resolver).  The CU's LDS executes the DS instructions of all its waves in one order and a wave's own in issue order, so
// "data, then the word" by the writer and "the word, then data" by a reader need no hardware wait -- only the compiler must
// keep the order (wave_sync() on both sides).  In the CPU emulation lanes are threads: acquire / release.
#ifdef AHIP_HOST_EMU
#define AHIP_LDS_POLL(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define AHIP_LDS_POST(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#else
#define AHIP_LDS_POLL(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define AHIP_LDS_POST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif

// Compiler-level ordering point for cross-lane traffic through LDS/global inside ONE wave.
// The hardware already executes a wave's DS (and vector-memory) instructions in issue order;
// this only stops hipcc from moving accesses across it.
AHIP_DEVINL void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// number of set bits of a wave mask below this lane (v_mbcnt: no per-lane 64-bit mask to keep in registers)
#ifdef AHIP_HOST_EMU
AHIP_DEVINL u32 wave_rank(u64 mask) { return (u32)__builtin_popcountll(mask & ((1ull << wave_emu::lane) - 1)); }
#else
AHIP_DEVINL u32 wave_rank(u64 mask) { return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u)); }
#endif

AHIP_DEVINL u32 uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
AHIP_DEVINL u64 uniform64(u64 v) {
  u32 lo = uniform((u32)v), hi = uniform((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// ---- cross-lane primitives on the DPP / readlane paths (no LDS crossbar round trip) ----
// gfx9-family DPP controls: row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK>
AHIP_DEVINL u32 dpp_zero(u32 v) {  // lanes without a source (or masked rows) read 0
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
AHIP_DEVINL u32 lane_bcast(u32 v, int src_lane /* wave-uniform */) {
  return (u32)__builtin_amdgcn_readlane((int)v, src_lane);
}
AHIP_DEVINL unsigned long long lane_bcast64(unsigned long long v, int src_lane /* wave-uniform */) {
  return (unsigned long long)lane_bcast((u32)v, src_lane) | ((unsigned long long)lane_bcast((u32)(v >> 32), src_lane) << 32);
}
// value of lane-1 (0 for lane 0)
AHIP_DEVINL u32 lane_prev(u32 v) { return dpp_zero<0x138, 0xf>(v); }
// inclusive prefix sum over the wave
AHIP_DEVINL u32 wave_incl_sum(u32 v) {
  v += dpp_zero<0x111, 0xf>(v);
  v += dpp_zero<0x112, 0xf>(v);
  v += dpp_zero<0x114, 0xf>(v);
  v += dpp_zero<0x118, 0xf>(v);
  v += dpp_zero<0x142, 0xa>(v);
  v += dpp_zero<0x143, 0xc>(v);
  return v;
}
// exclusive prefix sum; total = wave sum (uniform)
AHIP_DEVINL u32 wave_excl_sum(u32 v, u32 &total) {
  u32 inc = wave_incl_sum(v);
  total = lane_bcast(inc, 63);
  return inc - v;
}
// inclusive prefix maximum over the wave (0 is the identity)
AHIP_DEVINL u32 wave_incl_umax(u32 v) {
  u32 t;
  t = dpp_zero<0x111, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x112, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x114, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x118, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x142, 0xa>(v); v = t > v ? t : v;
  t = dpp_zero<0x143, 0xc>(v); v = t > v ? t : v;
  return v;
}
// value held by lane `src` (per-lane index): LDS crossbar
AHIP_DEVINL u32 lane_gather(u32 v, u32 src) { return (u32)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }
// wave maximum (uniform result)
AHIP_DEVINL u32 wave_umax(u32 v) {
  u32 t;
  t = dpp_zero<0x111, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x112, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x114, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x118, 0xf>(v); v = t > v ? t : v;
  t = dpp_zero<0x142, 0xa>(v); v = t > v ? t : v;
  t = dpp_zero<0x143, 0xc>(v); v = t > v ? t : v;
  return lane_bcast(v, 63);
}

struct __attribute__((packed, aligned(1))) unaligned_u64 { u64 v; };
struct __attribute__((packed, aligned(1))) unaligned_u32 { u32 v; };
struct __attribute__((packed, aligned(1))) unaligned_u16 { u16 v; };
struct __attribute__((packed, aligned(1))) unaligned_u128 { uint4 v; };
AHIP_DEVINL uint4 load_u128_unaligned(const u8 *p) { return ((const unaligned_u128 *)p)->v; }
AHIP_DEVINL u64 load_u64_unaligned(const u8 *p) { return ((const unaligned_u64 *)p)->v; }
AHIP_DEVINL u32 load_u32_unaligned(const u8 *p) { return ((const unaligned_u32 *)p)->v; }

}  // namespace ahip
