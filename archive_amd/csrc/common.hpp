// common.hpp -- shared device/host types for libarchive_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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
  MS_FARREF = 17,   // back-reference reaches before this member's first byte
  MS_OVERSUB = 18,  // over-subscribed Huffman code lengths (not reproduced)
};

struct MemberDesc {
  u64 in_off;     // byte offset of the first DEFLATE byte
  u64 out_off;    // byte offset of this member's output window in the output buffer
  u64 out_limit;  // size of that window
  u64 expect_end; // expected reference stream position after the deflate data (~0 = unknown)
};

struct MemberResult {
  u64 end_pos;  // reference InputStream position after Inflate returned
  u64 out_len;  // bytes produced
  u32 status;   // MS_*
  u32 blocks;   // DEFLATE blocks parsed
};

#define AHIP_DEVINL __device__ __forceinline__

// Compiler-level ordering point for cross-lane traffic through LDS/global inside ONE wave.
// The hardware already executes a wave's DS (and vector-memory) instructions in issue order;
// this only stops hipcc from moving accesses across it.
AHIP_DEVINL void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

AHIP_DEVINL u32 uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
AHIP_DEVINL u64 uniform64(u64 v) {
  u32 lo = uniform((u32)v), hi = uniform((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}

struct __attribute__((packed, aligned(1))) unaligned_u64 { u64 v; };
struct __attribute__((packed, aligned(1))) unaligned_u32 { u32 v; };
AHIP_DEVINL u64 load_u64_unaligned(const u8 *p) { return ((const unaligned_u64 *)p)->v; }
AHIP_DEVINL u32 load_u32_unaligned(const u8 *p) { return ((const unaligned_u32 *)p)->v; }

}  // namespace ahip
