// Micro benchmark (GPU box): what one DS instruction costs a CU, by kind and alignment, at the occupancy of the
// workgroup-per-member resolver (16 waves per CU, every wave issuing the same instruction back to back).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_ops tools/micro/lds_ops.hip && /tmp/lds_ops
// Reported: shader cycles per wave-instruction per CU (16 waves x ITER instructions / elapsed cycles of the slowest wave).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
constexpr int ITER = 2000;
// MODE: 0 read_b64  1 write_b64  2 read_b32  3 write_b32  4 write_b16  5 write_b8  6 mskor_b64  7 read_b128(aligned only)  8 read_u8
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t *offs, uint64_t *cyc, uint64_t *sink, int active) {
  __shared__ uint8_t buf[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += blockDim.x) ((uint32_t *)buf)[i] = i;
  __syncthreads();
  uint32_t a = offs[tid];  // byte address inside a 4 KiB page of the wave
  const uint32_t base = (tid >> 6) * 4096;
  uint64_t acc = 0;
  const bool on = lane < active;
  const uint32_t ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)buf + base + a;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  if (on) {
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
      uint64_t v = acc + it;
      uint32_t w = (uint32_t)v;
      if (MODE == 0) { uint64_t r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(ad)); acc ^= r; }
      if (MODE == 1) asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(v));
      if (MODE == 2) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(ad)); acc ^= r; }
      if (MODE == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(ad), "v"(w));
      if (MODE == 4) asm volatile("ds_write_b16 %0, %1" ::"v"(ad), "v"(w));
      if (MODE == 5) asm volatile("ds_write_b8 %0, %1" ::"v"(ad), "v"(w));
      if (MODE == 6) asm volatile("ds_mskor_b64 %0, %1, %2" ::"v"(ad), "v"(0xff00ffull << (8 * (lane & 3))), "v"(v));
      if (MODE == 7) { uint4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(ad)); acc ^= r.x; }
      if (MODE == 8) { uint32_t r; asm volatile("ds_read_u8 %0, %1" : "=v"(r) : "v"(ad)); acc ^= r; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[tid >> 6] = t1 - t0;
  sink[tid] = acc;
}
int main() {
  uint32_t *d_off; uint64_t *d_cyc, *d_sink;
  hipMalloc(&d_off, 4096); hipMalloc(&d_cyc, 8 * 16); hipMalloc(&d_sink, 8 * 1024);
  const char *names[] = {"read_b64", "write_b64", "read_b32", "write_b32", "write_b16", "write_b8", "mskor_b64", "read_b128", "read_u8"};
  // address patterns: lane l at stride 24 (bank-friendly) + align; or "token-like": consecutive lanes ~4 bytes apart (+ align jitter)
  for (int pat = 0; pat < 4; ++pat) {
    for (int active : {64, 16}) {
      printf("pattern %d (%s), %d lanes active:\n", pat, pat == 0 ? "stride 24 B, aligned 8" : pat == 1 ? "stride 24 B + 3 (unaligned)" : pat == 2 ? "tokens: 5 B apart (mixed alignment)" : "tokens: 8 B apart, aligned 8", active);
      std::vector<uint32_t> offs(1024);
      for (int t = 0; t < 1024; ++t) { int l = t & 63; offs[t] = pat == 0 ? l * 24 : pat == 1 ? l * 24 + 3 : pat == 2 ? l * 5 : l * 8; }
      hipMemcpy(d_off, offs.data(), 4096, hipMemcpyHostToDevice);
      for (int mode = 0; mode < 9; ++mode) {
        if (mode == 7 && pat != 0) continue;
        if (mode == 6 && (pat == 1 || pat == 2)) { for (auto &o : offs) o &= ~7u; hipMemcpy(d_off, offs.data(), 4096, hipMemcpyHostToDevice); }
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 7: hipLaunchKernelGGL(k<7>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
          case 8: hipLaunchKernelGGL(k<8>, dim3(1), dim3(1024), 0, 0, d_off, d_cyc, d_sink, active); break;
        }
        uint64_t c[16];
        hipMemcpy(c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
        uint64_t mx = 0; for (auto v : c) mx = v > mx ? v : mx;
        printf("   %-10s %6.1f cycles per wave-instruction per CU\n", names[mode], (double)mx / (16.0 * ITER));
        if (mode == 6 && (pat == 1 || pat == 2)) { for (int t = 0; t < 1024; ++t) { int l = t & 63; offs[t] = pat == 1 ? l * 24 + 3 : l * 5; } hipMemcpy(d_off, offs.data(), 4096, hipMemcpyHostToDevice); }
      }
    }
  }
  return 0;
}
