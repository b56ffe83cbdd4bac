// Micro benchmark (GPU box), second part of far_fetch.hip: the same 4.15e8 random 16-byte fetches out of 160 MB of per-wave
// history (which alone stay in the 256 MB memory-side cache: 3.5 ms), now NEXT TO the streams the resolver also moves per
// decode -- 4.5 GB of tokens read once, 4.3 GB of output written once -- with ordinary and with non-temporal loads / stores
// for the streams.  Does the history stay cached?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/far_fetch2 tools/micro/far_fetch2.hip && /tmp/far_fetch2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
struct __attribute__((packed, aligned(1))) U128 { uint4 v; };
template <int NT, int RINGW>
__global__ __launch_bounds__(64) void k(uint8_t *hist, uint32_t hist_bytes, int iters, uint32_t *sink, const uint4 *sin, uint4 *sout,
                                        size_t per_wave16) {
  const uint32_t wave = blockIdx.x, lane = threadIdx.x;
  uint8_t *mine = hist + (size_t)wave * hist_bytes;
  const uint4 *ra = sin + (size_t)wave * per_wave16;
  uint4 *wa = sout + (size_t)wave * per_wave16;
  uint32_t x = wave * 2654435761u + lane * 40503u + 1u, acc = 0;
  size_t sp = lane;
#pragma unroll 2
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    const uint32_t off = (x >> 8) % (hist_bytes - 32);
    const uint4 v = ((const U128 *)(mine + off))->v;
    acc ^= v.x ^ v.w;
    if (lane < 44) {  // ~ 10.8 bytes of tokens read and 10.3 bytes of output written per fetch
      uint4 t;
      if (NT) { t.x = __builtin_nontemporal_load(&ra[sp].x); t.y = __builtin_nontemporal_load(&ra[sp].y); t.z = __builtin_nontemporal_load(&ra[sp].z); t.w = __builtin_nontemporal_load(&ra[sp].w); }
      else t = ra[sp];
      acc ^= t.y;
      if (lane < 41) {
        if (NT) { __builtin_nontemporal_store(v.x, &wa[sp].x); __builtin_nontemporal_store(v.y, &wa[sp].y); __builtin_nontemporal_store(v.z, &wa[sp].z); __builtin_nontemporal_store(v.w, &wa[sp].w); }
        else wa[sp] = v;
      }
      if (RINGW && (it & 1)) ((uint4 *)mine)[((size_t)it * 32 + lane) % (hist_bytes / 16)] = v;  // the history itself being written (ordinary stores)
      sp += 64;
    }
  }
  sink[wave * 64 + lane] = acc;
}
int main() {
  const uint32_t waves = 5120, hist = 32768;
  const int iters = 1266;
  const size_t per_wave16 = (size_t)iters * 64 + 64;  // 16-byte units of stream per wave
  uint32_t *d_sink; hipMalloc(&d_sink, waves * 64 * 4);
  uint8_t *buf; uint4 *sin, *sout;
  hipMalloc(&buf, (size_t)waves * hist);
  hipMalloc(&sin, (size_t)waves * per_wave16 * 16);
  hipMalloc(&sout, (size_t)waves * per_wave16 * 16);
  hipMemset(buf, 1, (size_t)waves * hist);
  hipMemset(sin, 2, (size_t)waves * per_wave16 * 16);
  printf("streams: %.2f GB read, %.2f GB written per run\n", waves * 44.0 * iters * 16 / 1e9, waves * 41.0 * iters * 16 / 1e9);
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL((k<0, 0>), dim3(waves), dim3(64), 0, 0, buf, hist, iters, d_sink, sin, sout, per_wave16);
      if (mode == 1) hipLaunchKernelGGL((k<1, 0>), dim3(waves), dim3(64), 0, 0, buf, hist, iters, d_sink, sin, sout, per_wave16);
      if (mode == 2) hipLaunchKernelGGL((k<0, 1>), dim3(waves), dim3(64), 0, 0, buf, hist, iters, d_sink, sin, sout, per_wave16);
      if (mode == 3) hipLaunchKernelGGL((k<1, 1>), dim3(waves), dim3(64), 0, 0, buf, hist, iters, d_sink, sin, sout, per_wave16);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms = 0; hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
    }
    printf("%s streams%s: %.2f ms for 4.15e8 fetches\n", (mode & 1) ? "non-temporal" : "ordinary", (mode & 2) ? ", history rewritten as it goes" : "", best);
  }
  return 0;
}
