// Micro benchmark (GPU box): the resolver's far fetches -- every wave reads 16 bytes per lane at random places of ITS OWN
// 32 KiB of history, 5 120 waves live (20 per CU) -- from ordinary device memory and from memory allocated uncached
// (hipDeviceMallocUncached: no L2 line fill, the fabric moves what was asked for).  How many fetches per second, and does
// the request size change?  (rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ... over this binary gives the sizes.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/far_fetch tools/micro/far_fetch.hip && /tmp/far_fetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
struct __attribute__((packed, aligned(1))) U128 { uint4 v; };
__global__ __launch_bounds__(64) void k(const uint8_t *hist, uint32_t hist_bytes, int iters, uint32_t *sink, int writes) {
  const uint32_t wave = blockIdx.x, lane = threadIdx.x;
  const uint8_t *mine = hist + (size_t)wave * hist_bytes;
  uint32_t x = wave * 2654435761u + lane * 40503u + 1u, acc = 0;
#pragma unroll 4
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    const uint32_t off = (x >> 8) % (hist_bytes - 32);
    const uint4 v = ((const U128 *)(mine + off))->v;
    acc ^= v.x ^ v.w;
    if (writes && (it & 15) == 0) ((uint4 *)(const_cast<uint8_t *>(mine)))[(it >> 4) % (hist_bytes / 16)] = v;  // (a window going out now and then)
  }
  sink[wave * 64 + lane] = acc;
}
int main(int argc, char **argv) {
  // hist: bytes of history per wave -- 32 KiB (160 MB in all: fits the 256 MB memory-side cache) or, to see what HBM itself
  // does with such reads, 512 KiB (2.6 GB in all)
  const uint32_t waves = 5120, hist = argc > 1 ? (uint32_t)atoi(argv[1]) * 1024u : 32768u;
  const int iters = 1266;  // 5120 x 64 x 1266 = 4.15e8 fetches: one decode's worth
  uint32_t *d_sink; hipMalloc(&d_sink, waves * 64 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    uint8_t *buf = nullptr;
    hipError_t e = mode == 0 ? hipMalloc(&buf, (size_t)waves * hist) : hipExtMallocWithFlags((void **)&buf, (size_t)waves * hist, hipDeviceMallocUncached);
    if (e != hipSuccess) { printf("mode %d: allocation failed: %s\n", mode, hipGetErrorString(e)); continue; }
    hipMemset(buf, 1, (size_t)waves * hist);
    for (int writes = 0; writes < 2; ++writes) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, buf, hist, 50, d_sink, writes);
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, buf, hist, iters, d_sink, writes);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms = 0; hipEventElapsedTime(&ms, a, b);
      printf("%s memory%s: %.2f ms for %.3g fetches of 16 bytes = %.1f G fetches/s (as 128-byte lines: %.2f TB/s)\n", mode ? "uncached" : "ordinary",
             writes ? " (+ stores)" : "", ms, (double)waves * 64 * iters, (double)waves * 64 * iters / ms / 1e6, (double)waves * 64 * iters * 128 / ms / 1e9);
    }
    hipFree(buf);
  }
  return 0;
}
