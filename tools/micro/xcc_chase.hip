// Dev tool: (1) which XCD does workgroup i run on?  (s_getreg XCC_ID against blockIdx.x mod 8)
//           (2) dependent random 4-byte reads over a vector of `mb` MB walked by the workgroups of ONE XCD each
//               (the bzip2 inverse-BWT walk's access pattern): reads per second against the vector's size.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcc_chase tools/micro/xcc_chase.hip && /tmp/xcc_chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
typedef unsigned int u32;
__global__ void xcc_ids(u32 *out) {
  u32 id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) out[blockIdx.x] = id;
}
// vec[x * n + i] = next index (a random cycle per XCD x); every thread walks `steps` steps from its own start
__global__ __launch_bounds__(256) void chase(const u32 *__restrict__ vec, u32 n, u32 steps, u32 wpx, u32 *sink, u32 same) {
  const u32 xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const u32 *v = vec + (size_t)(same ? 0 : xcd) * n;
  u32 cur = (u32)(((unsigned long long)(j * 256 + threadIdx.x) * 2654435761ull) % n);
  for (u32 s = 0; s < steps; ++s) cur = v[cur];
  if (cur == 0xffffffffu) sink[0] = cur;
}
int main() {
  u32 *d_ids; hipMalloc(&d_ids, 4096 * 4);
  hipLaunchKernelGGL(xcc_ids, dim3(4096), dim3(64), 0, 0, d_ids);
  std::vector<u32> ids(4096); hipMemcpy(ids.data(), d_ids, 4096 * 4, hipMemcpyDeviceToHost);
  u32 match = 0; for (u32 i = 0; i < 4096; ++i) match += ((ids[i] & 15) == (i & 7));
  printf("xcc id == blockIdx mod 8 for %u of 4096 workgroups; first 16:", match);
  for (u32 i = 0; i < 16; ++i) printf(" %u", ids[i] & 15);
  printf("\n");
  std::mt19937 rng(1);
  for (double mb : {1.0, 2.0, 3.0, 3.6, 4.0, 6.0, 16.0, 64.0}) {
    const u32 n = (u32)(mb * 1e6 / 4);
    std::vector<u32> h((size_t)n * 8);
    for (u32 x = 0; x < 8; ++x) {
      std::vector<u32> p(n); std::iota(p.begin(), p.end(), 0u); std::shuffle(p.begin(), p.end(), rng);
      for (u32 i = 0; i < n; ++i) h[(size_t)x * n + p[i]] = p[(i + 1) % n];
    }
    u32 *d, *sink; hipMalloc(&d, h.size() * 4); hipMalloc(&sink, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (u32 wpx : {28u, 56u, 112u}) {
      for (u32 same : {0u, 1u}) {
        const u32 steps = 512;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(chase, dim3(8 * wpx), dim3(256), 0, 0, d, n, steps, wpx, sink, same);
        hipEventRecord(e0);
        hipLaunchKernelGGL(chase, dim3(8 * wpx), dim3(256), 0, 0, d, n, steps, wpx, sink, same);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double reads = 8.0 * wpx * 256 * steps;
        printf("vector %5.1f MB per XCD%s, %3u workgroups per XCD: %7.3f ms  %6.1f G reads/s  (%5.0f ns per dependent read)\n", mb,
               same ? " (ALL XCDs on ONE vector)" : "", wpx, ms, reads / ms / 1e6, ms * 1e6 / steps);
      }
    }
    hipFree(d); hipFree(sink);
  }
  return 0;
}
