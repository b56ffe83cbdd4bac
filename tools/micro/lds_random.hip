// Micro benchmark (GPU box): what the tokenizer's table look-ups cost the LDS and why two thirds of its LDS cycles are
// bank-conflict cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 66 %, profiles/r0*_sq_counters.md).  64 lanes read a dword
// each at an index that is (a) random in a 512-entry table, the way the next nine bits of a compressed stream index the
// literal/length table, (b) the same but in a table padded to 33 dwords per 32, (c) consecutive, (d) random in a 128-entry
// table (the distance table).  12 waves per CU like the tokenizer.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_random tools/micro/lds_random.hip && /tmp/lds_random
//   (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS over it: conflict share per pattern)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr int ITER = 4000;
template <int MODE>
__global__ __launch_bounds__(768) void k(uint64_t *cyc, uint32_t *sink) {
  __shared__ uint32_t tab[12][640];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = lane; i < 640; i += 64) tab[wave][i] = i * 2654435761u;
  __syncthreads();
  uint32_t x = tid * 2654435761u + 12345u, acc = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int it = 0; it < ITER; ++it) {
    x = x * 1664525u + 1013904223u;
    uint32_t idx;
    if (MODE == 0) idx = (x >> 10) & 511;
    if (MODE == 1) { idx = (x >> 10) & 511; idx += idx >> 5; }  // padded: 33 dwords per 32
    if (MODE == 2) idx = (lane + it) & 511;
    if (MODE == 3) idx = (x >> 10) & 127;
    acc += tab[wave][idx];
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[wave] = t1 - t0;
  sink[tid] = acc;
}
int main() {
  uint64_t *d_cyc; uint32_t *d_sink;
  hipMalloc(&d_cyc, 8 * 12); hipMalloc(&d_sink, 4 * 768);
  const char *names[] = {"random in 512 entries", "random in 512 entries, table padded 33/32", "consecutive", "random in 128 entries"};
  for (int mode = 0; mode < 4; ++mode) {
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(768), 0, 0, d_cyc, d_sink);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(768), 0, 0, d_cyc, d_sink);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(768), 0, 0, d_cyc, d_sink);
    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(768), 0, 0, d_cyc, d_sink);
    uint64_t c[12]; hipMemcpy(c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
    uint64_t mx = 0; for (auto v : c) mx = v > mx ? v : mx;
    printf("%-44s %5.2f cycles per wave-instruction per CU\n", names[mode], (double)mx / (12.0 * ITER));
  }
  return 0;
}
