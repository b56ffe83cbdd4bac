// Micro test (GPU box): two adjacent unaligned 8-byte LDS reads are fused by hipcc into ONE ds_read_b128 at an
// arbitrary byte address (the token-centric resolver variant, -DAHIP_TOKEN_RESOLVER, relies on it).  Does gfx950 return
// the right 16 bytes at every alignment?  Round 2, MI355X: CORRECT at all 16 alignments (64 lanes, stride 48).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned128 tools/micro/lds_unaligned128.hip && /tmp/lds_unaligned128
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
__global__ void k(const uint32_t *offs, uint64_t *out) {
  __shared__ uint8_t buf[4096 + 64];
  const int l = threadIdx.x;
  for (int i = l; i < 4096 + 64; i += 64) buf[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  const uint32_t o = offs[l];
  const uint64_t w0 = ((const U64 *)&buf[o])->v, w1 = ((const U64 *)&buf[o + 8])->v;
  out[2 * l] = w0;
  out[2 * l + 1] = w1;
}
int main() {
  uint32_t *d_off; uint64_t *d_out;
  hipMalloc(&d_off, 256); hipMalloc(&d_out, 1024);
  int bad = 0;
  for (int align = 0; align < 16; ++align) {
    uint32_t offs[64];
    for (int l = 0; l < 64; ++l) offs[l] = l * 48 + align;
    hipMemcpy(d_off, offs, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_off, d_out);
    uint64_t got[128];
    hipMemcpy(got, d_out, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
      for (int h = 0; h < 2; ++h) {
        uint64_t want = 0;
        for (int b = 0; b < 8; ++b) want |= (uint64_t)(uint8_t)((offs[l] + 8 * h + b) * 7 + 3) << (8 * b);
        if (got[2 * l + h] != want) ok = 0;
      }
    printf("align %2d: 16-byte unaligned LDS read %s\n", align, ok ? "CORRECT" : "WRONG");
    bad += !ok;
  }
  return bad != 0;
}
