// Micro test (GPU box): do unaligned ds_write_b64/b32 and ds_read_b64/b32 work on gfx950, and what do they cost?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned tools/micro/lds_unaligned.hip && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
struct __attribute__((packed, aligned(1))) U32 { uint32_t v; };
template <int MODE>  // 0 aligned b64, 1 unaligned b64 (offset +k), 2 byte stores
__global__ void k(const uint32_t *offs, uint8_t *out, uint64_t *cyc, int iters) {
  __shared__ uint8_t buf[8192 + 64];
  const int l = threadIdx.x;
  for (int i = l; i < 8192 + 64; i += 64) buf[i] = 0;
  __syncthreads();
  const uint32_t off = offs[l];
  const uint64_t val = 0x0807060504030201ull + 0x1010101010101010ull * (uint64_t)l;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t o = (off + it * 8) & 8191;
    if (MODE == 2) { for (int b = 0; b < 8; ++b) buf[o + b] = (uint8_t)(val >> (8 * b)); }
    else ((U64 *)(buf + o))->v = val + it;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    acc += ((U64 *)(buf + ((o + 3) & 8191)))->v;
    acc += ((U32 *)(buf + ((o + 1) & 8191)))->v;
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (l == 0) cyc[0] = t1 - t0;
  cyc[1 + l] = acc;
  for (int i = l; i < 8192; i += 64) out[i] = buf[i];
}
int main() {
  uint32_t *d_off; uint8_t *d_out; uint64_t *d_cyc;
  hipMalloc(&d_off, 256); hipMalloc(&d_out, 8192); hipMalloc(&d_cyc, 8 * 65);
  for (int align = 0; align < 8; ++align) {
    uint32_t offs[64];
    for (int l = 0; l < 64; ++l) offs[l] = l * 24 + align;  // disjoint 8-byte targets, every alignment
    hipMemcpy(d_off, offs, 256, hipMemcpyHostToDevice);
    // correctness: one iteration, compare with a host model
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d_off, d_out, d_cyc, 1);
    std::vector<uint8_t> got(8192), want(8192, 0);
    hipMemcpy(got.data(), d_out, 8192, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { uint64_t v = 0x0807060504030201ull + 0x1010101010101010ull * (uint64_t)l; memcpy(&want[offs[l]], &v, 8); }
    const bool ok = got == want;
    uint64_t c[3];
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0) { uint32_t o2[64]; for (int l = 0; l < 64; ++l) o2[l] = l * 24; hipMemcpy(d_off, o2, 256, hipMemcpyHostToDevice); }
      else hipMemcpy(d_off, offs, 256, hipMemcpyHostToDevice);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d_off, d_out, d_cyc, 1000);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d_off, d_out, d_cyc, 1000);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d_off, d_out, d_cyc, 1000);
      hipMemcpy(&c[mode], d_cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("align %d: unaligned b64 store %s; cycles/iter aligned %.1f unaligned %.1f bytewise %.1f\n", align, ok ? "CORRECT" : "WRONG",
           c[0] / 1000.0, c[1] / 1000.0, c[2] / 1000.0);
  }
  return 0;
}
