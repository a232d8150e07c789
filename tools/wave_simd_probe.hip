// Which SIMD does wave w of a 768-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8] on gfx9)
//   hipcc --offload-arch=gfx950 -O2 -o tools/wave_simd_probe tools/wave_simd_probe.hip && ./tools/wave_simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(768) void probe(unsigned *out) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6;
  unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID = 4, offset 0, size 32
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + wave] = id;
}
int main() {
  unsigned *d, h[12 * 8];
  hipMalloc(&d, sizeof(h));
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  probe<<<8, 768, 150 * 1024>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 8; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 12; ++w) printf("  w%d simd%u(wv%u,cu%u)", w, (h[b * 12 + w] >> 4) & 3, h[b * 12 + w] & 15, (h[b * 12 + w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
