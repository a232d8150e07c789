// Does fp32 MFMA issued by one wave overlap VALU work of another wave on the same SIMD (gfx950)?
// 512-thread workgroups, one per CU: waves 0-3 run an MFMA chain, waves 4-7 a VALU chain
// (one of each per SIMD).  mode 1 = MFMA only, 2 = VALU only, 3 = both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      if (SHAPE == 16) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int i = 0; i < iters; ++i) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
      } else if (SHAPE == 1616) {  // bf16 16x16x16 (matrix core proper)
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        s16x4 a = {(short)threadIdx.x, 0x3f80, 0x3f00, 0x3e80}, b = {0x3f80, (short)(threadIdx.x * 3), 0x3f00, 0x3f80};
        for (int i = 0; i < iters; ++i) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
      } else {
        f32x16 c0 = {0}, c1 = c0;
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int i = 0; i < iters; ++i) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        }
        r = c0[0] + c1[1];
      }
    }
  } else {
    if (mode & 2) {
      float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
      const float m = 0.999f, c = 0.001f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
          x4 = fmaf(x4, m, c); x5 = fmaf(x5, m, c); x6 = fmaf(x6, m, c); x7 = fmaf(x7, m, c);
        }
      }
      r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int SHAPE>
void run(const char *name, float *d, int iters) {
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE><<<256, 512>>>(d, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE><<<256, 512>>>(d, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (SHAPE == 32 ? 2.0 : 4.0) * iters, vf = 32.0 * iters;
    printf("%s mode %d (%s): %.3f ms  -> per MFMA %.1f ns, per VALU fma %.2f ns\n", name, mode,
           mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms, ms * 1e6 / mf, ms * 1e6 / vf);
  }
}
int main() {
  float *d; hipMalloc(&d, 4096);
  run<16>("16x16x4f32", d, 20000);
  run<32>("32x32x2f32", d, 20000);
  run<1616>("16x16x16bf16", d, 20000);
  return 0;
}
