// Micro-benchmark: VALU issue cost per wave64 instruction on gfx950 at 1/2/4 waves per SIMD.
// Not part of the product; used to calibrate the VALU-bound estimates in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int OP>
__global__ void k(float *out, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
        a6 = a0 + 6, a7 = a0 + 7;
  const float c = 0.999f, dd = 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) {  // v_fma_f32
        a0 = fmaf(a0, c, dd); a1 = fmaf(a1, c, dd); a2 = fmaf(a2, c, dd); a3 = fmaf(a3, c, dd);
        a4 = fmaf(a4, c, dd); a5 = fmaf(a5, c, dd); a6 = fmaf(a6, c, dd); a7 = fmaf(a7, c, dd);
      } else if (OP == 1) {  // v_exp_f32
        a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2);
        a3 = __builtin_amdgcn_exp2f(a3); a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5);
        a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
      } else if (OP == 2) {  // v_rcp_f32
        a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2);
        a3 = __builtin_amdgcn_rcpf(a3); a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5);
        a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
      } else if (OP == 3) {  // v_pk_fma_f32 (two floats per instruction)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 x0 = {a0, a1}, x1 = {a2, a3}, x2 = {a4, a5}, x3 = {a6, a7}, cc = {c, c}, d2 = {dd, dd};
        x0 = __builtin_elementwise_fma(x0, cc, d2); x1 = __builtin_elementwise_fma(x1, cc, d2);
        x2 = __builtin_elementwise_fma(x2, cc, d2); x3 = __builtin_elementwise_fma(x3, cc, d2);
        x0 = __builtin_elementwise_fma(x0, cc, d2); x1 = __builtin_elementwise_fma(x1, cc, d2);
        x2 = __builtin_elementwise_fma(x2, cc, d2); x3 = __builtin_elementwise_fma(x3, cc, d2);
        a0 = x0[0]; a1 = x0[1]; a2 = x1[0]; a3 = x1[1]; a4 = x2[0]; a5 = x2[1]; a6 = x3[0]; a7 = x3[1];
      } else if (OP == 4) {  // v_add_f32 with DPP operand
#define DPPADD(x) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true))
        DPPADD(a0); DPPADD(a1); DPPADD(a2); DPPADD(a3); DPPADD(a4); DPPADD(a5); DPPADD(a6); DPPADD(a7);
      } else if (OP == 5) {  // mixed: 6 fma + 2 exp
        a0 = fmaf(a0, c, dd); a1 = fmaf(a1, c, dd); a2 = fmaf(a2, c, dd); a3 = __builtin_amdgcn_exp2f(a3);
        a4 = fmaf(a4, c, dd); a5 = fmaf(a5, c, dd); a6 = fmaf(a6, c, dd); a7 = __builtin_amdgcn_exp2f(a7);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char *name, int waves_per_simd) {
  int dev_cus = 256;
  const int threads = 256 * waves_per_simd;  // 256 threads = 1 wave per SIMD on one CU
  float *out;
  hipMalloc(&out, (size_t)dev_cus * threads * sizeof(float));
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<dev_cus, threads>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<dev_cus, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_wave = (double)iters * REP * (OP == 3 ? 1.0 : 1.0);
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-14s waves/SIMD=%d  %.2f ms  => %.2f cycles per wave-instruction per SIMD (@2.4GHz), per-wave %.2f\n", name,
         waves_per_simd, ms, cycles / (insts_per_wave * waves_per_simd), cycles / insts_per_wave);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w);
    run<1>("v_exp_f32", w);
    run<2>("v_rcp_f32", w);
    run<3>("v_pk_fma_f32", w);
    run<4>("v_add_dpp", w);
    run<5>("6fma+2exp", w);
  }
  return 0;
}
