// Probe (gfx950): is a VALU read of the result of v_cvt_pk_f16_f32 / v_cvt_f16_f32 issued
// back-to-back by the same wave safe?  Each variant converts x -> f16 -> f32 with the dependent
// instruction immediately behind the producer (inline asm pins the order) and counts lanes whose
// result differs from the same conversion done with s_nop padding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VARIANT>
__global__ void probe(const float *x, unsigned *bad, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned count = 0;
  float a = x[i], b = x[i] * 1.7f + 0.3f;
  for (int it = 0; it < iters; ++it) {
    float r0, r1, s0, s1;
    unsigned pk;
    // reference: padded
    asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\ts_nop 7\n\tv_cvt_f32_f16_e32 %1, %0\n\t"
                 "v_cvt_f32_f16_sdwa %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\ts_nop 7"
                 : "=&v"(pk), "=&v"(s0), "=&v"(s1) : "v"(a), "v"(b));
    if (VARIANT == 0) {  // cvt_pk -> cvt back, back to back
      asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\tv_cvt_f32_f16_e32 %1, %0\n\t"
                   "v_cvt_f32_f16_sdwa %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\ts_nop 7"
                   : "=&v"(pk), "=&v"(r0), "=&v"(r1) : "v"(a), "v"(b));
    } else if (VARIANT == 1) {  // one independent instruction between
      float dummy;
      asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_mov_b32 %3, %4\n\tv_cvt_f32_f16_e32 %1, %0\n\t"
                   "v_cvt_f32_f16_sdwa %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\ts_nop 7"
                   : "=&v"(pk), "=&v"(r0), "=&v"(r1), "=&v"(dummy) : "v"(a), "v"(b));
    } else if (VARIANT == 2) {  // single cvt_f16 -> cvt back, back to back
      unsigned h0, h1;
      asm volatile("v_cvt_f16_f32_e32 %0, %4\n\tv_cvt_f32_f16_e32 %2, %0\n\tv_cvt_f16_f32_e32 %1, %5\n\t"
                   "v_cvt_f32_f16_e32 %3, %1\n\ts_nop 7"
                   : "=&v"(h0), "=&v"(h1), "=&v"(r0), "=&v"(r1) : "v"(a), "v"(b));
    } else if (VARIANT == 3) {  // cvt_pk -> v_pk_add reading the converted-back values immediately
      asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\tv_cvt_f32_f16_sdwa %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                   "v_cvt_f32_f16_e32 %1, %0\n\ts_nop 7"
                   : "=&v"(pk), "=&v"(r0), "=&v"(r1) : "v"(a), "v"(b));
    }
    count += (__float_as_uint(r0) != __float_as_uint(s0)) + (__float_as_uint(r1) != __float_as_uint(s1));
    a = a * 1.0001f + 1e-3f;
    b = b * 0.9999f - 1e-3f;
  }
  if (count) atomicAdd(bad, count);
}

int main() {
  const int blocks = 1024, threads = 512, iters = 20000;
  std::vector<float> hx(blocks * threads);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 100000) * 1e-5f - 0.5f;
  float *x;
  unsigned *bad;
  hipMalloc(&x, hx.size() * 4);
  hipMalloc(&bad, 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  const char *names[4] = {"cvt_pk -> cvt_f32 (lo, hi sdwa) back to back", "one v_mov in between", "v_cvt_f16_f32 -> v_cvt_f32_f16 back to back",
                          "cvt_pk -> sdwa hi first"};
  for (int v = 0; v < 4; ++v) {
    hipMemset(bad, 0, 4);
    if (v == 0) probe<0><<<blocks, threads>>>(x, bad, iters);
    if (v == 1) probe<1><<<blocks, threads>>>(x, bad, iters);
    if (v == 2) probe<2><<<blocks, threads>>>(x, bad, iters);
    if (v == 3) probe<3><<<blocks, threads>>>(x, bad, iters);
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("variant %d (%s): %u mismatching results of %.3g\n", v, names[v], h, 2.0 * blocks * threads * (double)iters);
  }
  return 0;
}
