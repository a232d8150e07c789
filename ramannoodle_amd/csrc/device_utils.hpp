// Device-side helpers shared by the PotGNN kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>

namespace rn {

constexpr int kWave = 64;

// ---------------------------------------------------------------- lane-group reductions
// A "lane group" is an aligned run of LG consecutive lanes (LG = 4, 8, 16 or 32) that
// together own one embedding row: lane q of the group holds columns 4q..4q+3.
// All-reduce sums use DPP butterflies (one v_add_f32 with a DPP operand per step); only
// the LG = 32 step has to leave the 16-lane DPP row and uses ds_swizzle.

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float swizzle_xor16(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}
__device__ __forceinline__ double swizzle_xor16(double v) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_ds_swizzle((int)(b & 0xffffffffLL), 0x401F);
  int hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), 0x401F);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int LG, typename T>
__device__ __forceinline__ T lg_sum(T v) {
  static_assert(LG == 4 || LG == 8 || LG == 16 || LG == 32, "lane group size");
  v += dpp_mov<0xB1>(v);                        // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);                        // quad_perm [2,3,0,1]
  if constexpr (LG >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if constexpr (LG >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  if constexpr (LG >= 32) v += swizzle_xor16(v);
  return v;
}

// ---------------------------------------------------------------- scalar math
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double fast_rsq(double x) { return 1.0 / sqrt(x); }

// sigmoid(f) * tanh(c) for already layer-normalised pre-activations.
//   sigmoid(f) tanh(c) = (e2 - 1) / ((1 + e1)(1 + e2)),  e1 = exp(-f), e2 = exp(2c)
// c is clamped to +-15 (tanh is 1 to 13 digits there) so e2 cannot overflow.
__device__ __forceinline__ float gate(float f, float c) {
  const float kLog2e = 1.4426950408889634f;
  float e1 = fast_exp2(-kLog2e * f);
  c = fminf(fmaxf(c, -15.0f), 15.0f);
  float e2 = fast_exp2((2.0f * kLog2e) * c);
  float den = (1.0f + e1) * (1.0f + e2);
  return (e2 - 1.0f) * fast_rcp(den);
}
// float64: the same identity with one division, sigmoid(f) tanh(c) = expm1(2c) / ((1 + e^-f)(expm1(2c) + 2));
// expm1 keeps tanh's relative accuracy near c = 0, and beyond |c| = 20 tanh is +-1 to 17 digits.
__device__ __forceinline__ double gate(double f, double c) {
  c = c > 20.0 ? 20.0 : (c < -20.0 ? -20.0 : c);  // (comparisons, not fmin / fmax: a NaN input stays a NaN)
  const double em = expm1(2.0 * c);
  return em / ((1.0 + exp(-f)) * (em + 2.0));
}

__device__ __forceinline__ float acc_tanh(float x) { return tanhf(x); }
// tanh from the hardware exp2 / rcp: 1 - 2 / (1 + e^{2x}); absolute error ~1e-7 (one fp32
// rounding of a value in [-1, 1]), saturates correctly at both ends.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = fast_exp2(x * (2.0f * 1.4426950408889634f));
  return 1.0f - 2.0f * fast_rcp(1.0f + e);
}
__device__ __forceinline__ double acc_tanh(double x) { return tanh(x); }

// ShiftedSoftplus: softplus(x) - log 2, softplus with torch's threshold of 20.
__device__ __forceinline__ float ssp(float x) {
  float sp = (x > 20.0f) ? x : log1pf(expf(x));
  return sp - 0.6931471805599453f;
}
// Same function from the hardware exp2/log2 (abs error ~1e-7): used in the projection
// epilogues where the libm form made the kernel VALU-bound.
__device__ __forceinline__ float ssp_fast(float x) {
  const float t = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
  const float sp = (x > 20.0f) ? x : 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + t);
  return sp - 0.6931471805599453f;
}
__device__ __forceinline__ double ssp(double x) {
  double sp = (x > 20.0) ? x : log1p(exp(x));
  return sp - 0.6931471805599453;
}

// ---------------------------------------------------------------- row fragments
// Four consecutive columns of a row, held by one lane.
template <typename T>
struct Vec4 {
  T v[4];
};
template <typename T>
__device__ __forceinline__ Vec4<T> load4(const T *p);
template <>
__device__ __forceinline__ Vec4<float> load4<float>(const float *p) {
  float4 t = *reinterpret_cast<const float4 *>(p);
  return {{t.x, t.y, t.z, t.w}};
}
template <>
__device__ __forceinline__ Vec4<double> load4<double>(const double *p) {
  double2 a = *reinterpret_cast<const double2 *>(p);
  double2 b = *reinterpret_cast<const double2 *>(p + 2);
  return {{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void store4(float *p, const Vec4<float> &x) {
  *reinterpret_cast<float4 *>(p) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
}
__device__ __forceinline__ void store4(double *p, const Vec4<double> &x) {
  *reinterpret_cast<double2 *>(p) = make_double2(x.v[0], x.v[1]);
  *reinterpret_cast<double2 *>(p + 2) = make_double2(x.v[2], x.v[3]);
}

// LayerNorm parameters for the 4 (or 4+4) columns a lane owns.
template <typename T>
struct LnParams {
  Vec4<T> g, b;
};

// LayerNorm over a [filter | core] row of logical width 2F spread over a lane group
// (lane holds 4 filter + 4 core columns), followed by the sigmoid*tanh gate.
// `nvalid` = number of this lane's 4 columns that are real (< F); padded columns hold
// exact zeros on input and are excluded from the variance.
template <int LG, bool PAD, typename T>
__device__ __forceinline__ Vec4<T> ln_gate(const Vec4<T> &xf, const Vec4<T> &xc,
                                            const LnParams<T> &pf, const LnParams<T> &pc,
                                            T inv_n, int nvalid) {
  T s = (xf.v[0] + xf.v[1]) + (xf.v[2] + xf.v[3]) + (xc.v[0] + xc.v[1]) + (xc.v[2] + xc.v[3]);
  s = lg_sum<LG>(s);
  const T mean = s * inv_n;
  T df[4], dc[4];
  T q = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    df[i] = xf.v[i] - mean;
    dc[i] = xc.v[i] - mean;
    if (PAD) {
      if (i >= nvalid) { df[i] = 0; dc[i] = 0; }
    }
    q += df[i] * df[i];
    q += dc[i] * dc[i];
  }
  q = lg_sum<LG>(q);
  const T rstd = fast_rsq(q * inv_n + (T)1e-5);
  Vec4<T> out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T yf = df[i] * rstd * pf.g.v[i] + pf.b.v[i];
    T yc = dc[i] * rstd * pc.g.v[i] + pc.b.v[i];
    out.v[i] = gate(yf, yc);
  }
  return out;
}

// The same for a row that has zero mean by construction (a Linear whose weights and bias were centred over the row's
// real columns) and exact zeros in its padded columns: the variance is the plain sum of squares -- no mean, no masking.
template <int LG, typename T>
__device__ __forceinline__ Vec4<T> ln_gate_zero_mean(const Vec4<T> &xf, const Vec4<T> &xc, const LnParams<T> &pf,
                                                      const LnParams<T> &pc, T inv_n) {
  T q = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) q += xf.v[i] * xf.v[i] + xc.v[i] * xc.v[i];
  q = lg_sum<LG>(q);
  const T rstd = fast_rsq(q * inv_n + (T)1e-5);
  Vec4<T> out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out.v[i] = gate(xf.v[i] * rstd * pf.g.v[i] + pf.b.v[i], xc.v[i] * rstd * pc.g.v[i] + pc.b.v[i]);
  return out;
}

// LayerNorm over a row of logical width F (lane holds 4 columns).
template <int LG, bool PAD, typename T>
__device__ __forceinline__ Vec4<T> ln_row(const Vec4<T> &x, const LnParams<T> &p, T inv_n,
                                           int nvalid) {
  T s = (x.v[0] + x.v[1]) + (x.v[2] + x.v[3]);
  s = lg_sum<LG>(s);
  const T mean = s * inv_n;
  T d[4];
  T q = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d[i] = x.v[i] - mean;
    if (PAD) {
      if (i >= nvalid) d[i] = 0;
    }
    q += d[i] * d[i];
  }
  q = lg_sum<LG>(q);
  const T rstd = fast_rsq(q * inv_n + (T)1e-5);
  Vec4<T> out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out.v[i] = d[i] * rstd * p.g.v[i] + p.b.v[i];
  return out;
}

// ---------------------------------------------------------------- split-f16 matrix products
// The f32-input MFMA runs at the vector rate (1/16 of the f16/bf16 rate) and blocks the SIMD's
// VALU issue while it does.  A two-way f16 split  x = hi + lo  (hi = f16(x), lo = f16(x - hi):
// 22 significant bits, absolute error <= 2^-25 below the f16 normal range) and the three products
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on the f16 MFMA (f16 products are exact in f32,
// accumulation in f32) reproduce the f32 product to ~3e-7 * sum|a||b| -- the size of f32's own
// rounding in a K = 64 dot product -- in 3/16 of the matrix-pipe time, and leave the VALU free.
// Operand values here are tanh outputs / normalised rows and O(1) weights: far inside f16 range.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// 8 consecutive k of one operand row -> (hi, lo) fragments of one K = 32 slice
__device__ __forceinline__ void split_f16x8(const float *x, f16x8 &hi, f16x8 &lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = (_Float16)x[j];
    lo[j] = (_Float16)(x[j] - (float)hi[j]);
  }
}
// acc += A * B over one K = 32 slice (8 k per lane), smallest terms first.
// K32 = true: one v_mfma_f32_16x16x32_f16 (new on gfx950) per product.  K32 = false: each product as TWO
// v_mfma_f32_16x16x16_f16 (the halves of the 8-element fragments; A and B use the same lane-local element order, so the k
// pairing is consistent) at half the matrix-pipe rate.
//
// History (profiles/r02/determinism.txt, r03/determinism.txt, r04/mfma_k32.txt): with the K = 32 instruction the per-frame
// fused EdgeBlock (edge_block_fused_kernel: two 4-wave workgroups per CU, every wave both multiplies and runs the triplet
// loop) is not reproducible under hipcc / ROCm 7.2 -- ~80 % of the frames of a 10 000-frame trajectory differ between two
// evaluations -- while every other kernel of the library is: round 4 re-ran the probe on the K = 32 build with the
// role-specialised EdgeBlock (one workgroup per CU, MFMAs only in the producer waves) and found 0 differing frames in
// 5 x 4 x 10 000 evaluations (fused NodeBlock at four workgroups per CU, fused readout, split projections, unfused chain).
// So K = 32 is the default everywhere EXCEPT in edge_block_fused_kernel, which keeps the two-instruction form
// (WaveB::product_split16) and stays bit-reproducible; -DRN_MFMA_K32=0 builds the whole library on K = 16.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#ifndef RN_MFMA_K32
#define RN_MFMA_K32 1
#endif
template <bool K32>
__device__ __forceinline__ f32x4_t mfma_split3_t(const f16x8 &ah, const f16x8 &al, const f16x8 &bh, const f16x8 &bl,
                                                 f32x4_t acc) {
  if constexpr (K32) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  } else {
    const f16x4 ah0 = ah.lo, ah1 = ah.hi, al0 = al.lo, al1 = al.hi;
    const f16x4 bh0 = bh.lo, bh1 = bh.hi, bl0 = bl.lo, bl1 = bl.hi;
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al0, bh0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al1, bh1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah0, bl0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah1, bl1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah0, bh0, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(ah1, bh1, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ f32x4_t mfma_split3(const f16x8 &ah, const f16x8 &al, const f16x8 &bh, const f16x8 &bl,
                                               f32x4_t acc) {
  return mfma_split3_t<(RN_MFMA_K32 != 0)>(ah, al, bh, bl, acc);
}

}  // namespace rn
