// Reverse-mode kernels of the PotGNN forward pass with respect to ACTIVATIONS and atomic
// positions (no weight gradients): they give d(alpha)/d(r), the quantity the Raman-tensor
// contraction needs (the reference obtains it by finite differences,
// ramannoodle/dynamics/_phonon.py:93-106).
//
// Correctness-first kernels (the Jacobian is taken at ONE structure per model, so they are
// never on a throughput path): same lane-group row layout as the forward kernels, rows
// read straight from global memory, cross-row accumulations by float/double atomics.
//
// Indexing: cotangent "instance" c in [0, C) belongs to forward frame s = c / B (B
// cotangents per frame: 6 for the Jacobian of the six polarizability components).
// Forward arrays are indexed by s, cotangent arrays by c.
#include <cstdlib>

#include "device_utils.hpp"
#include "kernels.hpp"

namespace rn {

template <typename T>
__device__ __forceinline__ void atomic_add4(T *p, const Vec4<T> &v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) atomicAdd(p + k, v.v[k]);
}

// Parameter-gradient accumulators (LayerNorm gamma / beta) are per lane group; every group of
// every workgroup adding them straight to the same few hundred global addresses serialises in
// the memory-side atomic unit (measured: 0.3-0.8 ms per launch at 8 k adders per address,
// more than the rest of the kernel at training batch sizes).  Sum the G = 256/LG groups of
// the workgroup in LDS first: one atomic per column and workgroup.  `scratch`: 4 NT elements.
// Must be reached by all NT threads of the workgroup.
template <int LG, typename T, int NT = 256>
__device__ __forceinline__ void wg_sum_atomic_add(T *scratch, const Vec4<T> &v, T *dst) {
  constexpr int W = 4 * LG, G = NT / LG;
  const int grp = threadIdx.x / LG, q = threadIdx.x % LG;
  store4(scratch + grp * W + 4 * q, v);
  __syncthreads();
  if ((int)threadIdx.x < W) {
    T s = 0;
#pragma unroll 4
    for (int g = 0; g < G; ++g) s += scratch[g * W + threadIdx.x];
    atomicAdd(dst + threadIdx.x, s);
  }
  __syncthreads();
}

// ---- LayerNorm pieces on the lane-group layout -------------------------------------------
// forward: normalised values (padded columns -> 0) and 1/sigma of a [filter|core] row
template <int LG, typename T>
__device__ __forceinline__ T ln2_hat(Vec4<T> &xf, Vec4<T> &xc, T inv_n, int nvalid) {
  T s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += xf.v[k] + xc.v[k];
  const T mean = lg_sum<LG>(s) * inv_n;
  T q = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xf.v[k] = k < nvalid ? xf.v[k] - mean : (T)0;
    xc.v[k] = k < nvalid ? xc.v[k] - mean : (T)0;
    q += xf.v[k] * xf.v[k] + xc.v[k] * xc.v[k];
  }
  const T rstd = (T)1 / sqrt(lg_sum<LG>(q) * inv_n + (T)1e-5);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xf.v[k] *= rstd;
    xc.v[k] *= rstd;
  }
  return rstd;
}
// backward: dxhat (= dy * gamma) -> dx, given xhat and 1/sigma
template <int LG, typename T>
__device__ __forceinline__ void ln2_bwd(Vec4<T> &df, Vec4<T> &dc, const Vec4<T> &hf,
                                        const Vec4<T> &hc, T rstd, T inv_n, int nvalid) {
  T a = 0, b = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= nvalid) { df.v[k] = 0; dc.v[k] = 0; }
    a += df.v[k] + dc.v[k];
    b += df.v[k] * hf.v[k] + dc.v[k] * hc.v[k];
  }
  a = lg_sum<LG>(a) * inv_n;
  b = lg_sum<LG>(b) * inv_n;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    df.v[k] = k < nvalid ? rstd * (df.v[k] - a - hf.v[k] * b) : (T)0;
    dc.v[k] = k < nvalid ? rstd * (dc.v[k] - a - hc.v[k] * b) : (T)0;
  }
}
template <int LG, typename T>
__device__ __forceinline__ T ln1_hat(Vec4<T> &x, T inv_n, int nvalid) {
  T s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += x.v[k];
  const T mean = lg_sum<LG>(s) * inv_n;
  T q = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x.v[k] = k < nvalid ? x.v[k] - mean : (T)0;
    q += x.v[k] * x.v[k];
  }
  const T rstd = (T)1 / sqrt(lg_sum<LG>(q) * inv_n + (T)1e-5);
#pragma unroll
  for (int k = 0; k < 4; ++k) x.v[k] *= rstd;
  return rstd;
}
template <int LG, typename T>
__device__ __forceinline__ void ln1_bwd(Vec4<T> &d, const Vec4<T> &h, T rstd, T inv_n, int nvalid) {
  T a = 0, b = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= nvalid) d.v[k] = 0;
    a += d.v[k];
    b += d.v[k] * h.v[k];
  }
  a = lg_sum<LG>(a) * inv_n;
  b = lg_sum<LG>(b) * inv_n;
#pragma unroll
  for (int k = 0; k < 4; ++k) d.v[k] = k < nvalid ? rstd * (d.v[k] - a - h.v[k] * b) : (T)0;
}
template <typename T>
__device__ __forceinline__ T sigmoid_acc(T x) { return (T)1 / ((T)1 + exp(-x)); }

// gate value and its partial derivatives for normalised, affine-transformed inputs
template <typename T>
__device__ __forceinline__ void gate_grad(T yf, T yc, T &g, T &dgf, T &dgc) {
  const T sg = sigmoid_acc(yf), th = tanh(yc);
  g = sg * th;
  dgf = th * sg * ((T)1 - sg);
  dgc = sg * ((T)1 - th * th);
}

// =========================================================================== small GEMM
// Y[r][k] (+)= sum_n X[r][n] * W[k*ldw + n]   (W in the forward's [K][N] layout, so this is
// the activation-gradient product dX = dY * W^T of Y = X * W).
template <typename T>
__global__ void gemm_nt_kernel(const T *__restrict__ X, int64_t R, int N, const T *__restrict__ W,
                               int ldw, int K, T *__restrict__ Y, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * K) return;
  const int64_t r = idx / K;
  const int k = (int)(idx % K);
  const T *x = X + r * N;
  const T *w = W + (int64_t)k * ldw;
  T acc = 0;
  for (int n = 0; n < N; ++n) acc += x[n] * w[n];
  if (accumulate) Y[idx] += acc;
  else Y[idx] = acc;
}
template <typename T>
void launch_gemm_nt(const T *X, int64_t R, int N, const T *W, int ldw, int K, T *Y, bool accumulate,
                    hipStream_t st) {
  if (R == 0) return;
  const int64_t total = R * K;
  gemm_nt_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(X, R, N, W, ldw, K, Y,
                                                                     accumulate ? 1 : 0);
}
template void launch_gemm_nt<float>(const float *, int64_t, int, const float *, int, int, float *,
                                    bool, hipStream_t);
template void launch_gemm_nt<double>(const double *, int64_t, int, const double *, int, int,
                                     double *, bool, hipStream_t);

// =========================================================================== readout
// out_k = (1/E) sum_e v_k(pol_e, unit_e) (closed form of _gnn.py:372-415).  Given dL/dout
// (one 6-vector per cotangent instance) produce dL/dpol [C*E,32] and dL/dunit [C*E,4].
template <typename T>
__global__ void readout_bwd_kernel(const T *__restrict__ dout6, const T *__restrict__ pol,
                                   const T *__restrict__ unit4, int C, int B, Graph g,
                                   T *__restrict__ dpol, T *__restrict__ dunit) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)C * g.E) return;
  const int c = (int)(idx / g.E), e = (int)(idx % g.E);
  const int64_t frow = (int64_t)(c / B) * g.E + e;
  const T inv = (T)1 / (T)g.E;
  T dv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) dv[k] = dout6[c * 6 + k] * inv;
  const T ux = unit4[frow * 4], uy = unit4[frow * 4 + 1], uz = unit4[frow * 4 + 2];
  const T *m = pol + frow * 32;
  T *dp = dpol + idx * 32;
  for (int k = 12; k < 32; ++k) dp[k] = 0;
  // v3 = (m0-m1) ux uy ; v4 = (m2-m3) ux uz ; v5 = (m4-m5) uy uz
  // v0 = m7 + (m6-m7) ux^2 ; v1 = m9 + (m8-m9) uy^2 ; v2 = m11 + (m10-m11) uz^2
  dp[0] = dv[3] * ux * uy;  dp[1] = -dp[0];
  dp[2] = dv[4] * ux * uz;  dp[3] = -dp[2];
  dp[4] = dv[5] * uy * uz;  dp[5] = -dp[4];
  dp[6] = dv[0] * ux * ux;  dp[7] = dv[0] * ((T)1 - ux * ux);
  dp[8] = dv[1] * uy * uy;  dp[9] = dv[1] * ((T)1 - uy * uy);
  dp[10] = dv[2] * uz * uz; dp[11] = dv[2] * ((T)1 - uz * uz);
  const T a3 = m[0] - m[1], a4 = m[2] - m[3], a5 = m[4] - m[5];
  const T a0 = m[6] - m[7], a1 = m[8] - m[9], a2 = m[10] - m[11];
  T *du = dunit + idx * 4;
  du[0] = dv[3] * a3 * uy + dv[4] * a4 * uz + (T)2 * dv[0] * a0 * ux;
  du[1] = dv[3] * a3 * ux + dv[5] * a5 * uz + (T)2 * dv[1] * a1 * uy;
  du[2] = dv[4] * a4 * ux + dv[5] * a5 * uy + (T)2 * dv[2] * a2 * uz;
  du[3] = 0;
}
template <typename T>
void launch_readout_bwd(const T *dout6, const T *pol, const T *unit4, int C, int B, const Graph &g,
                        T *dpol, T *dunit, hipStream_t st) {
  const int64_t total = (int64_t)C * g.E;
  if (total == 0) return;
  readout_bwd_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(dout6, pol, unit4, C, B, g,
                                                                         dpol, dunit);
}
template void launch_readout_bwd<float>(const float *, const float *, const float *, int, int,
                                        const Graph &, float *, float *, hipStream_t);
template void launch_readout_bwd<double>(const double *, const double *, const double *, int, int,
                                         const Graph &, double *, double *, hipStream_t);

// dz = dh * ssp'(z) * scale, with ssp'(z) = sigmoid(z) = 1 - exp(-(h + ln2)) expressed through
// the stored activation h = ssp(z).  `h` is indexed by forward row, `d` by cotangent row.
// One thread per four columns of a row (width a multiple of 4): the row arithmetic (64-bit divisions) once per
// sixteen bytes instead of once per element.
template <typename T>
__global__ void ssp_bwd_kernel(T *__restrict__ d, const T *__restrict__ h, const T *__restrict__ scale,
                               int64_t rows_per_frame, int width, int C, int B) {
  const int cg = width / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)C * rows_per_frame * cg;
  if (idx >= total) return;
  const int col = (int)(idx % cg) * 4;
  const int64_t crow = idx / cg;
  int64_t frow = crow;
  if (B != 1) {
    const int c = (int)(crow / rows_per_frame);
    frow = (int64_t)(c / B) * rows_per_frame + crow % rows_per_frame;
  }
  const Vec4<T> hv = load4<T>(h + frow * width + col);
  Vec4<T> dv = load4<T>(d + crow * width + col);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T sg = (T)1 - (T)0.5 * exp(-hv.v[k]);
    dv.v[k] *= sg * (scale ? scale[col + k] : (T)1);
  }
  store4(d + crow * width + col, dv);
}
template <typename T>
void launch_ssp_bwd(T *d, const T *h, const T *scale, int64_t rows_per_frame, int width, int C,
                    int B, hipStream_t st) {
  const int64_t total = (int64_t)C * rows_per_frame * (width / 4);
  if (total == 0) return;
  ssp_bwd_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(d, h, scale, rows_per_frame,
                                                                     width, C, B);
}
template void launch_ssp_bwd<float>(float *, const float *, const float *, int64_t, int, int, int,
                                    hipStream_t);
template void launch_ssp_bwd<double>(double *, const double *, const double *, int64_t, int, int, int,
                                     hipStream_t);

// =========================================================================== edge block
// Reverse of _EdgeBlock.forward (_gnn.py:294-351) for one destination edge d per lane group.
// (Straightforward version: everything from global memory, dQ by global atomics.  Kept as
// the cross-check of the tiled kernel below; RN_POTGNN_BWD_SIMPLE=1 selects it.)
//   in : dedge_next [C*E,FP]  (cotangent of the block's output)
//   out: dedge_prev [C*E,FP]  = residual part (the projection parts are added by GEMMs later)
//        dpq [C*E,4FP] (first half written here, second half = dQ accumulated atomically)
//        dnp3 [C*N,6FP] (atomic), dc2pre [C*E,2FP]
template <int LG, typename T>
__global__ __launch_bounds__(256) void edge_bwd_simple_kernel(
    const T *__restrict__ pq, const T *__restrict__ np3, const T *__restrict__ c2pre,
    const T *__restrict__ edge_next, const T *__restrict__ dedge_next, T *__restrict__ dedge_prev,
    T *__restrict__ dpq, T *__restrict__ dnp3, T *__restrict__ dc2pre, int C, int B, Graph g, Dims d,
    PassW<T> w, PassW<T> gw, int want_param_grads) {
  constexpr int FP = LG * 4;
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LG;
  const int q = threadIdx.x % LG;
  if (gid >= (int64_t)C * g.E) return;
  const int c = (int)(gid / g.E), dst = (int)(gid % g.E);
  // LayerNorm parameter gradients of this lane's columns (training only)
  Vec4<T> G31f{{0, 0, 0, 0}}, B31f = G31f, G31c = G31f, B31c = G31f, G32 = G31f, B32 = G31f;
  Vec4<T> G21f = G31f, B21f = G31f, G21c = G31f, B21c = G31f, G22 = G31f, B22 = G31f;
  const int s = c / B;
  const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
  const int64_t cerow0 = (int64_t)c * g.E, cnrow0 = (int64_t)c * g.N;
  const int nvalid = min(max(d.Fe - 4 * q, 0), 4);
  const T inv2n = (T)1 / (T)(2 * d.Fe), invn = (T)1 / (T)d.Fe;
  const int ad = g.edge_a[dst], bd = g.edge_b[dst];
  const int64_t drow = erow0 + dst;

  // tanh residual
  const Vec4<T> e1 = load4<T>(edge_next + drow * FP + 4 * q);
  Vec4<T> dz = load4<T>(dedge_next + (cerow0 + dst) * FP + 4 * q);
#pragma unroll
  for (int k = 0; k < 4; ++k) dz.v[k] *= ((T)1 - e1.v[k] * e1.v[k]);
  store4(dedge_prev + (cerow0 + dst) * FP + 4 * q, dz);

  // P'_d
  Vec4<T> pf = load4<T>(pq + drow * (4 * FP) + 4 * q), pc = load4<T>(pq + drow * (4 * FP) + FP + 4 * q);
  {
    const T *nj = np3 + (nrow0 + bd) * (6 * FP) + 2 * FP + 4 * q;
    const T *nk = np3 + (nrow0 + ad) * (6 * FP) + 4 * FP + 4 * q;
    const Vec4<T> jf = load4<T>(nj), jc = load4<T>(nj + FP), kf = load4<T>(nk), kc = load4<T>(nk + FP);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pf.v[k] += jf.v[k] + kf.v[k];
      pc.v[k] += jc.v[k] + kc.v[k];
    }
  }
  const Vec4<T> g1f = load4<T>(w.c3_norm_1.g + 4 * q), b1f = load4<T>(w.c3_norm_1.b + 4 * q);
  const Vec4<T> g1c = load4<T>(w.c3_norm_1.g + FP + 4 * q), b1c = load4<T>(w.c3_norm_1.b + FP + 4 * q);
  const int rb = g.out_ptr[bd], re = g.out_ptr[bd + 1];

  auto source_row = [&](int e, Vec4<T> &xf, Vec4<T> &xc) {  // x = P'_d + Q'_e
    const T *qp = pq + (erow0 + e) * (4 * FP) + 2 * FP + 4 * q;
    const T *np = np3 + (nrow0 + g.edge_b[e]) * (6 * FP) + 4 * q;
    const Vec4<T> qf = load4<T>(qp), qc = load4<T>(qp + FP), nf = load4<T>(np), nc = load4<T>(np + FP);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xf.v[k] = pf.v[k] + qf.v[k] + nf.v[k];
      xc.v[k] = pc.v[k] + qc.v[k] + nc.v[k];
    }
  };

  // ---- recompute the aggregated gate sum, then LayerNorm (c3_norm_2) backward
  Vec4<T> agg{{0, 0, 0, 0}};
  for (int e = rb; e < re; ++e) {
    if (g.edge_b[e] == ad) continue;
    Vec4<T> xf, xc;
    source_row(e, xf, xc);
    ln2_hat<LG>(xf, xc, inv2n, nvalid);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T gv, a, b;
      gate_grad(xf.v[k] * g1f.v[k] + b1f.v[k], xc.v[k] * g1c.v[k] + b1c.v[k], gv, a, b);
      agg.v[k] += gv;
    }
  }
  Vec4<T> dagg;
  {
    const Vec4<T> g2 = load4<T>(w.c3_norm_2.g + 4 * q);
    Vec4<T> hat = agg;
    const T rstd = ln1_hat<LG>(hat, invn, nvalid);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dagg.v[k] = dz.v[k] * g2.v[k];
      B32.v[k] += dz.v[k];
      G32.v[k] += dz.v[k] * hat.v[k];
    }
    ln1_bwd<LG>(dagg, hat, rstd, invn, nvalid);
  }
  // ---- per triplet: gate + LayerNorm (c3_norm_1) backward; dx goes to P'_d and Q'_e
  Vec4<T> dpf{{0, 0, 0, 0}}, dpc{{0, 0, 0, 0}};
  for (int e = rb; e < re; ++e) {
    if (g.edge_b[e] == ad) continue;
    Vec4<T> xf, xc;
    source_row(e, xf, xc);
    const T rstd = ln2_hat<LG>(xf, xc, inv2n, nvalid);
    Vec4<T> df, dc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T gv, a, b;
      gate_grad(xf.v[k] * g1f.v[k] + b1f.v[k], xc.v[k] * g1c.v[k] + b1c.v[k], gv, a, b);
      const T dyf = dagg.v[k] * a, dyc = dagg.v[k] * b;
      B31f.v[k] += dyf;
      G31f.v[k] += dyf * xf.v[k];
      B31c.v[k] += dyc;
      G31c.v[k] += dyc * xc.v[k];
      df.v[k] = dyf * g1f.v[k];
      dc.v[k] = dyc * g1c.v[k];
    }
    ln2_bwd<LG>(df, dc, xf, xc, rstd, inv2n, nvalid);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dpf.v[k] += df.v[k];
      dpc.v[k] += dc.v[k];
    }
    T *dq = dpq + (cerow0 + e) * (4 * FP) + 2 * FP + 4 * q;  // dQ'_e: many destinations add here
    atomic_add4(dq, df);
    atomic_add4(dq + FP, dc);
  }
  store4(dpq + (cerow0 + dst) * (4 * FP) + 4 * q, dpf);
  store4(dpq + (cerow0 + dst) * (4 * FP) + FP + 4 * q, dpc);
  {
    T *dj = dnp3 + (cnrow0 + bd) * (6 * FP) + 2 * FP + 4 * q;
    T *dk = dnp3 + (cnrow0 + ad) * (6 * FP) + 4 * FP + 4 * q;
    atomic_add4(dj, dpf);
    atomic_add4(dj + FP, dpc);
    atomic_add4(dk, dpf);
    atomic_add4(dk + FP, dpc);
  }
  // ---- c2 = LN(gate(LN(c2pre)))  (_gnn.py:223-228)
  {
    Vec4<T> xf = load4<T>(c2pre + drow * (2 * FP) + 4 * q), xc = load4<T>(c2pre + drow * (2 * FP) + FP + 4 * q);
    const T rstd1 = ln2_hat<LG>(xf, xc, inv2n, nvalid);
    const Vec4<T> gf = load4<T>(w.c2_norm_1.g + 4 * q), bf = load4<T>(w.c2_norm_1.b + 4 * q);
    const Vec4<T> gc = load4<T>(w.c2_norm_1.g + FP + 4 * q), bc = load4<T>(w.c2_norm_1.b + FP + 4 * q);
    Vec4<T> gv, da, db;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      gate_grad(xf.v[k] * gf.v[k] + bf.v[k], xc.v[k] * gc.v[k] + bc.v[k], gv.v[k], da.v[k], db.v[k]);
    Vec4<T> hat = gv;
    const T rstd2 = ln1_hat<LG>(hat, invn, nvalid);
    const Vec4<T> g22 = load4<T>(w.c2_norm_2.g + 4 * q);
    Vec4<T> dg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dg.v[k] = dz.v[k] * g22.v[k];
      B22.v[k] += dz.v[k];
      G22.v[k] += dz.v[k] * hat.v[k];
    }
    ln1_bwd<LG>(dg, hat, rstd2, invn, nvalid);
    Vec4<T> df, dc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T dyf = dg.v[k] * da.v[k], dyc = dg.v[k] * db.v[k];
      B21f.v[k] += dyf;
      G21f.v[k] += dyf * xf.v[k];
      B21c.v[k] += dyc;
      G21c.v[k] += dyc * xc.v[k];
      df.v[k] = dyf * gf.v[k];
      dc.v[k] = dyc * gc.v[k];
    }
    ln2_bwd<LG>(df, dc, xf, xc, rstd1, inv2n, nvalid);
    store4(dc2pre + (cerow0 + dst) * (2 * FP) + 4 * q, df);
    store4(dc2pre + (cerow0 + dst) * (2 * FP) + FP + 4 * q, dc);
  }
  if (want_param_grads) {  // gw mirrors w: same offsets inside the gradient blob
    atomic_add4(const_cast<T *>(gw.c3_norm_1.g) + 4 * q, G31f);
    atomic_add4(const_cast<T *>(gw.c3_norm_1.b) + 4 * q, B31f);
    atomic_add4(const_cast<T *>(gw.c3_norm_1.g) + FP + 4 * q, G31c);
    atomic_add4(const_cast<T *>(gw.c3_norm_1.b) + FP + 4 * q, B31c);
    atomic_add4(const_cast<T *>(gw.c3_norm_2.g) + 4 * q, G32);
    atomic_add4(const_cast<T *>(gw.c3_norm_2.b) + 4 * q, B32);
    atomic_add4(const_cast<T *>(gw.c2_norm_1.g) + 4 * q, G21f);
    atomic_add4(const_cast<T *>(gw.c2_norm_1.b) + 4 * q, B21f);
    atomic_add4(const_cast<T *>(gw.c2_norm_1.g) + FP + 4 * q, G21c);
    atomic_add4(const_cast<T *>(gw.c2_norm_1.b) + FP + 4 * q, B21c);
    atomic_add4(const_cast<T *>(gw.c2_norm_2.g) + 4 * q, G22);
    atomic_add4(const_cast<T *>(gw.c2_norm_2.b) + 4 * q, B22);
  }
}


// ---- tiled EdgeBlock backward ---------------------------------------------------------------
// Same decomposition as the forward edge_agg_kernel: a workgroup owns one atom tile for the
// whole launch (topology in LDS once), stages the tile's centred source rows Q' per cotangent
// instance, and -- new here -- accumulates the source-row cotangents dQ' in LDS (every
// destination edge entering the tile's atoms adds to them; each source row belongs to
// exactly one tile, so they leave the CU with plain stores).  One triplet loop: the
// pre-LayerNorm sums come from the forward tape.
template <typename T>
__device__ __forceinline__ void gate_parts(T yf, T yc, T &sg, T &th);
template <>
__device__ __forceinline__ void gate_parts<float>(float yf, float yc, float &sg, float &th) {
  const float kL = 1.4426950408889634f;
  yc = fminf(fmaxf(yc, -15.0f), 15.0f);
  sg = fast_rcp(1.0f + fast_exp2(-kL * yf));
  th = 1.0f - 2.0f * fast_rcp(1.0f + fast_exp2(2.0f * kL * yc));
}
template <>
__device__ __forceinline__ void gate_parts<double>(double yf, double yc, double &sg, double &th) {
  sg = 1.0 / (1.0 + exp(-yf));
  th = tanh(yc);
}

template <int FP, typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 8 ? 1 : 2)) void edge_bwd_tile_kernel(
    const T *__restrict__ pq, const T *__restrict__ np3, const T *__restrict__ c2pre,
    const T *__restrict__ edge_next, const T *__restrict__ agg_tape, const T *__restrict__ dedge_next,
    T *__restrict__ dedge_prev, T *__restrict__ dpq, T *__restrict__ dnp3, T *__restrict__ dc2pre,
    int C, int B, Graph g, Dims d, PassW<T> w, PassW<T> gw, int want_param_grads) {
  constexpr int LG = FP / 4;
  constexpr int G = 256 / LG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int maxR = g.max_tile_out_rows, maxD = g.max_tile_in_rows, maxN = g.max_tile_nodes;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    unsigned char *p = smem_raw + off;
    off += (bytes + 15) & ~size_t(15);
    return p;
  };
  T *qrows = reinterpret_cast<T *>(carve((size_t)maxR * 2 * FP * sizeof(T)));
  T *dq = reinterpret_cast<T *>(carve((size_t)maxR * 2 * FP * sizeof(T)));
  if (off < (size_t)1024 * sizeof(T)) off = (size_t)1024 * sizeof(T);  // scratch of wg_sum_atomic_add
  T *sq = reinterpret_cast<T *>(carve((size_t)maxR * sizeof(T)));
  T *nj = reinterpret_cast<T *>(carve((size_t)maxN * 2 * FP * sizeof(T)));
  T *dnj = reinterpret_cast<T *>(carve((size_t)maxN * 2 * FP * sizeof(T)));  // d(Wj node[j] + bias), tile atoms
  int *qb = reinterpret_cast<int *>(carve((size_t)maxR * 4));
  int *dl = reinterpret_cast<int *>(carve((size_t)maxD * 6 * 4));
  int *d_edge = dl, *d_a = dl + maxD, *d_bl = dl + 2 * maxD, *d_rb = dl + 3 * maxD,
      *d_cnt = dl + 4 * maxD, *d_skip = dl + 5 * maxD;

  const int tile = blockIdx.x % g.num_tiles;
  const int cg = blockIdx.x / g.num_tiles, ncg = gridDim.x / g.num_tiles;
  const int j0 = g.tile_begin[tile], j1 = g.tile_begin[tile + 1];
  const int eo0 = g.out_ptr[j0], rows = g.out_ptr[j1] - eo0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
  for (int r = threadIdx.x; r < rows; r += 256) qb[r] = g.edge_b[eo0 + r];
  for (int i = threadIdx.x; i < dcount; i += 256) {
    const int dst = g.in_edge[di0 + i];
    const int ad = g.edge_a[dst], bd = g.edge_b[dst];
    const int rb = g.out_ptr[bd] - eo0, re = g.out_ptr[bd + 1] - eo0;
    const int rev = g.rev_edge[dst];
    d_edge[i] = dst;
    d_a[i] = ad;
    d_bl[i] = bd - j0;
    d_rb[i] = rb;
    d_cnt[i] = (re - rb) - (rev >= 0 ? 1 : 0);
    d_skip[i] = rev >= 0 ? rev - eo0 : re;
  }
  __syncthreads();

  const int grp = threadIdx.x / LG, q = threadIdx.x % LG;
  const int nvalid = min(max(d.Fe - 4 * q, 0), 4);
  const T inv2n = (T)1 / (T)(2 * d.Fe), invn = (T)1 / (T)d.Fe;
  const Vec4<T> g1f = load4<T>(w.c3_norm_1.g + 4 * q), b1f = load4<T>(w.c3_norm_1.b + 4 * q);
  const Vec4<T> g1c = load4<T>(w.c3_norm_1.g + FP + 4 * q), b1c = load4<T>(w.c3_norm_1.b + FP + 4 * q);
  Vec4<T> G31f{{0, 0, 0, 0}}, B31f = G31f, G31c = G31f, B31c = G31f, G32 = G31f, B32 = G31f;
  Vec4<T> G21f = G31f, B21f = G31f, G21c = G31f, B21c = G31f, G22 = G31f, B22 = G31f;

  for (int c = cg; c < C; c += ncg) {
    const int s = c / B;
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
    const int64_t cerow0 = (int64_t)c * g.E, cnrow0 = (int64_t)c * g.N;
    for (int i = threadIdx.x; i < (j1 - j0) * (2 * FP / 4); i += 256) {
      const int n = i / (2 * FP / 4), cc = (i % (2 * FP / 4)) * 4;
      store4(nj + (size_t)n * 2 * FP + cc, load4<T>(np3 + (nrow0 + j0 + n) * (6 * FP) + 2 * FP + cc));
      store4(dnj + (size_t)n * 2 * FP + cc, Vec4<T>{{0, 0, 0, 0}});
    }
    for (int r = grp; r < rows; r += G) {  // centred source rows, as in the forward kernel
      const T *qp = pq + (erow0 + eo0 + r) * (4 * FP) + 2 * FP + 4 * q;
      const T *np = np3 + (nrow0 + qb[r]) * (6 * FP) + 4 * q;
      Vec4<T> f = load4<T>(qp), cc = load4<T>(qp + FP);
      const Vec4<T> nf = load4<T>(np), nc = load4<T>(np + FP);
      T sum = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f.v[k] += nf.v[k];
        cc.v[k] += nc.v[k];
        sum += f.v[k] + cc.v[k];
      }
      const T mean = lg_sum<LG>(sum) * inv2n;
      T ss = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f.v[k] = k < nvalid ? f.v[k] - mean : (T)0;
        cc.v[k] = k < nvalid ? cc.v[k] - mean : (T)0;
        ss += f.v[k] * f.v[k] + cc.v[k] * cc.v[k];
      }
      ss = lg_sum<LG>(ss);
      store4(qrows + (size_t)r * 2 * FP + 4 * q, f);
      store4(qrows + (size_t)r * 2 * FP + FP + 4 * q, cc);
      const Vec4<T> zero{{0, 0, 0, 0}};
      store4(dq + (size_t)r * 2 * FP + 4 * q, zero);
      store4(dq + (size_t)r * 2 * FP + FP + 4 * q, zero);
      if (q == 0) sq[r] = ss;
    }
    __syncthreads();

    for (int i = grp; i < dcount; i += G) {
      const int dst = d_edge[i];
      const int64_t drow = erow0 + dst, cdrow = cerow0 + dst;
      const Vec4<T> e1 = load4<T>(edge_next + drow * FP + 4 * q);
      Vec4<T> dz = load4<T>(dedge_next + cdrow * FP + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) dz.v[k] *= ((T)1 - e1.v[k] * e1.v[k]);
      store4(dedge_prev + cdrow * FP + 4 * q, dz);

      // P'_d centred
      Vec4<T> pf = load4<T>(pq + drow * (4 * FP) + 4 * q), pc = load4<T>(pq + drow * (4 * FP) + FP + 4 * q);
      {
        const T *nk = np3 + (nrow0 + d_a[i]) * (6 * FP) + 4 * FP + 4 * q;
        const Vec4<T> kf = load4<T>(nk), kc = load4<T>(nk + FP);
        const Vec4<T> jf = load4<T>(nj + (size_t)d_bl[i] * 2 * FP + 4 * q);
        const Vec4<T> jc = load4<T>(nj + (size_t)d_bl[i] * 2 * FP + FP + 4 * q);
        T sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pf.v[k] += jf.v[k] + kf.v[k];
          pc.v[k] += jc.v[k] + kc.v[k];
          sum += pf.v[k] + pc.v[k];
        }
        const T mean = lg_sum<LG>(sum) * inv2n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pf.v[k] = k < nvalid ? pf.v[k] - mean : (T)0;
          pc.v[k] = k < nvalid ? pc.v[k] - mean : (T)0;
        }
      }
      T sp = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) sp += pf.v[k] * pf.v[k] + pc.v[k] * pc.v[k];
      sp = lg_sum<LG>(sp);

      // LayerNorm (c3_norm_2) backward from the taped sums
      Vec4<T> dagg;
      {
        const Vec4<T> g2 = load4<T>(w.c3_norm_2.g + 4 * q);
        Vec4<T> hat = load4<T>(agg_tape + drow * FP + 4 * q);
        const T rstd = ln1_hat<LG>(hat, invn, nvalid);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dagg.v[k] = dz.v[k] * g2.v[k];
          B32.v[k] += dz.v[k];
          G32.v[k] += dz.v[k] * hat.v[k];
        }
        ln1_bwd<LG>(dagg, hat, rstd, invn, nvalid);
      }
      Vec4<T> dpf{{0, 0, 0, 0}}, dpc{{0, 0, 0, 0}};
      const int rb = d_rb[i], cnt = d_cnt[i], rskip = d_skip[i];
      // destinations of one atom walk the same source rows: start each lane group at a
      // different row so that their LDS atomics on dq do not meet at one address
      const int tstart = cnt > 0 ? (grp * 5) % cnt : 0;
      for (int tt = 0; tt < cnt; ++tt) {
        const int t = tt + tstart < cnt ? tt + tstart : tt + tstart - cnt;
        const int r = rb + t + ((rb + t >= rskip) ? 1 : 0);
        const T *qr = qrows + (size_t)r * 2 * FP + 4 * q;
        const Vec4<T> qf = load4<T>(qr), qc = load4<T>(qr + FP);
        T dot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) dot += pf.v[k] * qf.v[k] + pc.v[k] * qc.v[k];
        dot = lg_sum<LG>(dot);
        T var = (sp + sq[r] + (T)2 * dot) * inv2n;
        var = var > (T)0 ? var : (T)0;
        const T rstd = fast_rsq(var + (T)1e-5);
        Vec4<T> hf, hc, df, dc;
        T sa = 0, sb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hf.v[k] = (pf.v[k] + qf.v[k]) * rstd;
          hc.v[k] = (pc.v[k] + qc.v[k]) * rstd;
          T sg, th;
          gate_parts<T>(hf.v[k] * g1f.v[k] + b1f.v[k], hc.v[k] * g1c.v[k] + b1c.v[k], sg, th);
          const T dyf = dagg.v[k] * th * sg * ((T)1 - sg), dyc = dagg.v[k] * sg * ((T)1 - th * th);
          B31f.v[k] += dyf;
          G31f.v[k] += dyf * hf.v[k];
          B31c.v[k] += dyc;
          G31c.v[k] += dyc * hc.v[k];
          df.v[k] = dyf * g1f.v[k];
          dc.v[k] = dyc * g1c.v[k];
          sa += df.v[k] + dc.v[k];
          sb += df.v[k] * hf.v[k] + dc.v[k] * hc.v[k];
        }
        sa = lg_sum<LG>(sa) * inv2n;
        sb = lg_sum<LG>(sb) * inv2n;
        T *dqr = dq + (size_t)r * 2 * FP + 4 * q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const T xf = k < nvalid ? rstd * (df.v[k] - sa - hf.v[k] * sb) : (T)0;
          const T xc = k < nvalid ? rstd * (dc.v[k] - sa - hc.v[k] * sb) : (T)0;
          dpf.v[k] += xf;
          dpc.v[k] += xc;
          atomicAdd(dqr + k, xf);        // LDS: destinations of the same atom share source rows
          atomicAdd(dqr + FP + k, xc);
        }
      }
      store4(dpq + cdrow * (4 * FP) + 4 * q, dpf);
      store4(dpq + cdrow * (4 * FP) + FP + 4 * q, dpc);
      {
        T *dj = dnj + (size_t)d_bl[i] * 2 * FP + 4 * q;  // LDS: the atom belongs to this tile only
        T *dk = dnp3 + (cnrow0 + d_a[i]) * (6 * FP) + 4 * FP + 4 * q;
        atomic_add4(dj, dpf);
        atomic_add4(dj + FP, dpc);
        atomic_add4(dk, dpf);
        atomic_add4(dk + FP, dpc);
      }
      // c2 = LN(gate(LN(c2pre)))  (_gnn.py:223-228)
      {
        Vec4<T> xf = load4<T>(c2pre + drow * (2 * FP) + 4 * q), xc = load4<T>(c2pre + drow * (2 * FP) + FP + 4 * q);
        const T rstd1 = ln2_hat<LG>(xf, xc, inv2n, nvalid);
        const Vec4<T> gf = load4<T>(w.c2_norm_1.g + 4 * q), bf = load4<T>(w.c2_norm_1.b + 4 * q);
        const Vec4<T> gc = load4<T>(w.c2_norm_1.g + FP + 4 * q), bc = load4<T>(w.c2_norm_1.b + FP + 4 * q);
        Vec4<T> gv, da, db;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          gate_grad(xf.v[k] * gf.v[k] + bf.v[k], xc.v[k] * gc.v[k] + bc.v[k], gv.v[k], da.v[k], db.v[k]);
        Vec4<T> hat = gv;
        const T rstd2 = ln1_hat<LG>(hat, invn, nvalid);
        const Vec4<T> g22 = load4<T>(w.c2_norm_2.g + 4 * q);
        Vec4<T> dg;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dg.v[k] = dz.v[k] * g22.v[k];
          B22.v[k] += dz.v[k];
          G22.v[k] += dz.v[k] * hat.v[k];
        }
        ln1_bwd<LG>(dg, hat, rstd2, invn, nvalid);
        Vec4<T> df, dc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const T dyf = dg.v[k] * da.v[k], dyc = dg.v[k] * db.v[k];
          B21f.v[k] += dyf;
          G21f.v[k] += dyf * xf.v[k];
          B21c.v[k] += dyc;
          G21c.v[k] += dyc * xc.v[k];
          df.v[k] = dyf * gf.v[k];
          dc.v[k] = dyc * gc.v[k];
        }
        ln2_bwd<LG>(df, dc, xf, xc, rstd1, inv2n, nvalid);
        store4(dc2pre + cdrow * (2 * FP) + 4 * q, df);
        store4(dc2pre + cdrow * (2 * FP) + FP + 4 * q, dc);
      }
    }
    __syncthreads();
    // the tile's source-row cotangents leave the CU once (each row belongs to this tile only)
    for (int r = grp; r < rows; r += G) {
      T *o = dpq + (cerow0 + eo0 + r) * (4 * FP) + 2 * FP + 4 * q;
      store4(o, load4<T>(dq + (size_t)r * 2 * FP + 4 * q));
      store4(o + FP, load4<T>(dq + (size_t)r * 2 * FP + FP + 4 * q));
    }
    for (int ii = threadIdx.x; ii < (j1 - j0) * (2 * FP / 4); ii += 256) {
      const int n = ii / (2 * FP / 4), cc = (ii % (2 * FP / 4)) * 4;
      store4(dnp3 + (cnrow0 + j0 + n) * (6 * FP) + 2 * FP + cc, load4<T>(dnj + (size_t)n * 2 * FP + cc));
    }
    __syncthreads();
  }
  if (want_param_grads) {  // (uniform) qrows is free after the last frame: reduction scratch
    __syncthreads();
    wg_sum_atomic_add<LG>(qrows, G31f, const_cast<T *>(gw.c3_norm_1.g));
    wg_sum_atomic_add<LG>(qrows, B31f, const_cast<T *>(gw.c3_norm_1.b));
    wg_sum_atomic_add<LG>(qrows, G31c, const_cast<T *>(gw.c3_norm_1.g) + FP);
    wg_sum_atomic_add<LG>(qrows, B31c, const_cast<T *>(gw.c3_norm_1.b) + FP);
    wg_sum_atomic_add<LG>(qrows, G32, const_cast<T *>(gw.c3_norm_2.g));
    wg_sum_atomic_add<LG>(qrows, B32, const_cast<T *>(gw.c3_norm_2.b));
    wg_sum_atomic_add<LG>(qrows, G21f, const_cast<T *>(gw.c2_norm_1.g));
    wg_sum_atomic_add<LG>(qrows, B21f, const_cast<T *>(gw.c2_norm_1.b));
    wg_sum_atomic_add<LG>(qrows, G21c, const_cast<T *>(gw.c2_norm_1.g) + FP);
    wg_sum_atomic_add<LG>(qrows, B21c, const_cast<T *>(gw.c2_norm_1.b) + FP);
    wg_sum_atomic_add<LG>(qrows, G22, const_cast<T *>(gw.c2_norm_2.g));
    wg_sum_atomic_add<LG>(qrows, B22, const_cast<T *>(gw.c2_norm_2.b));
  }
}

// ---- wave-per-(atom, half of its destinations) variant --------------------------------------
// In edge_bwd_tile_kernel every (destination d, source row r) pair adds its 2Fe-wide term to
// dP'_d (registers) AND to dQ'_r (ds_add_f32): 8 LDS float atomics per pair and lane, and the
// LDS atomic unit, not the VALU (7 % busy), sets the pace (measured 0.9 ms per launch at batch 32).
// Round 2 evaluated the pair term twice instead (once per owner of d, once per owner of r: 0.70 ms).
// Here it is evaluated ONCE and without LDS atomics (measured: two ds_add_f32 per lane and pair cost
// more than the second evaluation did, profiles/r03/train_backward.txt): a wave owns one HALF of the
// destinations entering an atom, takes them 64/LG at a time (one per lane group) and walks the atom's
// source rows with all its lane groups in step.  dP'_d stays in the group's registers; the wave's
// contribution to dQ'_r is a sum ACROSS its lane groups (a reduce-scatter, 6 ds_bpermute at LG = 16,
// leaves every lane two of the 128 values), added with a plain read-modify-write to the wave's OWN copy
// of the dQ' rows (one copy per half), so no two waves ever touch the same LDS word.
// LDS: the centred Q' rows, two copies of the dQ' rows and the per-destination LayerNorm cotangents of
// the tile (one workgroup per CU at Fe = 64); the centred P' rows are parked in dpq's P' half, which
// the same wave overwrites with dP' when it is done with them.
#ifndef RN_BWD_PROBE
#define RN_BWD_PROBE 0  // timing experiments only (results are wrong): 1 no pair terms, 2 no reduce-scatter
#endif
template <int FP, typename T, int NT>
__global__ __launch_bounds__(NT) void edge_bwd_tile2_kernel(
    const T *__restrict__ pq, const T *__restrict__ np3, const T *__restrict__ c2pre,
    const T *__restrict__ edge_next, const T *__restrict__ agg_tape, const T *__restrict__ dedge_next,
    T *__restrict__ dedge_prev, T *__restrict__ dpq, T *__restrict__ dnp3, T *__restrict__ dc2pre,
    int C, int B, Graph g, Dims d, PassW<T> w, PassW<T> gw, int want_param_grads) {
  constexpr int LG = FP / 4;
  constexpr int G = NT / LG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // its own atom partition when the handle has one (Graph::bt_*), else the forward kernel's tiles
  const bool own = g.bt_num > 0;
  const int maxR = own ? g.bt_max_out_rows : g.max_tile_out_rows, maxD = own ? g.bt_max_in_rows : g.max_tile_in_rows,
            maxN = own ? g.bt_max_nodes : g.max_tile_nodes;
  const int num_tiles = own ? g.bt_num : g.num_tiles;
  const int *tile_begin = own ? g.bt_begin : g.tile_begin;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    unsigned char *p = smem_raw + off;
    off += (bytes + 15) & ~size_t(15);
    return p;
  };
  T *qrows = reinterpret_cast<T *>(carve((size_t)maxR * 2 * FP * sizeof(T)));   // centred Q' rows
  T *dqrows = reinterpret_cast<T *>(carve((size_t)2 * maxR * 2 * FP * sizeof(T)));  // dQ' rows, one copy per half
  if (off < (size_t)4 * NT * sizeof(T)) off = (size_t)4 * NT * sizeof(T);            // scratch of wg_sum_atomic_add
  T *dagg_s = reinterpret_cast<T *>(carve((size_t)maxD * FP * sizeof(T)));       // d(sum over triplets), per destination
  T *sq = reinterpret_cast<T *>(carve((size_t)maxR * sizeof(T)));
  T *sp_s = reinterpret_cast<T *>(carve((size_t)maxD * sizeof(T)));
  T *nj = reinterpret_cast<T *>(carve((size_t)maxN * 2 * FP * sizeof(T)));
  T *dnj = reinterpret_cast<T *>(carve((size_t)maxN * 2 * FP * sizeof(T)));
  int *qb = reinterpret_cast<int *>(carve((size_t)maxR * 4));
  int *qn = reinterpret_cast<int *>(carve((size_t)maxR * 4));                    // tile-local atom a_r of a source row
  int *dl = reinterpret_cast<int *>(carve((size_t)maxD * 6 * 4));
  int *d_edge = dl, *d_a = dl + maxD, *d_bl = dl + 2 * maxD, *d_rb = dl + 3 * maxD,
      *d_cnt = dl + 4 * maxD, *d_skip = dl + 5 * maxD;
  constexpr int GW = 64 / LG, NW = NT / 64;                                       // lane groups per wave, waves

  const int tile = blockIdx.x % num_tiles;
  const int cg = blockIdx.x / num_tiles, ncg = gridDim.x / num_tiles;
  const int j0 = tile_begin[tile], j1 = tile_begin[tile + 1];
  const int eo0 = g.out_ptr[j0], rows = g.out_ptr[j1] - eo0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
  for (int r = threadIdx.x; r < rows; r += NT) {
    qb[r] = g.edge_b[eo0 + r];
    qn[r] = g.edge_a[eo0 + r] - j0;
  }
  for (int i = threadIdx.x; i < dcount; i += NT) {
    const int dst = g.in_edge[di0 + i];
    const int ad = g.edge_a[dst], bd = g.edge_b[dst];
    const int rb = g.out_ptr[bd] - eo0, re = g.out_ptr[bd + 1] - eo0;
    const int rev = g.rev_edge[dst];
    d_edge[i] = dst;
    d_a[i] = ad;
    d_bl[i] = bd - j0;
    d_rb[i] = rb;
    d_cnt[i] = (re - rb) - (rev >= 0 ? 1 : 0);
    d_skip[i] = rev >= 0 ? rev - eo0 : re;
  }
  __syncthreads();

  const int grp = threadIdx.x / LG, q = threadIdx.x % LG;
  const int wave = threadIdx.x >> 6, wgrp = (threadIdx.x & 63) / LG;
  const int nvalid = min(max(d.Fe - 4 * q, 0), 4);
  const T inv2n = (T)1 / (T)(2 * d.Fe), invn = (T)1 / (T)d.Fe;
  const Vec4<T> g1f = load4<T>(w.c3_norm_1.g + 4 * q), b1f = load4<T>(w.c3_norm_1.b + 4 * q);
  const Vec4<T> g1c = load4<T>(w.c3_norm_1.g + FP + 4 * q), b1c = load4<T>(w.c3_norm_1.b + FP + 4 * q);
  Vec4<T> G31f{{0, 0, 0, 0}}, B31f = G31f, G31c = G31f, B31c = G31f, G32 = G31f, B32 = G31f;
  Vec4<T> G21f = G31f, B21f = G31f, G21c = G31f, B21c = G31f, G22 = G31f, B22 = G31f;

  // the term of pair (d, r): x = d(P'_d + Q'_r) through LayerNorm(c3_norm_1) and the gate; `stats` = also
  // accumulate the LayerNorm parameter gradients (a zero `dagg` makes the term and those sums vanish)
  auto pair_term = [&](const Vec4<T> &pf, const Vec4<T> &pc, T sp, const Vec4<T> &qf, const Vec4<T> &qc, T sqr,
                       const Vec4<T> &dagg, bool stats, Vec4<T> &xf, Vec4<T> &xc) {
    T dot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) dot += pf.v[k] * qf.v[k] + pc.v[k] * qc.v[k];
    dot = lg_sum<LG>(dot);
    T var = (sp + sqr + (T)2 * dot) * inv2n;
    var = var > (T)0 ? var : (T)0;
    const T rstd = fast_rsq(var + (T)1e-5);
    Vec4<T> hf, hc, df, dc;
    T sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hf.v[k] = (pf.v[k] + qf.v[k]) * rstd;
      hc.v[k] = (pc.v[k] + qc.v[k]) * rstd;
      T sg, th;
      gate_parts<T>(hf.v[k] * g1f.v[k] + b1f.v[k], hc.v[k] * g1c.v[k] + b1c.v[k], sg, th);
      const T dyf = dagg.v[k] * th * sg * ((T)1 - sg), dyc = dagg.v[k] * sg * ((T)1 - th * th);
      if (stats) {
        B31f.v[k] += dyf;
        G31f.v[k] += dyf * hf.v[k];
        B31c.v[k] += dyc;
        G31c.v[k] += dyc * hc.v[k];
      }
      df.v[k] = dyf * g1f.v[k];
      dc.v[k] = dyc * g1c.v[k];
      sa += df.v[k] + dc.v[k];
      sb += df.v[k] * hf.v[k] + dc.v[k] * hc.v[k];
    }
    sa = lg_sum<LG>(sa) * inv2n;
    sb = lg_sum<LG>(sb) * inv2n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xf.v[k] = k < nvalid ? rstd * (df.v[k] - sa - hf.v[k] * sb) : (T)0;
      xc.v[k] = k < nvalid ? rstd * (dc.v[k] - sa - hc.v[k] * sb) : (T)0;
    }
  };

  for (int c = cg; c < C; c += ncg) {
    const int s = c / B;
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
    const int64_t cerow0 = (int64_t)c * g.E, cnrow0 = (int64_t)c * g.N;
    for (int i = threadIdx.x; i < (j1 - j0) * (2 * FP / 4); i += NT) {
      const int n = i / (2 * FP / 4), cc = (i % (2 * FP / 4)) * 4;
      store4(nj + (size_t)n * 2 * FP + cc, load4<T>(np3 + (nrow0 + j0 + n) * (6 * FP) + 2 * FP + cc));
      store4(dnj + (size_t)n * 2 * FP + cc, Vec4<T>{{0, 0, 0, 0}});
    }
    for (int i = threadIdx.x; i < 2 * maxR * (2 * FP / 4); i += NT) store4(dqrows + (size_t)i * 4, Vec4<T>{{0, 0, 0, 0}});
    for (int r = grp; r < rows; r += G) {  // centred source rows, as in the forward kernel
      const T *qp = pq + (erow0 + eo0 + r) * (4 * FP) + 2 * FP + 4 * q;
      const T *np = np3 + (nrow0 + qb[r]) * (6 * FP) + 4 * q;
      Vec4<T> f = load4<T>(qp), cc = load4<T>(qp + FP);
      const Vec4<T> nf = load4<T>(np), nc = load4<T>(np + FP);
      T sum = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f.v[k] += nf.v[k];
        cc.v[k] += nc.v[k];
        sum += f.v[k] + cc.v[k];
      }
      const T mean = lg_sum<LG>(sum) * inv2n;
      T ss = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f.v[k] = k < nvalid ? f.v[k] - mean : (T)0;
        cc.v[k] = k < nvalid ? cc.v[k] - mean : (T)0;
        ss += f.v[k] * f.v[k] + cc.v[k] * cc.v[k];
      }
      ss = lg_sum<LG>(ss);
      store4(qrows + (size_t)r * 2 * FP + 4 * q, f);
      store4(qrows + (size_t)r * 2 * FP + FP + 4 * q, cc);
      if (q == 0) sq[r] = ss;
    }
    __syncthreads();  // nj complete

    // ---- per destination: residual-tanh and LayerNorm(c3_norm_2) backward, centred P' -> LDS; c2 branch
    for (int i = grp; i < dcount; i += G) {
      const int dst = d_edge[i];
      const int64_t drow = erow0 + dst, cdrow = cerow0 + dst;
      const Vec4<T> e1 = load4<T>(edge_next + drow * FP + 4 * q);
      Vec4<T> dz = load4<T>(dedge_next + cdrow * FP + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) dz.v[k] *= ((T)1 - e1.v[k] * e1.v[k]);
      store4(dedge_prev + cdrow * FP + 4 * q, dz);
      Vec4<T> pf = load4<T>(pq + drow * (4 * FP) + 4 * q), pc = load4<T>(pq + drow * (4 * FP) + FP + 4 * q);
      {
        const T *nk = np3 + (nrow0 + d_a[i]) * (6 * FP) + 4 * FP + 4 * q;
        const Vec4<T> kf = load4<T>(nk), kc = load4<T>(nk + FP);
        const Vec4<T> jf = load4<T>(nj + (size_t)d_bl[i] * 2 * FP + 4 * q);
        const Vec4<T> jc = load4<T>(nj + (size_t)d_bl[i] * 2 * FP + FP + 4 * q);
        T sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pf.v[k] += jf.v[k] + kf.v[k];
          pc.v[k] += jc.v[k] + kc.v[k];
          sum += pf.v[k] + pc.v[k];
        }
        const T mean = lg_sum<LG>(sum) * inv2n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pf.v[k] = k < nvalid ? pf.v[k] - mean : (T)0;
          pc.v[k] = k < nvalid ? pc.v[k] - mean : (T)0;
        }
      }
      T sp = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) sp += pf.v[k] * pf.v[k] + pc.v[k] * pc.v[k];
      sp = lg_sum<LG>(sp);
      store4(dpq + cdrow * (4 * FP) + 4 * q, pf);  // parked: read back and replaced by dP' in the pair pass
      store4(dpq + cdrow * (4 * FP) + FP + 4 * q, pc);
      if (q == 0) sp_s[i] = sp;
      {
        Vec4<T> dagg;
        const Vec4<T> g2 = load4<T>(w.c3_norm_2.g + 4 * q);
        Vec4<T> hat = load4<T>(agg_tape + drow * FP + 4 * q);
        const T rstd = ln1_hat<LG>(hat, invn, nvalid);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dagg.v[k] = dz.v[k] * g2.v[k];
          B32.v[k] += dz.v[k];
          G32.v[k] += dz.v[k] * hat.v[k];
        }
        ln1_bwd<LG>(dagg, hat, rstd, invn, nvalid);
        store4(dagg_s + (size_t)i * FP + 4 * q, dagg);
      }
      // c2 = LN(gate(LN(c2pre)))  (_gnn.py:223-228)
      {
        Vec4<T> xf = load4<T>(c2pre + drow * (2 * FP) + 4 * q), xc = load4<T>(c2pre + drow * (2 * FP) + FP + 4 * q);
        const T rstd1 = ln2_hat<LG>(xf, xc, inv2n, nvalid);
        const Vec4<T> gf = load4<T>(w.c2_norm_1.g + 4 * q), bf = load4<T>(w.c2_norm_1.b + 4 * q);
        const Vec4<T> gc = load4<T>(w.c2_norm_1.g + FP + 4 * q), bc = load4<T>(w.c2_norm_1.b + FP + 4 * q);
        Vec4<T> gv, da, db;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          gate_grad(xf.v[k] * gf.v[k] + bf.v[k], xc.v[k] * gc.v[k] + bc.v[k], gv.v[k], da.v[k], db.v[k]);
        Vec4<T> hat = gv;
        const T rstd2 = ln1_hat<LG>(hat, invn, nvalid);
        const Vec4<T> g22 = load4<T>(w.c2_norm_2.g + 4 * q);
        Vec4<T> dg;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dg.v[k] = dz.v[k] * g22.v[k];
          B22.v[k] += dz.v[k];
          G22.v[k] += dz.v[k] * hat.v[k];
        }
        ln1_bwd<LG>(dg, hat, rstd2, invn, nvalid);
        Vec4<T> df, dc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const T dyf = dg.v[k] * da.v[k], dyc = dg.v[k] * db.v[k];
          B21f.v[k] += dyf;
          G21f.v[k] += dyf * xf.v[k];
          B21c.v[k] += dyc;
          G21c.v[k] += dyc * xc.v[k];
          df.v[k] = dyf * gf.v[k];
          dc.v[k] = dyc * gc.v[k];
        }
        ln2_bwd<LG>(df, dc, xf, xc, rstd1, inv2n, nvalid);
        store4(dc2pre + cdrow * (2 * FP) + 4 * q, df);
        store4(dc2pre + cdrow * (2 * FP) + FP + 4 * q, dc);
      }
    }
    __syncthreads();  // qrows, dagg, the parked P' rows complete

    // ---- pair terms, once each: wave <- (atom n, half of the destinations entering it); GW destinations at a
    // time, all lane groups walk the atom's source rows together
    for (int item = wave; item < ((RN_BWD_PROBE & 1) ? 0 : 2 * (j1 - j0)); item += NW) {
      const int natoms = j1 - j0;
      const int half = item / natoms, n = item - half * natoms;  // (halves 0 of all atoms first: the longer ones)
      const int in0 = g.in_ptr[j0 + n] - di0, in1 = g.in_ptr[j0 + n + 1] - di0;
      const int nblk = (in1 - in0 + GW - 1) / GW, cut = (nblk + 1) / 2;
      const int r0 = g.out_ptr[j0 + n] - eo0, r1 = g.out_ptr[j0 + n + 1] - eo0;
      T *dqh = dqrows + (size_t)half * maxR * 2 * FP;
      for (int blk = half ? cut : 0; blk < (half ? nblk : cut); ++blk) {
        const int i = in0 + blk * GW + wgrp;
        const bool valid = i < in1;
        const int ic = valid ? i : in0;
        const int64_t cdrow = cerow0 + d_edge[ic];
        Vec4<T> pf = load4<T>(dpq + cdrow * (4 * FP) + 4 * q), pc = load4<T>(dpq + cdrow * (4 * FP) + FP + 4 * q);
        Vec4<T> dagg = load4<T>(dagg_s + (size_t)ic * FP + 4 * q);
        if (!valid) pf = pc = dagg = Vec4<T>{{0, 0, 0, 0}};  // (an idle lane group: its terms vanish)
        const T sp = sp_s[ic];
        const int rskip = d_skip[ic];
        Vec4<T> dpf{{0, 0, 0, 0}}, dpc{{0, 0, 0, 0}};
        for (int r = r0; r < r1; ++r) {
          const T *qr = qrows + (size_t)r * 2 * FP + 4 * q;
          Vec4<T> dg = dagg;
          if (r == rskip) dg = Vec4<T>{{0, 0, 0, 0}};  // r is the reverse of this destination: that triplet does not exist
          Vec4<T> xf, xc;
          pair_term(pf, pc, sp, load4<T>(qr), load4<T>(qr + FP), sq[r], dg, true, xf, xc);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            dpf.v[k] += xf.v[k];
            dpc.v[k] += xc.v[k];
          }
          T *dq = dqh + (size_t)r * 2 * FP + 4 * q;
          if constexpr ((RN_BWD_PROBE & 2) != 0) {
            dpf.v[0] += xc.v[0];
          } else if constexpr (GW == 4 && sizeof(T) == 4) {
            // reduce-scatter over the four lane groups on the VALU: v_permlane32_swap exchanges the upper half of one
            // register with the lower half of another, so (xc, xf) -> lower lanes hold both halves' xc, upper lanes both
            // halves' xf; v_permlane16_swap does the same between the 16-lane rows for the column pairs {0,1} / {2,3}
            const bool upper = (threadIdx.x & 32) != 0, odd = (threadIdx.x & 16) != 0;
            float a4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xc.v[k]), __float_as_uint(xf.v[k]), false, false);
              a4[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            float b2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a4[k]), __float_as_uint(a4[2 + k]), false, false);
              b2[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            T *o = dq + (upper ? 0 : FP) + (odd ? 2 : 0);  // this wave's copy: a plain read-modify-write
            o[0] += b2[0];
            o[1] += b2[1];
          } else if constexpr (GW == 4) {
            // (float64) reduce-scatter over the four lane groups: halves swap xf/xc, then quarters swap column pairs
            const bool upper = (threadIdx.x & 32) != 0, odd = (threadIdx.x & 16) != 0;
            T a4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const T send = upper ? xf.v[k] : xc.v[k], keep = upper ? xc.v[k] : xf.v[k];
              a4[k] = keep + __shfl_xor(send, 32);
            }
            T b2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const T send = odd ? a4[k] : a4[2 + k], keep = odd ? a4[2 + k] : a4[k];
              b2[k] = keep + __shfl_xor(send, 16);
            }
            T *o = dq + (upper ? FP : 0) + (odd ? 2 : 0);  // this wave's copy: a plain read-modify-write
            o[0] += b2[0];
            o[1] += b2[1];
          } else {
            // all-reduce over the wave's lane groups, then lane group 0 adds the row
#pragma unroll
            for (int sft = LG; sft < 64; sft <<= 1)
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                xf.v[k] += __shfl_xor(xf.v[k], sft);
                xc.v[k] += __shfl_xor(xc.v[k], sft);
              }
            if (wgrp == 0) {
              Vec4<T> of = load4<T>(dq), oc = load4<T>(dq + FP);
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                of.v[k] += xf.v[k];
                oc.v[k] += xc.v[k];
              }
              store4(dq, of);
              store4(dq + FP, oc);
            }
          }
        }
        if (valid) {
          store4(dpq + cdrow * (4 * FP) + 4 * q, dpf);
          store4(dpq + cdrow * (4 * FP) + FP + 4 * q, dpc);
          T *dj = dnj + (size_t)d_bl[i] * 2 * FP + 4 * q;  // LDS: the atom belongs to this tile only
          T *dk = dnp3 + (cnrow0 + d_a[i]) * (6 * FP) + 4 * FP + 4 * q;
          atomic_add4(dj, dpf);
          atomic_add4(dj + FP, dpc);
          atomic_add4(dk, dpf);
          atomic_add4(dk + FP, dpc);
        }
      }
    }
    __syncthreads();  // dQ' rows and dnj complete
    for (int i = threadIdx.x; i < rows * (2 * FP / 4); i += NT) {
      const int r = i / (2 * FP / 4), cc = (i % (2 * FP / 4)) * 4;
      Vec4<T> v = load4<T>(dqrows + (size_t)r * 2 * FP + cc);
      const Vec4<T> v1 = load4<T>(dqrows + ((size_t)maxR + r) * 2 * FP + cc);
#pragma unroll
      for (int k = 0; k < 4; ++k) v.v[k] += v1.v[k];
      store4(dpq + (cerow0 + eo0 + r) * (4 * FP) + 2 * FP + cc, v);
    }
    for (int ii = threadIdx.x; ii < (j1 - j0) * (2 * FP / 4); ii += NT) {
      const int n = ii / (2 * FP / 4), cc = (ii % (2 * FP / 4)) * 4;
      store4(dnp3 + (cnrow0 + j0 + n) * (6 * FP) + 2 * FP + cc, load4<T>(dnj + (size_t)n * 2 * FP + cc));
    }
    __syncthreads();
  }
  if (want_param_grads) {  // (uniform) qrows is free after the last frame: reduction scratch
    wg_sum_atomic_add<LG, T, NT>(qrows, G31f, const_cast<T *>(gw.c3_norm_1.g));
    wg_sum_atomic_add<LG, T, NT>(qrows, B31f, const_cast<T *>(gw.c3_norm_1.b));
    wg_sum_atomic_add<LG, T, NT>(qrows, G31c, const_cast<T *>(gw.c3_norm_1.g) + FP);
    wg_sum_atomic_add<LG, T, NT>(qrows, B31c, const_cast<T *>(gw.c3_norm_1.b) + FP);
    wg_sum_atomic_add<LG, T, NT>(qrows, G32, const_cast<T *>(gw.c3_norm_2.g));
    wg_sum_atomic_add<LG, T, NT>(qrows, B32, const_cast<T *>(gw.c3_norm_2.b));
    wg_sum_atomic_add<LG, T, NT>(qrows, G21f, const_cast<T *>(gw.c2_norm_1.g));
    wg_sum_atomic_add<LG, T, NT>(qrows, B21f, const_cast<T *>(gw.c2_norm_1.b));
    wg_sum_atomic_add<LG, T, NT>(qrows, G21c, const_cast<T *>(gw.c2_norm_1.g) + FP);
    wg_sum_atomic_add<LG, T, NT>(qrows, B21c, const_cast<T *>(gw.c2_norm_1.b) + FP);
    wg_sum_atomic_add<LG, T, NT>(qrows, G22, const_cast<T *>(gw.c2_norm_2.g));
    wg_sum_atomic_add<LG, T, NT>(qrows, B22, const_cast<T *>(gw.c2_norm_2.b));
  }
}

size_t edge_bwd_tile2_lds_bytes(int rows, int in_rows, int nodes, int FP, size_t elem) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  // (qrows and the two dQ' copies; qrows doubles as the scratch of wg_sum_atomic_add)
  const size_t row_arrays = std::max(up((size_t)rows * 2 * FP * elem) + up((size_t)2 * rows * 2 * FP * elem), (size_t)2048 * elem);
  return row_arrays + up((size_t)in_rows * FP * elem) + up((size_t)rows * elem) + up((size_t)in_rows * elem) +
         2 * up((size_t)nodes * 2 * FP * elem) + 2 * up((size_t)rows * 4) + up((size_t)in_rows * 6 * 4);
}
static size_t edge_bwd_tile2_lds(const Graph &g, int FP, size_t elem) {
  if (g.bt_num > 0) return edge_bwd_tile2_lds_bytes(g.bt_max_out_rows, g.bt_max_in_rows, g.bt_max_nodes, FP, elem);
  return edge_bwd_tile2_lds_bytes(g.max_tile_out_rows, g.max_tile_in_rows, g.max_tile_nodes, FP, elem);
}

static size_t edge_bwd_tile_lds(const Graph &g, int FP, size_t elem) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  // (the two row arrays double as the 1024-element scratch of wg_sum_atomic_add)
  const size_t row_arrays = std::max(2 * up((size_t)g.max_tile_out_rows * 2 * FP * elem), (size_t)1024 * elem);
  return row_arrays + up((size_t)g.max_tile_out_rows * elem) +
         2 * up((size_t)g.max_tile_nodes * 2 * FP * elem) + up((size_t)g.max_tile_out_rows * 4) +
         up((size_t)g.max_tile_in_rows * 6 * 4);
}

// dnp3[c, n][Wi block] = sum over the edges e entering atom n of dQ'_e  (the node part of the
// source-row projection).  Gather over the in-edge list, one lane group per (instance, atom):
// a single writer per row, no atomics (the scatter form spent 150 us per call on them).
template <int LG, typename T>
__global__ void q_gather_kernel(const T *__restrict__ dpq, T *__restrict__ dnp3, int C, Graph g) {
  constexpr int FP = LG * 4;
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LG;
  const int q = threadIdx.x % LG;
  if (gid >= (int64_t)C * g.N) return;
  const int c = (int)(gid / g.N), n = (int)(gid % g.N);
  Vec4<T> sf{{0, 0, 0, 0}}, sc{{0, 0, 0, 0}};
  const int i0 = g.in_ptr[n], i1 = g.in_ptr[n + 1];
  int i = i0;
  for (; i + 3 < i1; i += 4) {  // four rows in flight (the sum keeps the ascending order)
    const T *s0 = dpq + ((int64_t)c * g.E + g.in_edge[i]) * (4 * FP) + 2 * FP + 4 * q;
    const T *s1 = dpq + ((int64_t)c * g.E + g.in_edge[i + 1]) * (4 * FP) + 2 * FP + 4 * q;
    const T *s2 = dpq + ((int64_t)c * g.E + g.in_edge[i + 2]) * (4 * FP) + 2 * FP + 4 * q;
    const T *s3 = dpq + ((int64_t)c * g.E + g.in_edge[i + 3]) * (4 * FP) + 2 * FP + 4 * q;
    const Vec4<T> f0 = load4<T>(s0), c0 = load4<T>(s0 + FP), f1 = load4<T>(s1), c1 = load4<T>(s1 + FP);
    const Vec4<T> f2 = load4<T>(s2), c2 = load4<T>(s2 + FP), f3 = load4<T>(s3), c3 = load4<T>(s3 + FP);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sf.v[k] = (((sf.v[k] + f0.v[k]) + f1.v[k]) + f2.v[k]) + f3.v[k];
      sc.v[k] = (((sc.v[k] + c0.v[k]) + c1.v[k]) + c2.v[k]) + c3.v[k];
    }
  }
  for (; i < i1; ++i) {
    const T *src = dpq + ((int64_t)c * g.E + g.in_edge[i]) * (4 * FP) + 2 * FP + 4 * q;
    const Vec4<T> f = load4<T>(src), cc = load4<T>(src + FP);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sf.v[k] += f.v[k];
      sc.v[k] += cc.v[k];
    }
  }
  T *dst = dnp3 + gid * (6 * FP) + 4 * q;
  store4(dst, sf);
  store4(dst + FP, sc);
}

// d(node[b]*node[a]) -> dnode   (operand of c2_linear): atom n collects dprod_e * node[a_e] from
// the edges entering it and dprod_e * node[b_e] from the edges leaving it (contiguous range);
// one lane group per (instance, atom), single writer, no atomics.
template <int LG, typename T>
__global__ void prod_bwd_kernel(const T *__restrict__ dprod, const T *__restrict__ node,
                                T *__restrict__ dnode, int C, int B, Graph g) {
  constexpr int FP = LG * 4;
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LG;
  const int q = threadIdx.x % LG;
  if (gid >= (int64_t)C * g.N) return;
  const int c = (int)(gid / g.N), n = (int)(gid % g.N);
  const int64_t nrow0 = (int64_t)(c / B) * g.N, cerow0 = (int64_t)c * g.E;
  Vec4<T> acc = load4<T>(dnode + gid * FP + 4 * q);
  {  // (two edges in flight per step, summed in ascending order)
    const int i0 = g.in_ptr[n], i1 = g.in_ptr[n + 1];
    int i = i0;
    for (; i + 1 < i1; i += 2) {
      const int e0 = g.in_edge[i], e1 = g.in_edge[i + 1];
      const Vec4<T> dp0 = load4<T>(dprod + (cerow0 + e0) * FP + 4 * q), dp1 = load4<T>(dprod + (cerow0 + e1) * FP + 4 * q);
      const Vec4<T> na0 = load4<T>(node + (nrow0 + g.edge_a[e0]) * FP + 4 * q), na1 = load4<T>(node + (nrow0 + g.edge_a[e1]) * FP + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc.v[k] = (acc.v[k] + dp0.v[k] * na0.v[k]) + dp1.v[k] * na1.v[k];
    }
    if (i < i1) {
      const int e = g.in_edge[i];
      const Vec4<T> dp = load4<T>(dprod + (cerow0 + e) * FP + 4 * q);
      const Vec4<T> na = load4<T>(node + (nrow0 + g.edge_a[e]) * FP + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc.v[k] += dp.v[k] * na.v[k];
    }
    const int o1 = g.out_ptr[n + 1];
    int e = g.out_ptr[n];
    for (; e + 1 < o1; e += 2) {
      const Vec4<T> dp0 = load4<T>(dprod + (cerow0 + e) * FP + 4 * q), dp1 = load4<T>(dprod + (cerow0 + e + 1) * FP + 4 * q);
      const Vec4<T> nb0 = load4<T>(node + (nrow0 + g.edge_b[e]) * FP + 4 * q), nb1 = load4<T>(node + (nrow0 + g.edge_b[e + 1]) * FP + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc.v[k] = (acc.v[k] + dp0.v[k] * nb0.v[k]) + dp1.v[k] * nb1.v[k];
    }
    if (e < o1) {
      const Vec4<T> dp = load4<T>(dprod + (cerow0 + e) * FP + 4 * q);
      const Vec4<T> nb = load4<T>(node + (nrow0 + g.edge_b[e]) * FP + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc.v[k] += dp.v[k] * nb.v[k];
    }
  }
  store4(dnode + gid * FP + 4 * q, acc);
}

// prod[c-frame rows][FnP] = node[b_e] * node[a_e]: the operand of c2_linear written out once, so that the
// weight-gradient product reads plain rows (the gathering form of gemm_tn_mfma_kernel spent 3x the
// time of the plain one on its dependent index loads).
template <int LG, typename T>
__global__ void prod_fwd_kernel(const T *__restrict__ node, T *__restrict__ prod, int64_t rows, Graph g) {
  constexpr int FP = LG * 4;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LG;
  const int q = threadIdx.x % LG;
  if (row >= rows) return;
  const int64_t s = row / g.E;
  const int e = (int)(row - s * g.E);
  const Vec4<T> nb = load4<T>(node + (s * g.N + g.edge_b[e]) * FP + 4 * q);
  const Vec4<T> na = load4<T>(node + (s * g.N + g.edge_a[e]) * FP + 4 * q);
  Vec4<T> o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o.v[k] = nb.v[k] * na.v[k];
  store4(prod + row * FP + 4 * q, o);
}

// =========================================================================== node block
// Reverse of _NodeBlock.forward (_gnn.py:141-151) for one atom b per lane group.
template <int LG, typename T>
__global__ __launch_bounds__(256) void node_bwd_kernel(
    const T *__restrict__ npc1, const T *__restrict__ bc1, const T *__restrict__ node_next,
    const T *__restrict__ dnode_next, T *__restrict__ dnode_prev, T *__restrict__ dbc1,
    T *__restrict__ dnpc1, int C, int B, Graph g, Dims d, PassW<T> w, PassW<T> gw,
    int want_param_grads) {
  constexpr int FP = LG * 4;
  __shared__ __attribute__((aligned(16))) T scratch[1024];
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LG;
  const int q = threadIdx.x % LG;
  Vec4<T> G1f{{0, 0, 0, 0}}, B1f = G1f, G1c = G1f, B1c = G1f, G2 = G1f, B2 = G1f;
  if (gid < (int64_t)C * g.N) {
  const int c = (int)(gid / g.N), b = (int)(gid % g.N);
  const int s = c / B;
  const int64_t frow = (int64_t)s * g.N + b;
  const int nvalid = min(max(d.Fn - 4 * q, 0), 4);
  const T inv2n = (T)1 / (T)(2 * d.Fn), invn = (T)1 / (T)d.Fn;
  const Vec4<T> n1 = load4<T>(node_next + frow * FP + 4 * q);
  Vec4<T> dz = load4<T>(dnode_next + gid * FP + 4 * q);
#pragma unroll
  for (int k = 0; k < 4; ++k) dz.v[k] *= ((T)1 - n1.v[k] * n1.v[k]);
  store4(dnode_prev + gid * FP + 4 * q, dz);

  const Vec4<T> af = load4<T>(npc1 + frow * (2 * FP) + 4 * q), ac = load4<T>(npc1 + frow * (2 * FP) + FP + 4 * q);
  const Vec4<T> gf = load4<T>(w.c1_norm.g + 4 * q), bf = load4<T>(w.c1_norm.b + 4 * q);
  const Vec4<T> gc = load4<T>(w.c1_norm.g + FP + 4 * q), bc = load4<T>(w.c1_norm.b + FP + 4 * q);
  const int beg = g.in_ptr[b], end = g.in_ptr[b + 1];
  auto row = [&](int e, Vec4<T> &xf, Vec4<T> &xc) {
    const T *p = bc1 + ((int64_t)s * g.E + e) * (2 * FP) + 4 * q;
    xf = load4<T>(p);
    xc = load4<T>(p + FP);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xf.v[k] += af.v[k];
      xc.v[k] += ac.v[k];
    }
  };
  Vec4<T> agg{{0, 0, 0, 0}};
  for (int i = beg; i < end; ++i) {
    Vec4<T> xf, xc;
    row(g.in_edge[i], xf, xc);
    ln2_hat<LG>(xf, xc, inv2n, nvalid);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T gv, a, bb;
      gate_grad(xf.v[k] * gf.v[k] + bf.v[k], xc.v[k] * gc.v[k] + bc.v[k], gv, a, bb);
      agg.v[k] += gv;
    }
  }
  Vec4<T> dagg;
  {
    const Vec4<T> g2 = load4<T>(w.final_norm.g + 4 * q);
    Vec4<T> hat = agg;
    const T rstd = ln1_hat<LG>(hat, invn, nvalid);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dagg.v[k] = dz.v[k] * g2.v[k];
      B2.v[k] += dz.v[k];
      G2.v[k] += dz.v[k] * hat.v[k];
    }
    ln1_bwd<LG>(dagg, hat, rstd, invn, nvalid);
  }
  Vec4<T> sf{{0, 0, 0, 0}}, sc{{0, 0, 0, 0}};
  for (int i = beg; i < end; ++i) {
    const int e = g.in_edge[i];
    Vec4<T> xf, xc;
    row(e, xf, xc);
    const T rstd = ln2_hat<LG>(xf, xc, inv2n, nvalid);
    Vec4<T> df, dc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      T gv, a, bb;
      gate_grad(xf.v[k] * gf.v[k] + bf.v[k], xc.v[k] * gc.v[k] + bc.v[k], gv, a, bb);
      const T dyf = dagg.v[k] * a, dyc = dagg.v[k] * bb;
      B1f.v[k] += dyf;
      G1f.v[k] += dyf * xf.v[k];
      B1c.v[k] += dyc;
      G1c.v[k] += dyc * xc.v[k];
      df.v[k] = dyf * gf.v[k];
      dc.v[k] = dyc * gc.v[k];
    }
    ln2_bwd<LG>(df, dc, xf, xc, rstd, inv2n, nvalid);
    T *o = dbc1 + ((int64_t)c * g.E + e) * (2 * FP) + 4 * q;  // every edge has exactly one b
    store4(o, df);
    store4(o + FP, dc);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sf.v[k] += df.v[k];
      sc.v[k] += dc.v[k];
    }
  }
  store4(dnpc1 + gid * (2 * FP) + 4 * q, sf);
  store4(dnpc1 + gid * (2 * FP) + FP + 4 * q, sc);
  }
  if (want_param_grads) {  // (uniform) lane groups past the end contribute zeros
    wg_sum_atomic_add<LG>(scratch, G1f, const_cast<T *>(gw.c1_norm.g));
    wg_sum_atomic_add<LG>(scratch, B1f, const_cast<T *>(gw.c1_norm.b));
    wg_sum_atomic_add<LG>(scratch, G1c, const_cast<T *>(gw.c1_norm.g) + FP);
    wg_sum_atomic_add<LG>(scratch, B1c, const_cast<T *>(gw.c1_norm.b) + FP);
    wg_sum_atomic_add<LG>(scratch, G2, const_cast<T *>(gw.final_norm.g));
    wg_sum_atomic_add<LG>(scratch, B2, const_cast<T *>(gw.final_norm.b));
  }
}

// =========================================================================== geometry
// RBF + geometry backward: d(edge0), d(unit) -> d(fractional positions), accumulated
// atomically as float64 [C, N, 3].  (The minimum-image wrap is locally the identity.)
template <typename T>
__global__ void geom_bwd_kernel(const T *__restrict__ dedge0, const T *__restrict__ dunit,
                                const T *__restrict__ unit4, const T *__restrict__ lat,
                                const T *__restrict__ offs, T coef, int C, int B, Graph g, Dims d,
                                double *__restrict__ dpos) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)C * g.E) return;
  const int c = (int)(idx / g.E), e = (int)(idx % g.E);
  const int64_t frow = (int64_t)(c / B) * g.E + e;
  const T ux = unit4[frow * 4], uy = unit4[frow * 4 + 1], uz = unit4[frow * 4 + 2], dist = unit4[frow * 4 + 3];
  T ddist = 0;
  const T *de = dedge0 + idx * d.FeP;
  for (int f = 0; f < d.Fe; ++f) {
    const T x = dist - offs[f];
    ddist += de[f] * exp(coef * x * x) * ((T)2 * coef * x);
  }
  const T dux = dunit[idx * 4], duy = dunit[idx * 4 + 1], duz = dunit[idx * 4 + 2];
  const T proj = dux * ux + duy * uy + duz * uz;
  const T dc[3] = {ddist * ux + (dux - proj * ux) / dist, ddist * uy + (duy - proj * uy) / dist,
                   ddist * uz + (duz - proj * uz) / dist};
  const int a = g.edge_a[e], b = g.edge_b[e];
#pragma unroll
  for (int i = 0; i < 3; ++i) {  // cart_k = sum_i frac_i L[i][k]
    const T df = lat[3 * i] * dc[0] + lat[3 * i + 1] * dc[1] + lat[3 * i + 2] * dc[2];
    atomicAdd(dpos + ((int64_t)c * g.N + b) * 3 + i, (double)df);
    atomicAdd(dpos + ((int64_t)c * g.N + a) * 3 + i, -(double)df);
  }
}
template <typename T>
void launch_geom_bwd(const T *dedge0, const T *dunit, const T *unit4, const T *lat, const T *offs,
                     T coef, int C, int B, const Graph &g, Dims d, double *dpos, hipStream_t st) {
  const int64_t total = (int64_t)C * g.E;
  if (total == 0) return;
  geom_bwd_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(dedge0, dunit, unit4, lat, offs,
                                                                      coef, C, B, g, d, dpos);
}
template void launch_geom_bwd<float>(const float *, const float *, const float *, const float *,
                                     const float *, float, int, int, const Graph &, Dims, double *,
                                     hipStream_t);
template void launch_geom_bwd<double>(const double *, const double *, const double *, const double *,
                                      const double *, double, int, int, const Graph &, Dims,
                                      double *, hipStream_t);

// =========================================================================== launchers
#define RN_LG_SWITCH(FP, CALL)   \
  switch ((FP) / 4) {            \
    case 4: CALL(4); break;      \
    case 8: CALL(8); break;      \
    case 16: CALL(16); break;    \
    case 32: CALL(32); break;    \
  }

static bool edge_bwd_two_pass(const Graph &g, int FP, size_t elem) {
  static const bool no_two_pass = getenv("RN_POTGNN_BWD_ATOMIC") && atoi(getenv("RN_POTGNN_BWD_ATOMIC")) != 0;
  static const bool simple = getenv("RN_POTGNN_BWD_SIMPLE") && atoi(getenv("RN_POTGNN_BWD_SIMPLE")) != 0;
  if (FP != 16 && FP != 32 && FP != 64 && FP != 128) return false;
  return !simple && !no_two_pass && edge_bwd_tile2_lds(g, FP, elem) <= 160 * 1024 - 512;
}

template <int FP, typename T>
static bool launch_edge_bwd_tile(const T *pq, const T *np3, const T *c2pre, const T *edge_next,
                                 const T *agg, const T *dedge_next, T *dedge_prev, T *dpq, T *dnp3,
                                 T *dc2pre, int C, int B, const Graph &g, Dims d, const PassW<T> &w,
                                 const PassW<T> &gwv, int want, hipStream_t st) {
  const size_t lds2 = edge_bwd_tile2_lds(g, FP, sizeof(T));
  const bool two_pass = edge_bwd_two_pass(g, FP, sizeof(T));
  const size_t lds = two_pass ? lds2 : edge_bwd_tile_lds(g, FP, sizeof(T));
  if (lds > 160 * 1024 - 512) return false;
  // with its own small tiles: two 256-thread workgroups per CU (independent workgroups fill each other's latency-bound
  // phases); on the forward kernel's tiles: one 512-thread workgroup per CU
  if (two_pass && g.bt_num > 0) {
    constexpr int NTS = 256;
    auto k2 = &edge_bwd_tile2_kernel<FP, T, NTS>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu2 = 0, dev2 = 0, cus2 = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, k2, NTS, lds) != hipSuccess || per_cu2 < 1) per_cu2 = 1;
    hipDeviceProp_t prop2;
    if (hipGetDevice(&dev2) == hipSuccess && hipGetDeviceProperties(&prop2, dev2) == hipSuccess) cus2 = prop2.multiProcessorCount;
    int ncg2 = per_cu2 * cus2 / g.bt_num;
    ncg2 = ncg2 < 1 ? 1 : (ncg2 > C ? C : ncg2);
    edge_bwd_tile2_kernel<FP, T, NTS><<<(unsigned)ncg2 * (unsigned)g.bt_num, NTS, lds, st>>>(pq, np3, c2pre, edge_next, agg, dedge_next,
                                                                                    dedge_prev, dpq, dnp3, dc2pre, C, B, g, d, w, gwv, want);
    return true;
  }
  constexpr int NT2 = 512;
  const void *kern = two_pass ? reinterpret_cast<const void *>(&edge_bwd_tile2_kernel<FP, T, NT2>)
                              : reinterpret_cast<const void *>(&edge_bwd_tile_kernel<FP, T>);
  const int threads = two_pass ? NT2 : 256;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu < 1)
    per_cu = 1;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cus = prop.multiProcessorCount;
  int ncg = per_cu * cus / g.num_tiles;
  ncg = ncg < 1 ? 1 : (ncg > C ? C : ncg);
  const unsigned grid = (unsigned)ncg * (unsigned)g.num_tiles;
  if (two_pass)
    edge_bwd_tile2_kernel<FP, T, NT2><<<grid, NT2, lds, st>>>(pq, np3, c2pre, edge_next, agg, dedge_next,
                                                              dedge_prev, dpq, dnp3, dc2pre, C, B, g, d, w, gwv, want);
  else
    edge_bwd_tile_kernel<FP, T><<<grid, 256, lds, st>>>(pq, np3, c2pre, edge_next, agg, dedge_next, dedge_prev,
                                                        dpq, dnp3, dc2pre, C, B, g, d, w, gwv, want);
  return true;
}

template <typename T>
void launch_edge_bwd(const T *pq, const T *np3, const T *c2pre, const T *edge_next, const T *agg,
                     const T *dedge_next, T *dedge_prev, T *dpq, T *dnp3, T *dc2pre, int C, int B,
                     const Graph &g, Dims d, const PassW<T> &w, const PassW<T> *gw, hipStream_t st) {
  const int lg = d.FeP / 4;
  const PassW<T> gwv = gw ? *gw : w;
  const int want = gw ? 1 : 0;
  const int64_t threads = (int64_t)C * g.E * lg;
  if (threads == 0) return;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  // dnp3's Wk block is accumulated atomically by every variant; dpq only by the variants that scatter
  // into it (the two-pass kernel writes every row of both halves exactly once)
  (void)hipMemsetAsync(dnp3, 0, (size_t)C * g.N * 6 * d.FeP * sizeof(T), st);
  if (!edge_bwd_two_pass(g, d.FeP, sizeof(T)) || !agg)
    (void)hipMemsetAsync(dpq, 0, (size_t)C * g.E * 4 * d.FeP * sizeof(T), st);
  static const bool simple = getenv("RN_POTGNN_BWD_SIMPLE") && atoi(getenv("RN_POTGNN_BWD_SIMPLE")) != 0;
  bool done = false;
  if (!simple && agg) {
    switch (d.FeP) {
      case 16: done = launch_edge_bwd_tile<16, T>(pq, np3, c2pre, edge_next, agg, dedge_next, dedge_prev, dpq, dnp3, dc2pre, C, B, g, d, w, gwv, want, st); break;
      case 32: done = launch_edge_bwd_tile<32, T>(pq, np3, c2pre, edge_next, agg, dedge_next, dedge_prev, dpq, dnp3, dc2pre, C, B, g, d, w, gwv, want, st); break;
      case 64: done = launch_edge_bwd_tile<64, T>(pq, np3, c2pre, edge_next, agg, dedge_next, dedge_prev, dpq, dnp3, dc2pre, C, B, g, d, w, gwv, want, st); break;
      case 128: done = launch_edge_bwd_tile<128, T>(pq, np3, c2pre, edge_next, agg, dedge_next, dedge_prev, dpq, dnp3, dc2pre, C, B, g, d, w, gwv, want, st); break;
    }
  }
#define CALL(LGV)                                                                                    \
  if (!done)                                                                                         \
    edge_bwd_simple_kernel<LGV, T><<<blocks, 256, 0, st>>>(pq, np3, c2pre, edge_next, dedge_next,    \
                                                           dedge_prev, dpq, dnp3, dc2pre, C, B, g, d, w, \
                                                           gwv, want);                               \
  q_gather_kernel<LGV, T><<<(unsigned)(((int64_t)C * g.N * lg + 255) / 256), 256, 0, st>>>(dpq, dnp3, C, g)
  RN_LG_SWITCH(d.FeP, CALL)
#undef CALL
}
template void launch_edge_bwd<float>(const float *, const float *, const float *, const float *,
                                     const float *, const float *, float *, float *, float *, float *,
                                     int, int, const Graph &, Dims, const PassW<float> &,
                                     const PassW<float> *, hipStream_t);
template void launch_edge_bwd<double>(const double *, const double *, const double *, const double *,
                                      const double *, const double *, double *, double *, double *,
                                      double *, int, int, const Graph &, Dims, const PassW<double> &,
                                      const PassW<double> *, hipStream_t);

template <typename T>
void launch_prod_bwd(const T *dprod, const T *node, T *dnode, int C, int B, const Graph &g, Dims d,
                     hipStream_t st) {
  const int lg = d.FnP / 4;
  const int64_t threads = (int64_t)C * g.N * lg;
  if (threads == 0) return;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
#define CALL(LGV) prod_bwd_kernel<LGV, T><<<blocks, 256, 0, st>>>(dprod, node, dnode, C, B, g)
  RN_LG_SWITCH(d.FnP, CALL)
#undef CALL
}
template void launch_prod_bwd<float>(const float *, const float *, float *, int, int, const Graph &,
                                     Dims, hipStream_t);
template void launch_prod_bwd<double>(const double *, const double *, double *, int, int,
                                      const Graph &, Dims, hipStream_t);

template <typename T>
void launch_prod_fwd(const T *node, T *prod, int64_t rows, const Graph &g, Dims d, hipStream_t st) {
  const int lg = d.FnP / 4;
  if (rows == 0) return;
  const unsigned blocks = (unsigned)((rows * lg + 255) / 256);
#define CALL(LGV) prod_fwd_kernel<LGV, T><<<blocks, 256, 0, st>>>(node, prod, rows, g)
  RN_LG_SWITCH(d.FnP, CALL)
#undef CALL
}
template void launch_prod_fwd<float>(const float *, float *, int64_t, const Graph &, Dims, hipStream_t);
template void launch_prod_fwd<double>(const double *, double *, int64_t, const Graph &, Dims, hipStream_t);

template <typename T>
void launch_node_bwd(const T *npc1, const T *bc1, const T *node_next, const T *dnode_next,
                     T *dnode_prev, T *dbc1, T *dnpc1, int C, int B, const Graph &g, Dims d,
                     const PassW<T> &w, const PassW<T> *gw, hipStream_t st) {
  const int lg = d.FnP / 4;
  const PassW<T> gwv = gw ? *gw : w;
  const int want = gw ? 1 : 0;
  const int64_t threads = (int64_t)C * g.N * lg;
  if (threads == 0) return;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
#define CALL(LGV)                                                                                 \
  node_bwd_kernel<LGV, T><<<blocks, 256, 0, st>>>(npc1, bc1, node_next, dnode_next, dnode_prev, dbc1, \
                                                  dnpc1, C, B, g, d, w, gwv, want)
  RN_LG_SWITCH(d.FnP, CALL)
#undef CALL
}
template void launch_node_bwd<float>(const float *, const float *, const float *, const float *,
                                     float *, float *, float *, int, int, const Graph &, Dims,
                                     const PassW<float> &, const PassW<float> *, hipStream_t);
template void launch_node_bwd<double>(const double *, const double *, const double *, const double *,
                                      double *, double *, double *, int, int, const Graph &, Dims,
                                      const PassW<double> &, const PassW<double> *, hipStream_t);

}  // namespace rn

// ================================================================================ training
// Weight-gradient and BatchNorm(train) pieces (correctness-first; config 5 of BASELINE.json).
namespace rn {

// dWT[k][n] += sum_r X[r][k] * dY[r][n]   (WT = the forward's [K][N] layout), plus optional
// column sums of dY (bias gradient).  amode 1: X row r = node[b_e] * node[a_e] (c2 operand).
// One thread per (k, n) and row chunk; partial sums meet in global atomics.
template <typename T>
__global__ void gemm_tn_kernel(const T *__restrict__ X, int ldx, const T *__restrict__ dY, int ldy,
                               int64_t R, int K, int N, T *__restrict__ dWT, int ldw,
                               T *__restrict__ dbias, int amode, const T *__restrict__ node, Graph g,
                               int rows_per_block) {
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < K * N; idx += gridDim.x * blockDim.x) {
    const int k = idx / N, n = idx % N;
    T acc = 0, bsum = 0;
    for (int64_t r = r0; r < r1; ++r) {
      T x;
      if (amode == 0) {
        x = X[r * ldx + k];
      } else {
        const int64_t s = r / g.E;
        const int e = (int)(r % g.E);
        x = node[(s * g.N + g.edge_b[e]) * ldx + k] * node[(s * g.N + g.edge_a[e]) * ldx + k];
      }
      const T dy = dY[r * ldy + n];
      acc += x * dy;
      bsum += dy;
    }
    atomicAdd(dWT + (int64_t)k * ldw + n, acc);
    if (dbias && k == 0) atomicAdd(dbias + n, bsum);
  }
}
// Tiled version: a workgroup stages 128 rows of X (all K columns) and of a 64-column chunk of
// dY in LDS and forms the K x 64 partial product with register blocking (thread (tk, tn) owns
// k = tk + 16 i, n = 4 tn .. 4 tn + 3); workgroups walk row tiles persistently and add their
// partial sums to the gradient with one atomic per output element.
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void gemm_tn_tiled_kernel(const T *__restrict__ X, int ldx,
                                                            const T *__restrict__ dY, int ldy, int64_t R,
                                                            int K, int N, T *__restrict__ dWT, int ldw,
                                                            T *__restrict__ dbias, int amode,
                                                            const T *__restrict__ node, Graph g) {
  constexpr int BR = 128, NC = 64, KI = KMAX / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *xs = reinterpret_cast<T *>(smem_raw);   // [BR][KMAX + 1]
  T *ys = xs + BR * (KMAX + 1);              // [BR][NC]
  const int n0 = blockIdx.y * NC;
  const int tk = threadIdx.x / 16, tn = threadIdx.x % 16;
  T acc[KI][4];
#pragma unroll
  for (int i = 0; i < KI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  T bsum[4] = {0, 0, 0, 0};
  const int64_t tiles = (R + BR - 1) / BR;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t r0 = tile * BR;
    for (int i = threadIdx.x; i < BR * KMAX; i += 256) {
      const int r = i / KMAX, k = i % KMAX;
      const int64_t row = r0 + r;
      T v = 0;
      if (row < R && k < K) {
        if (amode == 0) {
          v = X[row * ldx + k];
        } else {
          const int64_t s = row / g.E;
          const int e = (int)(row % g.E);
          v = node[(s * g.N + g.edge_b[e]) * ldx + k] * node[(s * g.N + g.edge_a[e]) * ldx + k];
        }
      }
      xs[r * (KMAX + 1) + k] = v;
    }
    for (int i = threadIdx.x; i < BR * NC; i += 256) {
      const int r = i / NC, n = i % NC;
      const int64_t row = r0 + r;
      ys[i] = (row < R && n0 + n < N) ? dY[row * ldy + n0 + n] : (T)0;
    }
    __syncthreads();
    for (int r = 0; r < BR; ++r) {
      const T y0 = ys[r * NC + 4 * tn], y1 = ys[r * NC + 4 * tn + 1], y2 = ys[r * NC + 4 * tn + 2],
              y3 = ys[r * NC + 4 * tn + 3];
#pragma unroll
      for (int i = 0; i < KI; ++i) {
        const T x = xs[r * (KMAX + 1) + tk + 16 * i];
        acc[i][0] += x * y0;
        acc[i][1] += x * y1;
        acc[i][2] += x * y2;
        acc[i][3] += x * y3;
      }
      if (tk == 0) {
        bsum[0] += y0; bsum[1] += y1; bsum[2] += y2; bsum[3] += y3;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int k = tk + 16 * i;
    if (k < K)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + 4 * tn + j < N) atomicAdd(dWT + (int64_t)k * ldw + n0 + 4 * tn + j, acc[i][j]);
  }
  if (dbias && tk == 0)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + 4 * tn + j < N) atomicAdd(dbias + n0 + 4 * tn + j, bsum[j]);
}

// Weight gradients on the matrix pipe (float32): dW^T[k][n] += sum_r X[r][k] dY[r][n] is a
// [K x R] x [R x N] product with a huge reduction length, so v_mfma_f32_32x32x2_f32 takes TWO
// rows per instruction: lane (m, h) supplies X[2s+h][k0+m] and dY[2s+h][n0+m] -- both plain
// coalesced dword loads, no transposition, no LDS.  A workgroup owns a block of rows; wave w
// owns 32 output columns and keeps all K/32 accumulator tiles; partial sums leave with one
// atomic per element and workgroup.  (The VALU kernel above spent 164 us per call on this.)
typedef float f32x16_t __attribute__((ext_vector_type(16)));
// PART: instead of adding its [K][N] block sum to dWT with atomics (512 workgroups x K x N float
// atomics per product, up to 1024 of them on one address for the bias: measured 40-60 us of a 60-120 us
// launch), the workgroup stores it to its own slice of a partial-sum arena (`dWT` = arena base of this
// product, `ldw` = slice stride in elements); tn_reduce_kernel adds the slices up at the end of the
// reverse pass, one writer per gradient element (which also makes the weight gradients reproducible).
template <int KT, int AMODE, bool PART>
__global__ __launch_bounds__(256) void gemm_tn_mfma_kernel(const float *__restrict__ X, int ldx,
                                                           const float *__restrict__ dY, int ldy, int64_t R,
                                                           int K, int N, float *__restrict__ dWT, int ldw,
                                                           float *__restrict__ dbias,
                                                           const float *__restrict__ node, Graph g,
                                                           int rows_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int ntile = blockIdx.y * 4 + wave;
  if (ntile * 32 >= N) return;  // (no barriers in this kernel)
  const int ncol = ntile * 32 + l31;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  f32x16_t acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  const int64_t s0 = (AMODE == 1 && g.E > 0) ? r0 / g.E : 0, e0 = (AMODE == 1) ? r0 - s0 * g.E : 0;
#ifndef RN_TN_U
#define RN_TN_U 4
#endif
  constexpr int U = RN_TN_U;  // row pairs per step
  if constexpr (AMODE == 0) {
    // two steps in flight: the next step's rows are requested before this step's MFMAs (the loads of a step
    // otherwise wait behind 8 KT / 2 x 64 cycles of matrix-pipe issue, and the pipe then idles for their latency)
    float a[2][U][KT], b[2][U];
    auto request = [&](int64_t rb, int buf) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = rb + 2 * u + h;
        const bool ok = row < r1;
        b[buf][u] = ok ? dY[row * ldy + ncol] : 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const int k = 32 * t + l31;
          a[buf][u][t] = (ok && k < K) ? X[row * ldx + k] : 0.f;
        }
      }
    };
    auto consume = [&](int buf) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bsum += b[buf][u];
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][u][t], b[buf][u], acc[t], 0, 0, 0);
      }
    };
    request(r0, 0);
    for (int64_t rb = r0; rb < r1; rb += 4 * U) {
      request(rb + 2 * U, 1);  // (rows past r1 load nothing and contribute zeros)
      consume(0);
      request(rb + 4 * U, 0);
      consume(1);
    }
  } else {
    for (int64_t rb = r0; rb < r1; rb += 2 * U) {
      float a[U][KT], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = rb + 2 * u + h;
        const bool ok = row < r1;
        b[u] = ok ? dY[row * ldy + ncol] : 0.f;
        int64_t nb = 0, na = 0;
        if (ok) {  // (frame, edge) of the row without a 64-bit division per row
          int64_t s = s0;
          int64_t e64 = e0 + (row - r0);
          while (e64 >= g.E) {
            e64 -= g.E;
            ++s;
          }
          const int e = (int)e64;
          nb = (s * g.N + g.edge_b[e]) * ldx;
          na = (s * g.N + g.edge_a[e]) * ldx;
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const int k = 32 * t + l31;
          a[u][t] = (ok && k < K) ? node[nb + k] * node[na + k] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bsum += b[u];
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u], acc[t], 0, 0, 0);
      }
    }
  }
  if (PART) {
    float *slice = dWT + (size_t)blockIdx.x * ldw;  // [K][N] block sum, then [N] column sums of dY
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (k < K) slice[(size_t)k * N + ncol] = acc[t][i];
      }
    bsum += __shfl_xor(bsum, 32);
    if (h == 0) slice[(size_t)K * N + ncol] = bsum;
    return;
  }
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (k < K) atomicAdd(dWT + (int64_t)k * ldw + ncol, acc[t][i]);
    }
  if (dbias) atomicAdd(dbias + ncol, bsum);  // both lane halves hold half of the rows
}

// The same product with the operand rows staged through LDS: the workgroup loads 32-row tiles of X and dY with 16-byte
// accesses (a quarter of the load instructions of the per-lane 4-byte form above, which keeps the texture addresser ~75 %
// busy and the matrix pipe 27 %), the next tile in registers while the current one multiplies, and the four waves read
// their operands from LDS (row strides padded by 32 floats: the two lane halves of an instruction read different rows).
// K <= 128 (KT 32-column tiles of X), a 128-column slab of dY per workgroup; always writes a partial slice.
template <int KT>
__global__ __launch_bounds__(256) void gemm_tn_lds_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ dY,
                                                          int ldy, int64_t R, int K, int N, float *__restrict__ part,
                                                          int slice_stride, int rows_per_block) {
  constexpr int RT = 32, LDX = 32 * KT + 32, LDY = 128 + 32;
  __shared__ __attribute__((aligned(16))) float xs[2][RT * LDX];
  __shared__ __attribute__((aligned(16))) float ys[2][RT * LDY];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.y * 128;
  const int nslab = min(128, N - n0);           // columns of this workgroup (a multiple of 32)
  const bool active = wave * 32 < nslab;        // waves beyond the slab only help loading
  const int ncol = n0 + wave * 32 + l31;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  constexpr int XV = RT * KT * 8 / 256;         // float4 per thread of an X tile (KT 32-column tiles = 8 KT float4 per row)
  constexpr int YV = RT * 32 / 256;             // float4 per thread of a dY tile (128 columns)
  float4 px[XV], py[YV];
  auto request = [&](int64_t rb) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + 256 * j, r = i / (8 * KT), c = (i % (8 * KT)) * 4;
      const int64_t row = rb + r;
      px[j] = (row < r1 && c < K) ? *reinterpret_cast<const float4 *>(X + row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < YV; ++j) {
      const int i = tid + 256 * j, r = i / 32, c = (i % 32) * 4;
      const int64_t row = rb + r;
      py[j] = (row < r1 && c < nslab) ? *reinterpret_cast<const float4 *>(dY + row * ldy + n0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int b) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + 256 * j, r = i / (8 * KT), c = (i % (8 * KT)) * 4;
      *reinterpret_cast<float4 *>(&xs[b][r * LDX + c]) = px[j];
    }
#pragma unroll
    for (int j = 0; j < YV; ++j) {
      const int i = tid + 256 * j, r = i / 32, c = (i % 32) * 4;
      *reinterpret_cast<float4 *>(&ys[b][r * LDY + c]) = py[j];
    }
  };
  f32x16_t acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  request(r0);
  int b = 0;
  for (int64_t rb = r0; rb < r1; rb += RT, b ^= 1) {
    commit(b);
    if (rb + RT < r1) request(rb + RT);
    __syncthreads();  // tile b complete (tile b^1 is free: every wave has finished the previous tile's products)
    if (active) {
#pragma unroll 4
      for (int rp = 0; rp < RT / 2; ++rp) {
        const float bv = ys[b][(2 * rp + h) * LDY + wave * 32 + l31];
        bsum += bv;
#pragma unroll
        for (int t = 0; t < KT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[b][(2 * rp + h) * LDX + 32 * t + l31], bv, acc[t], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  float *slice = part + (size_t)blockIdx.x * slice_stride;  // [K][N] block sum, then [N] column sums of dY
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (k < K) slice[(size_t)k * N + ncol] = acc[t][i];
    }
  bsum += __shfl_xor(bsum, 32);
  if (h == 0) slice[(size_t)K * N + ncol] = bsum;
}

// Sum of the partial slices of every deferred product of a reverse pass: one workgroup per 32 gradient
// elements (8 lane groups of 32 take every 8th slice, then an LDS sum), one writer per element.
__global__ __launch_bounds__(256) void tn_reduce_kernel(TnReduceTable t) {
  __shared__ float red[256];
  int o = 0;
  while (o + 1 < t.num && (int)blockIdx.x >= t.wg_begin[o + 1]) ++o;
  const float *part = t.ops[o].part;
  const int K = t.ops[o].K, N = t.ops[o].N, nblk = t.ops[o].nblk, ldw = t.ops[o].ldw;
  float *dst = t.ops[o].dst, *dbias = t.ops[o].dbias;
  const size_t stride = (size_t)(K + 1) * N;
  const int e = ((int)blockIdx.x - t.wg_begin[o]) * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  const int count = K * N + (dbias ? N : 0);
  float sum = 0.f;
  if (e < count) {
    const float *p = part + e;
    int b = sl;
    for (; b + 24 < nblk; b += 32) {  // four independent loads in flight
      const float v0 = p[(size_t)b * stride], v1 = p[(size_t)(b + 8) * stride];
      const float v2 = p[(size_t)(b + 16) * stride], v3 = p[(size_t)(b + 24) * stride];
      sum += (v0 + v1) + (v2 + v3);
    }
    for (; b < nblk; b += 8) sum += p[(size_t)b * stride];
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  if (sl == 0 && e < count) {
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) total += red[32 * i + threadIdx.x];
    if (e < K * N) dst[(size_t)(e / N) * ldw + e % N] += total;
    else dbias[e - K * N] += total;
  }
}

void launch_tn_reduce(TnDeferred &df, hipStream_t st) {
  if (df.table.num == 0) return;
  tn_reduce_kernel<<<(unsigned)df.table.wg_begin[df.table.num], 256, 0, st>>>(df.table);
  df.table.num = 0;
  df.used = 0;
}

static int tn_rows_per_block(int64_t R) {
  static const int target_blocks = getenv("RN_POTGNN_TN_BLOCKS") ? std::max(1, atoi(getenv("RN_POTGNN_TN_BLOCKS"))) : 512;
  const int rows_per_block = (int)((R + target_blocks - 1) / target_blocks);
  return std::max(64, (rows_per_block + 7) / 8 * 8);
}

size_t tn_partial_elems(int64_t R, int K, int N) {
  const int rpb = tn_rows_per_block(R);
  return (size_t)((R + rpb - 1) / rpb) * (size_t)(K + 1) * N;
}

static bool launch_gemm_tn_mfma(const float *X, int ldx, const float *dY, int ldy, int64_t R, int K, int N,
                                float *dWT, int ldw, float *dbias, int amode, const float *node,
                                const Graph &g, TnDeferred *defer, hipStream_t st) {
  static const bool off = getenv("RN_POTGNN_BWD_VALU_GEMM") && atoi(getenv("RN_POTGNN_BWD_VALU_GEMM")) != 0;
  if (off || K > 128 || N % 32 != 0) return false;
  // ~512 workgroups or >= 64 rows each: enough parallelism, few partial slices per output element
  const int rows_per_block = tn_rows_per_block(R);
  const dim3 grid((unsigned)((R + rows_per_block - 1) / rows_per_block), (unsigned)((N + 127) / 128));
  const int kt = (K + 31) / 32;
  const size_t need = (size_t)grid.x * (size_t)(K + 1) * N;
  bool part = defer && amode == 0 && defer->arena && defer->used + need <= defer->capacity;
  if (part && defer->table.num == TnReduceTable::MAX_OPS) launch_tn_reduce(*defer, st), part = defer->used + need <= defer->capacity;
  float *out = dWT;
  int ld_out = ldw;
  if (part) {
    out = defer->arena + defer->used;
    ld_out = (K + 1) * N;
    TnReduceTable &t = defer->table;
    t.ops[t.num] = TnReduceOp{out, dWT, dbias, (int)grid.x, K, N, ldw};
    if (t.num == 0) t.wg_begin[0] = 0;
    t.wg_begin[t.num + 1] = t.wg_begin[t.num] + (K * N + (dbias ? N : 0) + 31) / 32;
    ++t.num;
    defer->used += need;
  }
  static const bool lds_form = !(getenv("RN_POTGNN_TN_LDS") && atoi(getenv("RN_POTGNN_TN_LDS")) == 0);
  if (part && lds_form && ldx % 4 == 0 && ldy % 4 == 0 && K % 4 == 0) {
    switch (kt) {
      case 1: gemm_tn_lds_kernel<1><<<grid, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N, out, ld_out, rows_per_block); break;
      case 2: gemm_tn_lds_kernel<2><<<grid, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N, out, ld_out, rows_per_block); break;
      case 3: gemm_tn_lds_kernel<3><<<grid, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N, out, ld_out, rows_per_block); break;
      default: gemm_tn_lds_kernel<4><<<grid, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N, out, ld_out, rows_per_block); break;
    }
    return true;
  }
#define RN_TNM(KTV)                                                                                           \
  do {                                                                                                        \
    if (part) gemm_tn_mfma_kernel<KTV, 0, true><<<grid, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N, out, ld_out,  \
                                                                      dbias, node, g, rows_per_block);        \
    else if (amode == 0) gemm_tn_mfma_kernel<KTV, 0, false><<<grid, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N,   \
                                                                      dWT, ldw, dbias, node, g, rows_per_block); \
    else gemm_tn_mfma_kernel<KTV, 1, false><<<grid, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N, dWT, ldw, dbias,  \
                                                           node, g, rows_per_block);                          \
  } while (0)
  switch (kt) {
    case 1: RN_TNM(1); break;
    case 2: RN_TNM(2); break;
    case 3: RN_TNM(3); break;
    default: RN_TNM(4); break;
  }
#undef RN_TNM
  return true;
}

template <typename T>
void launch_gemm_tn(const T *X, int ldx, const T *dY, int ldy, int64_t R, int K, int N, T *dWT,
                    int ldw, T *dbias, int amode, const T *node, const Graph &g, hipStream_t st,
                    TnDeferred *defer) {
  if (R == 0) return;
  if constexpr (sizeof(T) == 4) {
    if (launch_gemm_tn_mfma(X, ldx, dY, ldy, R, K, N, dWT, ldw, dbias, amode, node, g, defer, st)) return;
  }
  const int64_t tiles = (R + 127) / 128;
  dim3 grid((unsigned)(tiles < 64 ? tiles : 64), (unsigned)((N + 63) / 64));
#define RN_TN(KM)                                                                                   \
  do {                                                                                              \
    const size_t lds = ((size_t)128 * (KM + 1) + 128 * 64) * sizeof(T);                             \
    auto kern = &gemm_tn_tiled_kernel<T, KM>;                                                       \
    if (lds > 48 * 1024)                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    kern<<<grid, 256, lds, st>>>(X, ldx, dY, ldy, R, K, N, dWT, ldw, dbias, amode, node, g);        \
  } while (0)
  // (float64 at K = 128 would need 197 KiB of LDS for the tiled kernel: the plain kernel takes it)
  if (K <= 16) RN_TN(16);
  else if (K <= 32) RN_TN(32);
  else if (K <= 64) RN_TN(64);
  else if (K <= 128 && sizeof(T) == 4) RN_TN(128);
  else {
    const int rows_per_block = 512;
    dim3 g2((unsigned)((K * N + 255) / 256), (unsigned)((R + rows_per_block - 1) / rows_per_block));
    if (g2.x > 64) g2.x = 64;
    gemm_tn_kernel<T><<<g2, 256, 0, st>>>(X, ldx, dY, ldy, R, K, N, dWT, ldw, dbias, amode, node, g,
                                          rows_per_block);
  }
#undef RN_TN
}
template void launch_gemm_tn<float>(const float *, int, const float *, int, int64_t, int, int,
                                    float *, int, float *, int, const float *, const Graph &,
                                    hipStream_t, TnDeferred *);
template void launch_gemm_tn<double>(const double *, int, const double *, int, int64_t, int, int,
                                     double *, int, double *, int, const double *, const Graph &,
                                     hipStream_t, TnDeferred *);

// ---- BatchNorm1d in training mode over all R rows (_gnn.py:534; batch statistics)
// stats[c] = sum z, stats[W + c] = sum z^2 (float64 accumulators)
// Thread layout of the four kernels below: 256 threads = row lanes x column groups of 4 (W / 4 groups; W = 64: 16 x 16);
// a block walks `rows_per_block` rows with 16-byte accesses.  Per-column quantities that need float64 division / sqrt
// are computed once per block into LDS, not per element (the per-element form made the elementwise kernels 5x slower
// than their traffic).
template <typename T>
__global__ __launch_bounds__(256) void col_sums_kernel(const T *__restrict__ z, int64_t R, int W,
                                                       double *__restrict__ stats, int rows_per_block) {
  __shared__ double red[2][256][4];
  const int cg = W / 4, rl = 256 / cg;             // column groups, row lanes
  const int c4 = (threadIdx.x % cg) * 4, rsub = threadIdx.x / cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rsub < rl)
    for (int64_t r = r0 + rsub; r < r1; r += rl) {
      const Vec4<T> v = load4<T>(z + r * W + c4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s[k] += (double)v.v[k];
        q[k] += (double)v.v[k] * (double)v.v[k];
      }
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][threadIdx.x][k] = s[k];
    red[1][threadIdx.x][k] = q[k];
  }
  __syncthreads();
  if (threadIdx.x < 2 * W) {  // one thread per (sum kind, column): add the row lanes in a fixed order
    const int kind = threadIdx.x / W, c = threadIdx.x % W;
    double t = 0;
    for (int j = 0; j < rl; ++j) t += red[kind][j * cg + c / 4][c % 4];
    atomicAdd(stats + kind * W + c, t);
  }
}
// (float32: the hardware exp2 / log2 form of the projection epilogues, abs error ~1e-7; float64: libm)
__device__ __forceinline__ float ssp_hw(float x) { return ssp_fast(x); }
__device__ __forceinline__ double ssp_hw(double x) { return ssp(x); }
// h = ssp(gamma * (z - mean) * rstd + beta);  mean/rstd derived from the sums; also writes
// batch mean and biased variance (for the running-statistics update) into mv[0:W], mv[W:2W].
template <typename T>
__global__ __launch_bounds__(256) void bn_train_fwd_kernel(const T *__restrict__ z, int64_t R, int W, int F,
                                                           const double *__restrict__ stats, double count,
                                                           const T *__restrict__ gamma, const T *__restrict__ beta,
                                                           T *__restrict__ h, T *__restrict__ mv, int rows_per_block) {
  __shared__ T s_mean[256], s_rstd[256], s_gamma[256], s_beta[256];
  for (int c = threadIdx.x; c < W; c += 256) {
    T mean_t = 0, rstd = 0;
    if (c < F) {
      const double mean = stats[c] / count;
      double var = stats[W + c] / count - mean * mean;
      var = var > 0 ? var : 0;
      rstd = (T)(1.0 / sqrt(var + 1e-5));
      mean_t = (T)mean;
      if (blockIdx.x == 0) {
        mv[c] = (T)mean;
        mv[W + c] = (T)var;
      }
    }
    s_mean[c] = mean_t;
    s_rstd[c] = rstd;
    s_gamma[c] = c < F ? gamma[c] : (T)0;
    s_beta[c] = c < F ? beta[c] : (T)0;
  }
  __syncthreads();
  const int cg = W / 4, rl = 256 / cg;
  const int c4 = (threadIdx.x % cg) * 4, rsub = threadIdx.x / cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  if (rsub >= rl) return;
  for (int64_t r = r0 + rsub; r < r1; r += rl) {
    const Vec4<T> v = load4<T>(z + r * W + c4);
    Vec4<T> o;
#pragma unroll
    for (int k = 0; k < 4; ++k)  // (the same operation order as the per-element form: results unchanged)
      o.v[k] = (c4 + k < F) ? ssp_hw(s_gamma[c4 + k] * ((v.v[k] - s_mean[c4 + k]) * s_rstd[c4 + k]) + s_beta[c4 + k]) : (T)0;
    store4(h + r * W + c4, o);
  }
}
// BatchNorm backward.  dy (cotangent of the BN output, i.e. already through ssp') in `d`:
//   pass 1 (bn_bwd_sums): sums[c] = sum dy, sums[W+c] = sum dy * zhat      (also = dbeta, dgamma)
//   pass 2 (bn_bwd_apply): d <- gamma * rstd * (dy - mean(dy) - zhat * mean(dy zhat))
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const T *__restrict__ dy, const T *__restrict__ z, int64_t R,
                                                          int W, int F, const double *__restrict__ stats, double count,
                                                          double *__restrict__ sums, int rows_per_block) {
  __shared__ double red[2][256][4];
  __shared__ double s_mean[256], s_rstd[256];
  for (int c = threadIdx.x; c < W; c += 256) {
    const double mean = stats[c] / count;
    double var = stats[W + c] / count - mean * mean;
    var = var > 0 ? var : 0;
    s_mean[c] = mean;
    s_rstd[c] = 1.0 / sqrt(var + 1e-5);
  }
  __syncthreads();
  const int cg = W / 4, rl = 256 / cg;
  const int c4 = (threadIdx.x % cg) * 4, rsub = threadIdx.x / cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  if (rsub < rl)
    for (int64_t r = r0 + rsub; r < r1; r += rl) {
      const Vec4<T> g = load4<T>(dy + r * W + c4), v = load4<T>(z + r * W + c4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] += (double)g.v[k];
        b[k] += (double)g.v[k] * ((double)v.v[k] - s_mean[c4 + k]) * s_rstd[c4 + k];
      }
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][threadIdx.x][k] = a[k];
    red[1][threadIdx.x][k] = b[k];
  }
  __syncthreads();
  if (threadIdx.x < 2 * W) {
    const int kind = threadIdx.x / W, c = threadIdx.x % W;
    if (c < F) {
      double t = 0;
      for (int j = 0; j < rl; ++j) t += red[kind][j * cg + c / 4][c % 4];
      atomicAdd(sums + kind * W + c, t);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(T *__restrict__ d, const T *__restrict__ z, int64_t R, int W,
                                                           int F, const double *__restrict__ stats, double count,
                                                           const double *__restrict__ sums,
                                                           const double *__restrict__ own_sums,
                                                           const T *__restrict__ gamma, T *__restrict__ dgamma,
                                                           T *__restrict__ dbeta, int rows_per_block) {
  __shared__ double s_mean[256], s_rstd[256], s_ma[256], s_mb[256], s_g[256];
  for (int c = threadIdx.x; c < W; c += 256) {
    double mean = 0, rstd = 0, ma = 0, mb = 0, gm = 0;
    if (c < F) {
      mean = stats[c] / count;
      double var = stats[W + c] / count - mean * mean;
      var = var > 0 ? var : 0;
      rstd = 1.0 / sqrt(var + 1e-5);
      ma = sums[c] / count;
      mb = sums[W + c] / count;
      gm = (double)gamma[c];
      if (blockIdx.x == 0) {  // parameter gradients from this rank's rows only (averaged across ranks later)
        dbeta[c] += (T)own_sums[c];
        dgamma[c] += (T)own_sums[W + c];
      }
    }
    s_mean[c] = mean;
    s_rstd[c] = rstd;
    s_ma[c] = ma;
    s_mb[c] = mb;
    s_g[c] = gm;
  }
  __syncthreads();
  const int cg = W / 4, rl = 256 / cg;
  const int c4 = (threadIdx.x % cg) * 4, rsub = threadIdx.x / cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  if (rsub >= rl) return;
  for (int64_t r = r0 + rsub; r < r1; r += rl) {
    const Vec4<T> g = load4<T>(d + r * W + c4), v = load4<T>(z + r * W + c4);
    Vec4<T> o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c4 + k;
      const double zhat = ((double)v.v[k] - s_mean[c]) * s_rstd[c];
      o.v[k] = c < F ? (T)(s_g[c] * s_rstd[c] * ((double)g.v[k] - s_ma[c] - zhat * s_mb[c])) : (T)0;
    }
    store4(d + r * W + c4, o);
  }
}

// The four steps are separate launches because a data-parallel run all-reduces the column
// sums between (col_sums, apply) and between (bwd_sums, bwd_apply); `count` is the number of
// rows the statistics cover (all ranks), R the rows of this rank.
// (W, the padded hidden width, is 32, 64 or 128: W / 4 column groups divide 256)
static int bn_rows_per_block(int64_t R) {
  const int64_t per = (R + 1023) / 1024;  // ~1024 blocks
  return (int)std::max<int64_t>(64, (per + 15) / 16 * 16);
}
template <typename T>
void launch_bn_col_sums(const T *z, int64_t R, int W, double *stats, hipStream_t st) {
  (void)hipMemsetAsync(stats, 0, sizeof(double) * 2 * W, st);
  if (R == 0) return;
  const int rpb = bn_rows_per_block(R);
  col_sums_kernel<T><<<(unsigned)((R + rpb - 1) / rpb), 256, 0, st>>>(z, R, W, stats, rpb);
}
template <typename T>
void launch_bn_train_apply(const T *z, int64_t R, int W, int F, const double *stats, double count,
                           const T *gamma, const T *beta, T *h, T *mv, hipStream_t st) {
  if (R == 0) return;
  const int rpb = bn_rows_per_block(R);
  bn_train_fwd_kernel<T><<<(unsigned)((R + rpb - 1) / rpb), 256, 0, st>>>(z, R, W, F, stats, count, gamma, beta, h, mv, rpb);
}
template <typename T>
void launch_bn_bwd_sums(const T *d, const T *z, int64_t R, int W, int F, const double *stats, double count,
                        double *sums, hipStream_t st) {
  (void)hipMemsetAsync(sums, 0, sizeof(double) * 2 * W, st);
  if (R == 0) return;
  const int rpb = bn_rows_per_block(R);
  bn_bwd_sums_kernel<T><<<(unsigned)((R + rpb - 1) / rpb), 256, 0, st>>>(d, z, R, W, F, stats, count, sums, rpb);
}
template <typename T>
void launch_bn_bwd_apply(T *d, const T *z, int64_t R, int W, int F, const double *stats, double count,
                         const double *sums, const double *own_sums, const T *gamma, T *dgamma, T *dbeta,
                         hipStream_t st) {
  if (R == 0) return;
  const int rpb = bn_rows_per_block(R);
  bn_bwd_apply_kernel<T><<<(unsigned)((R + rpb - 1) / rpb), 256, 0, st>>>(d, z, R, W, F, stats, count, sums, own_sums, gamma,
                                                                        dgamma, dbeta, rpb);
}
#define RN_BN_INST(T)                                                                                     \
  template void launch_bn_col_sums<T>(const T *, int64_t, int, double *, hipStream_t);                    \
  template void launch_bn_train_apply<T>(const T *, int64_t, int, int, const double *, double, const T *, \
                                         const T *, T *, T *, hipStream_t);                               \
  template void launch_bn_bwd_sums<T>(const T *, const T *, int64_t, int, int, const double *, double,    \
                                      double *, hipStream_t);                                             \
  template void launch_bn_bwd_apply<T>(T *, const T *, int64_t, int, int, const double *, double,         \
                                       const double *, const double *, const T *, T *, T *, hipStream_t);
RN_BN_INST(float)
RN_BN_INST(double)
#undef RN_BN_INST


// ---- node embedding MLP (Embedding -> ssp -> Linear -> ssp -> Linear, _gnn.py:508-514):
// gradient of the K x Fn table rows w.r.t. emb, W2, b2, W4, b4.  dnode0 [S*N, FnP] is first
// summed per atom type.  One workgroup; K and Fn are tiny.
// sums[k][o] += sum over the rows r (atoms of all frames) of type k of dnode0[r][o]: a block
// walks `rows_per_block` rows, thread o owns column o (private per-type partial sums in LDS),
// then one atomic per (type, column) and block.
template <typename T>
__global__ void type_col_sums_kernel(const T *__restrict__ dnode0, int64_t R, int N, int FnP, int Fn, int K,
                                     const int *__restrict__ atom_type /* [N], or [R] when per_sample */, int per_sample,
                                     T *__restrict__ sums, int rows_per_block) {
  extern __shared__ unsigned char smem_raw[];
  T *acc = reinterpret_cast<T *>(smem_raw);  // [K][Fn]
  const int o = threadIdx.x;
  if (o >= Fn) return;
  for (int k = 0; k < K; ++k) acc[k * Fn + o] = 0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  for (int64_t r = r0; r < r1; ++r) {
    const int k = per_sample ? atom_type[r] : atom_type[r % N];
    if (k >= 0 && k < K) acc[k * Fn + o] += dnode0[r * FnP + o];
  }
  for (int k = 0; k < K; ++k) atomicAdd(sums + k * Fn + o, acc[k * Fn + o]);
}

template <typename T>
__global__ void node_embed_bwd_kernel(const T *__restrict__ type_sums, int S, Graph g, Dims d, int K,
                                      const T *__restrict__ emb, const T *__restrict__ W2,
                                      const T *__restrict__ b2, const T *__restrict__ W4,
                                      T *__restrict__ demb, T *__restrict__ dW2, T *__restrict__ db2,
                                      T *__restrict__ dW4, T *__restrict__ db4) {
  extern __shared__ unsigned char smem_raw[];
  const int Fn = d.Fn;
  T *dt = reinterpret_cast<T *>(smem_raw);  // [K*Fn] d table
  T *a1 = dt + K * Fn;                      // ssp(emb)
  T *z2 = a1 + K * Fn;                      // pre-activation of the first Linear
  T *dz2 = z2 + K * Fn;
  for (int i = threadIdx.x; i < K * Fn; i += blockDim.x) {
    dt[i] = type_sums[i];  // d table: cotangent of the initial embedding summed per atom type
    a1[i] = ssp(emb[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * Fn; i += blockDim.x) {
    const int k = i / Fn, o = i % Fn;
    T acc = b2[o];
    for (int c = 0; c < Fn; ++c) acc += a1[k * Fn + c] * W2[o * Fn + c];
    z2[i] = acc;
  }
  __syncthreads();
  // table = W4 ssp(z2) + b4
  for (int i = threadIdx.x; i < Fn * Fn; i += blockDim.x) {
    const int o = i / Fn, c = i % Fn;
    T acc = 0;
    for (int k = 0; k < K; ++k) acc += dt[k * Fn + o] * ssp(z2[k * Fn + c]);
    dW4[i] += acc;
  }
  for (int o = threadIdx.x; o < Fn; o += blockDim.x) {
    T acc = 0;
    for (int k = 0; k < K; ++k) acc += dt[k * Fn + o];
    db4[o] += acc;
  }
  for (int i = threadIdx.x; i < K * Fn; i += blockDim.x) {
    const int k = i / Fn, c = i % Fn;
    T acc = 0;
    for (int o = 0; o < Fn; ++o) acc += dt[k * Fn + o] * W4[o * Fn + c];
    dz2[i] = acc * sigmoid_acc(z2[i]);  // ssp' = sigmoid
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Fn * Fn; i += blockDim.x) {
    const int o = i / Fn, c = i % Fn;
    T acc = 0;
    for (int k = 0; k < K; ++k) acc += dz2[k * Fn + o] * a1[k * Fn + c];
    dW2[i] += acc;
  }
  for (int o = threadIdx.x; o < Fn; o += blockDim.x) {
    T acc = 0;
    for (int k = 0; k < K; ++k) acc += dz2[k * Fn + o];
    db2[o] += acc;
  }
  for (int i = threadIdx.x; i < K * Fn; i += blockDim.x) {
    const int k = i / Fn, c = i % Fn;
    T acc = 0;
    for (int o = 0; o < Fn; ++o) acc += dz2[k * Fn + o] * W2[o * Fn + c];
    demb[i] += acc * sigmoid_acc(emb[i]);
  }
}
template <typename T>
void launch_node_embed_bwd(const T *dnode0, int S, const Graph &g, Dims d, int K, const T *emb,
                           const T *W2, const T *b2, const T *W4, T *demb, T *dW2, T *db2, T *dW4,
                           T *db4, T *sums /* scratch [K * Fn] */,
                           const int *types /* [S*N] atom type per (sample, atom), or null: g.atom_type */, hipStream_t st) {
  const size_t lds = (size_t)4 * K * d.Fn * sizeof(T);
  const size_t need = (size_t)K * d.Fn * sizeof(T);
  (void)hipMemsetAsync(sums, 0, need, st);
  const int64_t R = (int64_t)S * g.N;
  const int rpb = 64, threads = (d.Fn + 63) / 64 * 64;
  type_col_sums_kernel<T><<<(unsigned)((R + rpb - 1) / rpb), threads, need, st>>>(
      dnode0, R, g.N, d.FnP, d.Fn, K, types ? types : g.atom_type, types ? 1 : 0, sums, rpb);
  node_embed_bwd_kernel<T><<<1, 256, lds, st>>>(sums, S, g, d, K, emb, W2, b2, W4, demb, dW2, db2, dW4,
                                                db4);
}
template void launch_node_embed_bwd<float>(const float *, int, const Graph &, Dims, int, const float *,
                                           const float *, const float *, const float *, float *,
                                           float *, float *, float *, float *, float *, const int *, hipStream_t);
template void launch_node_embed_bwd<double>(const double *, int, const Graph &, Dims, int,
                                            const double *, const double *, const double *,
                                            const double *, double *, double *, double *, double *,
                                            double *, double *, const int *, hipStream_t);

}  // namespace rn
