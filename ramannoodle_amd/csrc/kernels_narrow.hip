// Narrow-width PotGNN blocks for gfx950: one LANE per row instead of a lane group per row.
//
// The reference documents ONE set of hyper-parameters (Fn = 5, Fe = 14, P = 4:
// docs/source/notebooks/machine-learning.ipynb:187-192).  At such widths the 64-wide kernels
// (kernels_fused.hip / kernels_agg.hip) pad every row to 16 columns per half, spend DPP butterflies
// on 4-lane groups and move the padding through HBM.  Here a whole row lives in ONE lane's registers:
//   * no padding (compile-time Fn, Fe), no cross-lane reductions at all;
//   * every weight is wave-uniform with a compile-time index, so the compiler keeps the weights in
//     SGPRs (s_load batches) and the projections are plain / packed fma chains -- at K = 5..16 an
//     MFMA tile would be mostly padding (the exception is the readout MLP: below);
//   * NodeBlock (_gnn.py:122-151): one WAVE per tile of atoms streams the tile's in-edge rows -- one contiguous
//     block -- through its own LDS slot, one lane per row, one lane per atom for the sum; no barrier;
//   * EdgeBlock (_gnn.py:200-351): tile-resident like the wide kernel (the source rows Q' of a
//     tile's atoms are built once per frame in LDS and reused by every destination edge entering
//     them), one lane per destination edge, two columns per packed-f32 instruction, two barriers per frame;
//   * readout MLP + edge tensors + per-structure mean (_gnn.py:532-539, 354-415, 658-665) in one
//     launch, one workgroup per frame, the three layers on the exact-float32 MFMA, fixed summation order.
// Embeddings keep the padded [rows][16] float layout of the other kernels; padded columns are written as exact zeros.
// EDGE ROWS ARE IN (b, a) ORDER in this pipeline (row i of a frame = edge Graph::in_edge[i]; geom_rbf_kernel writes them
// so, rn_potgnn_debug_stage undoes it): everything that belongs to an atom tile -- the in-edge rows the NodeBlock sums, the
// destination rows the EdgeBlock reads and the rows it writes -- is then ONE CONTIGUOUS BLOCK of HBM; only the EdgeBlock's
// source rows (the tile's out-edges) are gathered, a whole 64-byte line each.
// Widths: the kernels are instantiated for compile-time widths FN, FE that are either exact (the
// documented 5 / 14 and the 5 / 5 of the reference's own tests) or the model's widths rounded up to a
// multiple of four (PADDED: 4, 8, 12, 16 -- every Fn, Fe <= 16 is covered).  Rounded-up columns carry
// zero weights, biases and LayerNorm parameters, so they stay exact zeros through every product, gate
// and tanh; only the LayerNorm statistics have to leave them out (run-time width f, at most three
// masked columns per half).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace rn {

namespace {


// Weights and LayerNorm parameters are read through the CONSTANT address space: they do not change
// while a kernel runs, every index is wave-uniform, and loads from that address space are scalar
// (s_load_dwordx16 into SGPRs) wherever they sit -- through the generic pointers the compiler falls
// back to per-lane vector loads of the same address as soon as a barrier or a store precedes them
// (2-3x the VGPRs, spills).
typedef const __attribute__((address_space(4))) float *cptr;
__device__ __forceinline__ cptr as_const(const float *p) { return (cptr)p; }

template <int W>
__device__ __forceinline__ void load_row(const float *p, float (&x)[W]) {
  // rows are 16-float (64-byte) aligned: whole float4 pieces, the tail piece may carry padding
#pragma unroll
  for (int j = 0; j < (W + 3) / 4; ++j) {
    const float4 v = *reinterpret_cast<const float4 *>(p + 4 * j);
    const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * j + i < W) x[4 * j + i] = t[i];
  }
}
template <int W, int WP>
__device__ __forceinline__ void store_row(float *p, const float (&x)[W]) {
#pragma unroll
  for (int j = 0; j < WP / 4; ++j) {
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (4 * j + i < W) ? x[4 * j + i] : 0.0f;
    *reinterpret_cast<float4 *>(p + 4 * j) = make_float4(t[0], t[1], t[2], t[3]);
  }
}

// column of the padded [filter | core] layout (half width HP) for logical column c of 2F
template <int F, int HP>
__device__ __forceinline__ constexpr int gcol(int c) {
  return c < F ? c : HP + (c - F);
}

// acc[c] += sum_k W[k * LD + OFF + gcol(c)] * x[k] for the 2F gated columns of a padded [filter|core]
// weight block.  k is the OUTER loop: the 2F weights of one k are two contiguous runs in memory, i.e.
// one or two s_load batches that are consumed at once -- with the column loop outside, the compiler
// loaded every weight up front and spilled hundreds of SGPRs to VGPR lanes (v_writelane/v_readlane).
template <int K, int F, int HP, int LD, int OFF>
__device__ __forceinline__ void gated_matvec(cptr W, const float (&x)[K], float (&acc)[2 * F]) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int c = 0; c < 2 * F; ++c) acc[c] = fmaf(W[k * LD + OFF + gcol<F, HP>(c)], x[k], acc[c]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// LayerNorm(2F) -> sigmoid * tanh over a row held by one lane (torch semantics: biased variance,
// eps 1e-5); g/b are the padded [filter|core] parameter arrays (wave-uniform -> SGPRs).
// PADDED: only the first f of the F columns of each half are real (F - 3 <= f <= F)
template <int F, int HP, bool PADDED = false>
__device__ __forceinline__ void ln_gate_row(const float (&x)[2 * F], cptr g, cptr b, float (&out)[F], int f = F) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 2 * F; ++c) s += x[c];
  const float inv = PADDED ? 1.0f / (float)(2 * f) : 1.0f / (2 * F);
  const float mean = s * inv;
  float d[2 * F], q = 0.f;
#pragma unroll
  for (int c = 0; c < 2 * F; ++c) {
    d[c] = x[c] - mean;
    if (PADDED && (c % F) >= F - 3 && (c % F) >= f) d[c] = 0.f;
    q = fmaf(d[c], d[c], q);
  }
  const float rstd = fast_rsq(q * inv + 1e-5f);
#pragma unroll
  for (int k = 0; k < F; ++k) {
    const float yf = fmaf(d[k], rstd * g[k], b[k]);  // (one scalar operand per instruction)
    const float yc = fmaf(d[F + k], rstd * g[HP + k], b[HP + k]);
    out[k] = gate(yf, yc);
  }
}
// The same for a ZERO-MEAN row (a centred Linear's output: api.hip centre_ops; padded columns exact zeros) and the norm's
// parameters already multiplied by the gate's exp2 scales (PassW::c1_norm_s): no mean, no mask, no scale multiplies
template <int F, int HP>
__device__ __forceinline__ void ln_gate_row_c(const float (&x)[2 * F], cptr gs, cptr bs, float inv2n, float (&out)[F]) {
  float q0 = 0.f, q1 = 0.f;
#pragma unroll
  for (int c = 0; c < 2 * F; c += 2) {
    q0 = fmaf(x[c], x[c], q0);
    q1 = fmaf(x[c + 1], x[c + 1], q1);
  }
  const float rstd = fast_rsq(fmaf(q0 + q1, inv2n, 1e-5f));
#pragma unroll
  for (int k = 0; k < F; ++k) {
    const float yf = fmaf(x[k], rstd * gs[k], bs[k]);
    float yc = fmaf(x[F + k], rstd * gs[HP + k], bs[HP + k]);
    yc = fminf(fmaxf(yc, -43.28f), 43.28f);  // tanh is 1 to 13 digits at |c| = 15: 2 log2e * 15
    const float e1 = fast_exp2(yf), e2 = fast_exp2(yc);
    const float t2 = 1.0f + e2;
    out[k] = (e2 - 1.0f) * fast_rcp(fmaf(e1, t2, t2));
  }
}
template <int F, bool PADDED = false>
__device__ __forceinline__ void ln_row1(const float (&x)[F], cptr g, cptr b, float (&out)[F], int f = F) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < F; ++c) s += x[c];
  const float inv = PADDED ? 1.0f / (float)f : 1.0f / F;
  const float mean = s * inv;
  float d[F], q = 0.f;
#pragma unroll
  for (int c = 0; c < F; ++c) {
    d[c] = x[c] - mean;
    if (PADDED && c >= F - 3 && c >= f) d[c] = 0.f;
    q = fmaf(d[c], d[c], q);
  }
  const float rstd = fast_rsq(q * inv + 1e-5f);
#pragma unroll
  for (int c = 0; c < F; ++c) out[c] = fmaf(d[c], rstd * g[c], b[c]);
}


// ---------------------------------------------------------------- two columns per instruction
// The gfx950 VALU has packed float32 forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE float32 operations
// on a 64-bit register pair at the issue cost of one).  A [filter | core] row of logical width 2F is therefore kept as
// H = ceil(F / 2) column PAIRS per half: x2[j] = filter columns (2j, 2j + 1), x2[H + j] = core columns (2j, 2j + 1).
// The padded weight layout puts both columns of a pair next to each other on an 8-byte boundary, so a pair of weights
// is one 64-bit scalar operand.  With F odd the last pair's second column is a padded one: zero weights, bias and
// LayerNorm parameters keep it at exact zero (the Linears are the CENTRED copies, api.hip centre_ops: projections come
// out with zero row mean over the real columns and exact zeros in the padded ones, so LayerNorm(2F) needs neither a mean
// nor a mask).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v2f *cptr2;
__device__ __forceinline__ v2f ldw2(cptr p) { return *reinterpret_cast<cptr2>(p); }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f bcast2(float x) { return v2f{x, x}; }

template <int F>
constexpr int pairs_of() { return (F + 1) / 2; }

// acc += W x for the gated columns of a padded [filter | core] weight block, two columns per instruction
// (k outermost for the same reason as in gated_matvec: the weights of one k are consumed as they are loaded)
template <int K, int F, int HP, int LD, int OFF>
__device__ __forceinline__ void gated_matvec2(cptr W, const float (&x)[K], v2f (&acc)[2 * pairs_of<F>()]) {
  constexpr int H = pairs_of<F>();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const v2f xk = bcast2(x[k]);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      acc[j] = fma2(ldw2(W + k * LD + OFF + 2 * j), xk, acc[j]);
      acc[H + j] = fma2(ldw2(W + k * LD + OFF + HP + 2 * j), xk, acc[H + j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int F, int HP, int OFF>
__device__ __forceinline__ void load_pairs(cptr v, v2f (&x)[2 * pairs_of<F>()]) {
  constexpr int H = pairs_of<F>();
#pragma unroll
  for (int j = 0; j < H; ++j) {
    x[j] = ldw2(v + OFF + 2 * j);
    x[H + j] = ldw2(v + OFF + HP + 2 * j);
  }
}
// sum of squares of a zero-mean [filter | core] row (padded columns are exact zeros)
template <int H2>
__device__ __forceinline__ float sumsq2(const v2f (&x)[H2]) {
  v2f s0 = x[0] * x[0], s1 = H2 > 1 ? x[1] * x[1] : v2f{0.f, 0.f};
#pragma unroll
  for (int j = 2; j < H2; j += 2) {
    s0 = fma2(x[j], x[j], s0);
    if (j + 1 < H2) s1 = fma2(x[j + 1], x[j + 1], s1);
  }
  s0 += s1;
  return s0.x + s0.y;
}
// sigmoid * tanh of two columns whose pre-activations already carry the exp2 scales (-log2e | 2 log2e):
//   (e2 - 1) / ((1 + e1)(1 + e2)),  (1 + e1)(1 + e2) = t2 + e1 t2 with t2 = 1 + e2: one fma
template <bool CLAMP>
__device__ __forceinline__ v2f gate2s(v2f yf, v2f yc, v2f acc = v2f{0.f, 0.f}) {  // acc + gate
  if (CLAMP) {  // tanh is 1 to 13 digits at |c| = 15: 2 log2e * 15
    yc.x = fminf(fmaxf(yc.x, -43.28f), 43.28f);
    yc.y = fminf(fmaxf(yc.y, -43.28f), 43.28f);
  }
  const v2f e1 = {fast_exp2(yf.x), fast_exp2(yf.y)}, e2 = {fast_exp2(yc.x), fast_exp2(yc.y)};
  const v2f t2 = e2 + 1.0f;
  const v2f den = fma2(e1, t2, t2);
  const v2f rd = {fast_rcp(den.x), fast_rcp(den.y)};
  return fma2(e2 - 1.0f, rd, acc);
}
// LayerNorm(2F) -> gate of a zero-mean row in pairs; gs / bs: the norm's parameters times the exp2 scales
template <int F, int HP, bool CLAMP>
__device__ __forceinline__ void ln_gate_pairs(const v2f (&x)[2 * pairs_of<F>()], cptr gs, cptr bs, float inv2n,
                                              v2f (&out)[pairs_of<F>()]) {
  constexpr int H = pairs_of<F>();
  const v2f rstd2 = bcast2(fast_rsq(fmaf(sumsq2<2 * H>(x), inv2n, 1e-5f)));
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const v2f yf = fma2(x[j], rstd2 * ldw2(gs + 2 * j), ldw2(bs + 2 * j));
    const v2f yc = fma2(x[H + j], rstd2 * ldw2(gs + HP + 2 * j), ldw2(bs + HP + 2 * j));
    out[j] = gate2s<CLAMP>(yf, yc);
  }
}
// LayerNorm(F) of a row in pairs (F even; torch semantics); PADDED: only the first f of the F columns are real
template <int F, bool PADDED>
__device__ __forceinline__ void ln_pairs(const v2f (&x)[F / 2], cptr g, cptr b, v2f (&out)[F / 2], int f = F) {
  static_assert(F % 2 == 0, "column pairs");
  constexpr int H = F / 2;
  v2f s = x[0];
#pragma unroll
  for (int j = 1; j < H; ++j) s += x[j];
  const float inv = PADDED ? 1.0f / (float)f : 1.0f / F;
  const v2f mean2 = bcast2((s.x + s.y) * inv);
  v2f d[H], q = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < H; ++j) {
    d[j] = x[j] - mean2;
    if (PADDED && 2 * j >= F - 3 && 2 * j >= f) d[j].x = 0.f;
    if (PADDED && 2 * j + 1 >= F - 3 && 2 * j + 1 >= f) d[j].y = 0.f;
    q = fma2(d[j], d[j], q);
  }
  const v2f rstd2 = bcast2(fast_rsq(fmaf(q.x + q.y, inv, 1e-5f)));
#pragma unroll
  for (int j = 0; j < H; ++j) out[j] = fma2(d[j], rstd2 * ldw2(g + 2 * j), ldw2(b + 2 * j));
}
// tanh of two columns from the hardware exp2 / rcp (device_utils.hpp fast_tanh)
__device__ __forceinline__ v2f tanh2(v2f x) {
  const v2f t = x * (2.0f * 1.4426950408889634f);
  const v2f e = v2f{fast_exp2(t.x), fast_exp2(t.y)} + 1.0f;
  const v2f rc = {fast_rcp(e.x), fast_rcp(e.y)};
  return fma2(bcast2(-2.0f), rc, bcast2(1.0f));
}
typedef float v4f __attribute__((ext_vector_type(4)));
// a row of F (even) real columns held as pairs -> WP floats in memory, the rest zeros.  (Non-temporal stores were tried
// to keep the written rows out of the XCD's L2: the four 16-byte pieces of a row then reach HBM as separate partial
// writes -- WRITE_SIZE doubles, profiles/r06/edge_narrow_experiments.txt.)
template <int F, int WP>
__device__ __forceinline__ void store_pairs(float *p, const v2f (&x)[F / 2]) {
#pragma unroll
  for (int j = 0; j < WP / 4; ++j) {
    const v2f a = 2 * j < F / 2 ? x[2 * j < F / 2 ? 2 * j : 0] : v2f{0.f, 0.f};
    const v2f b = 2 * j + 1 < F / 2 ? x[2 * j + 1 < F / 2 ? 2 * j + 1 : 0] : v2f{0.f, 0.f};
    *reinterpret_cast<float4 *>(p + 4 * j) = make_float4(a.x, a.y, b.x, b.y);
  }
}

}  // namespace

// ============================================================================ NodeBlock
struct NodeNarrowArgs {
  const float *__restrict__ edge;     // [S*E, FeP]
  const float *__restrict__ node_in;  // [S*N, FnP]
  float *__restrict__ node_out;       // [S*N, FnP]
  int S;
  int fn;  // the model's Fn (PADDED instantiations: FN is Fn rounded up to a multiple of four)
  Graph g;
  // c1_linear split into its node and edge parts, transposed, padded [filter|core] (PassW layout)
  const float *__restrict__ WnT, *__restrict__ WeT, *__restrict__ bias;
  const float *__restrict__ c1g, *__restrict__ c1b, *__restrict__ fing, *__restrict__ finb;
  // node_tiled_kernel: the centred copy of c1_linear and c1_norm times the gate's exp2 scales
  const float *__restrict__ WnTc, *__restrict__ WeTc, *__restrict__ biasc, *__restrict__ c1gs, *__restrict__ c1bs;
};

template <int FN, int FE, bool PADDED>
__global__ __launch_bounds__(256) void node_narrow_kernel(NodeNarrowArgs a) {
  constexpr int FnP = 16, FeP = 16;
  const cptr WnT = as_const(a.WnT), WeT = as_const(a.WeT), bias = as_const(a.bias), c1g = as_const(a.c1g),
             c1b = as_const(a.c1b), fing = as_const(a.fing), finb = as_const(a.finb);
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)a.S * a.g.N) return;
  const int s = (int)(gid / a.g.N), b = (int)(gid % a.g.N);
  float nb[FN];
  load_row<FN>(a.node_in + gid * FnP, nb);
  // node part of c1 (+ bias): the same for every edge entering b
  float base[2 * FN];
#pragma unroll
  for (int c = 0; c < 2 * FN; ++c) base[c] = bias[gcol<FN, FnP>(c)];
  gated_matvec<FN, FN, FnP, 2 * FnP, 0>(WnT, nb, base);
  float acc[FN];
#pragma unroll
  for (int k = 0; k < FN; ++k) acc[k] = 0.f;
  const int beg = a.g.in_ptr[b], end = a.g.in_ptr[b + 1];
  const float *erow0 = a.edge + (int64_t)s * a.g.E * FeP;
  for (int idx = beg; idx < end; ++idx) {  // ascending edge id == the reference's scatter order
    float x[FE];
    load_row<FE>(erow0 + (int64_t)idx * FeP, x);
    float c1[2 * FN];
#pragma unroll
    for (int c = 0; c < 2 * FN; ++c) c1[c] = base[c];
    gated_matvec<FE, FN, FnP, 2 * FnP, 0>(WeT, x, c1);
    float gt[FN];
    ln_gate_row<FN, FnP, PADDED>(c1, c1g, c1b, gt, a.fn);
#pragma unroll
    for (int k = 0; k < FN; ++k) acc[k] += gt[k];
  }
  float ln[FN], out[FN];
  ln_row1<FN, PADDED>(acc, fing, finb, ln, a.fn);
#pragma unroll
  for (int k = 0; k < FN; ++k) out[k] = fast_tanh(nb[k] + ln[k]);
  store_row<FN, FnP>(a.node_out + gid * FnP, out);
}

// ---------------------------------------------------------------------------- NodeBlock, LDS-staged rows
// node_narrow_kernel above lets every lane fetch the 64-byte rows of its atom's in-edges itself: a wave
// instruction then touches 64 different cache lines for 16 bytes each, and the kernel ends up bound by the
// vector-memory path at a third of the HBM rate.  Here ONE WAVE owns a tile of consecutive atoms (Graph::nt_*:
// at most 64 atoms, their in-edge rows a whole number of 64-row chunks as nearly as the graph allows -- seven
// atoms = 126 rows or fourteen = 252 on the benchmark cell).  With the edge rows in (b, a) order the tile's rows
// are one contiguous block of a frame, which the wave streams through its own LDS ring:
//   * LDS-DMA brings 16 whole rows per instruction (lane l: row l / 4, 16-byte piece l % 4), four instructions
//     = one 64-row chunk; slot (row, p) receives global
//     piece p ^ (row / 4 % 4), which makes the row-per-lane ds_read_b128 of 16 consecutive lanes conflict-free;
//   * ONE LANE PER ROW computes W_e edge_e + (W_n node[b] + bias) -> LayerNorm(2Fn) -> gate (centred Linear:
//     no mean) and leaves the Fn gated values in LDS (column-major: the per-atom pass reads without conflicts);
//   * one lane per atom then adds its in-edge rows in the reference's scatter order, LayerNorm(Fn), residual
//     tanh (_gnn.py:141-151).
// No workgroup barrier: rounds 3-5 ran this with four-wave workgroups, 256-row chunks and two barriers per chunk,
// which kept a CU's waves in lock step around every wait (wait_any 0.71 at a third of the HBM rate).
// chunks of a wave's ring: RN_POTGNN_NODE_RING = 1 (default: request, wait, compute) or 2 (one chunk ahead)
static int node_ring() {
  static const int d = getenv("RN_POTGNN_NODE_RING") ? (atoi(getenv("RN_POTGNN_NODE_RING")) == 2 ? 2 : 1) : 1;
  return d;
}
struct NodeTiledLds {
  size_t stage, gated, base, nb, ints, total;
};
__host__ __device__ inline NodeTiledLds node_tiled_lds(int fn, int maxD, int maxN, int ring) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  NodeTiledLds L;
  size_t off = 0;
  L.stage = off; off += (size_t)ring * 64 * 16 * 4;           // ring of [64][16] edge rows, pieces swizzled
  L.gated = off; off += up((size_t)fn * maxD * 4);            // [fn][maxD] gate outputs of the tile's in-edge rows
  L.base = off; off += up((size_t)maxN * 2 * fn * 4);         // [maxN][2 fn] W_n node[b] + bias
  L.nb = off; off += up((size_t)maxN * fn * 4);               // [maxN][fn] node[b] (the residual)
  L.ints = off; off += up((size_t)maxD);                      // tile-local atom of every row (a byte: at most 64 atoms)
  L.total = off;
  return L;
}

// What this kernel's rate depends on (profiles/r06/node_narrow_experiments.txt): with an LDS-DMA outstanding hipcc
// puts s_waitcnt vmcnt(0) in front of every LDS access, so a wave never computes under its own requests whatever
// the ring depth -- and the variants that do (requests from inline assembly, counted waits, the node rows through the
// ring as well) measured SLOWER, 0.45-0.56 ms against 0.29 per launch: more bytes in flight per wave bought less
// bandwidth, not more.  What hides the latency is the number of waves per CU, i.e. the LDS a wave needs: one
// 64-row slot (D = 1) leaves room for fourteen.
template <int FN, int FE, bool PADDED, int D>
__global__ __launch_bounds__(64) void node_tiled_kernel(NodeNarrowArgs a) {
  constexpr int FnP = 16, FeP = 16;
  const cptr WnT = as_const(a.WnTc), WeT = as_const(a.WeTc), bias = as_const(a.biasc), c1gs = as_const(a.c1gs),
             c1bs = as_const(a.c1bs), fing = as_const(a.fing), finb = as_const(a.finb);
  const float inv2n = PADDED ? 1.0f / (float)(2 * a.fn) : 1.0f / (2 * FN);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const int maxD = g.nt_max_in_rows;
  const NodeTiledLds L = node_tiled_lds(FN, maxD, g.nt_max_nodes, D);
  float *stage = reinterpret_cast<float *>(smem_raw + L.stage);
  float *gated = reinterpret_cast<float *>(smem_raw + L.gated);
  float *base = reinterpret_cast<float *>(smem_raw + L.base);
  float *nbs = reinterpret_cast<float *>(smem_raw + L.nb);
  unsigned char *d_bl = smem_raw + L.ints;
  const int lane = threadIdx.x;

  int logical = blockIdx.x;  // the tiles of one frame group read neighbouring rows: keep them on one XCD
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.nt_num;
  const int sg = logical / g.nt_num, nsg = gridDim.x / g.nt_num;
  const int j0 = g.nt_begin[tile], natoms = g.nt_begin[tile + 1] - j0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j0 + natoms] - di0;
  const int nchunks = max((dcount + 63) / 64, 1);  // (a tile without in-edges still has its atoms to finish)
  for (int i = lane; i < dcount; i += 64) d_bl[i] = (unsigned char)(g.edge_b[g.in_edge[di0 + i]] - j0);
  const int my_i0 = lane < natoms ? g.in_ptr[j0 + lane] - di0 : 0, my_i1 = lane < natoms ? g.in_ptr[j0 + lane + 1] - di0 : 0;
  if (sg >= a.S) return;
  const int nframes = (a.S - sg + nsg - 1) / nsg;  // frames of this wave
  const int nitems = nframes * nchunks;
  // chunk `item` (frame sg + (item / nchunks) nsg, rows 64 (item % nchunks) ..) -> stage slot item % D: four requests
  // (rows beyond the tile are clamped)
  auto request = [&](int item) {
    if (item >= nitems || dcount == 0) return;
    const int f = item / nchunks, k = item - f * nchunks;
    const int64_t erow0 = (int64_t)(sg + f * nsg) * g.E + di0;
    float *buf = stage + (item % D) * (64 * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 16 * j + (lane >> 2);                       // chunk-local row of this lane's piece
      const int piece = (lane & 3) ^ ((row >> 2) & 3);
      const int i = min(64 * k + row, dcount - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(a.edge + (erow0 + i) * FeP + 4 * piece),
          (__attribute__((address_space(3))) void *)(buf + (16 * j) * 16), 16, 0, 0);
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j) request(j);
  int f = 0, k = 0;
  for (int item = 0; item < nitems; ++item) {
    const int s = sg + f * nsg;
    const int64_t nrow0 = (int64_t)s * g.N;
    if (k == 0 && lane < natoms) {
      // ---- frame start: node[b] and W_n node[b] + bias of the tile's atoms (one lane per atom)
      float nb[FN], bs[2 * FN];
      load_row<FN>(a.node_in + (nrow0 + j0 + lane) * FnP, nb);
#pragma unroll
      for (int c = 0; c < 2 * FN; ++c) bs[c] = bias[gcol<FN, FnP>(c)];
      gated_matvec<FN, FN, FnP, 2 * FnP, 0>(WnT, nb, bs);
#pragma unroll
      for (int c = 0; c < 2 * FN; ++c) base[lane * 2 * FN + c] = bs[c];
#pragma unroll
      for (int c = 0; c < FN; ++c) nbs[lane * FN + c] = nb[c];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk `item` has landed (see above: nothing is left in flight anyway)
    __builtin_amdgcn_wave_barrier();
    // ---- one lane per in-edge row of the chunk
    if (const int i = 64 * k + lane; i < dcount) {
      const float *row = stage + (item % D) * (64 * 16) + lane * 16;
      const int sw = (lane >> 2) & 3;
      float x[FE];
#pragma unroll
      for (int j = 0; j < (FE + 3) / 4; ++j) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 4 * (j ^ sw));
        const float t4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * j + q < FE) x[4 * j + q] = t4[q];
      }
      float c1[2 * FN];
      const float *bs = base + (int)d_bl[i] * 2 * FN;
#pragma unroll
      for (int c = 0; c < 2 * FN; ++c) c1[c] = bs[c];
      gated_matvec<FE, FN, FnP, 2 * FnP, 0>(WeT, x, c1);
      float gt[FN];
      ln_gate_row_c<FN, FnP>(c1, c1gs, c1bs, inv2n, gt);
#pragma unroll
      for (int c = 0; c < FN; ++c) gated[c * maxD + i] = gt[c];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the chunk's rows are in registers: its slot is free
    __builtin_amdgcn_wave_barrier();
    request(item + D);
    if (k == nchunks - 1 && lane < natoms) {
      // ---- one lane per atom: sum of its in-edge rows (ascending = the reference's scatter order)
      float acc[FN];
#pragma unroll
      for (int c = 0; c < FN; ++c) acc[c] = 0.f;
      for (int i = my_i0; i < my_i1; ++i) {
#pragma unroll
        for (int c = 0; c < FN; ++c) acc[c] += gated[c * maxD + i];
      }
      float ln[FN], out[FN];
      ln_row1<FN, PADDED>(acc, fing, finb, ln, a.fn);
#pragma unroll
      for (int c = 0; c < FN; ++c) out[c] = fast_tanh(nbs[lane * FN + c] + ln[c]);
      store_row<FN, FnP>(a.node_out + (nrow0 + j0 + lane) * FnP, out);
    }
    if (++k == nchunks) {
      k = 0;
      ++f;
    }
  }
}

// (fn_kernel: the compile-time node width of the instantiation the launcher will pick)
size_t node_tiled_lds_bytes(int fn, int fe, int tile_in_rows, int tile_nodes) {
  int fn_kernel = (fn + 3) / 4 * 4;
  if ((fn == 5 && fe == 14) || (fn == 5 && fe == 5)) fn_kernel = fn;  // RN_NARROW_EXACT
  return node_tiled_lds(fn_kernel, tile_in_rows, tile_nodes, node_ring()).total;
}

// ============================================================================ EdgeBlock
struct EdgeNarrowArgs {
  const float *__restrict__ edge_in;  // [S*E, FeP]
  float *__restrict__ edge_out;
  const float *__restrict__ node;  // updated node embedding [S*N, FnP]
  int S;
  int fe;  // the model's Fe (PADDED instantiations: FE is Fe rounded up to a multiple of four)
  Graph g;
  // PassW layout (kernels.hpp), the CENTRED copies: c3_WeT_c [FeP][4FeP] = (W4 | W5), c3_WnT_c [FnP][6FeP] = (Wi | Wj | Wk),
  // c3_nshift_c [6FeP] = (0 | bias | 0), c2_WT_c [FnP][2FeP], c2_bias_c
  const float *__restrict__ c3WeT, *__restrict__ c3WnT, *__restrict__ c3shift, *__restrict__ c2WT,
      *__restrict__ c2bias;
  const float *__restrict__ c3n1gs, *__restrict__ c3n1bs;  // c3_norm_1 with the gate's exp2 scales folded in
  const float *__restrict__ c3n2g, *__restrict__ c3n2b;
  const float *__restrict__ c2n1gs, *__restrict__ c2n1bs;  // c2_norm_1 in the same form
  const float *__restrict__ c2n2g, *__restrict__ c2n2b;
};

// LDS row stride (floats) of the source rows Q': 2 FE values (FE even), a multiple of 4 with an odd number of
// 16-byte slots so that consecutive rows start on different bank groups
__host__ __device__ constexpr int narrow_ldq(int fe, bool fold = false) {  // fold: one more float, |q|^2 / 2Fe
  int s = (2 * fe + (fold ? 1 : 0) + 3) / 4 * 4;
  return ((s / 4) % 2 == 0) ? s + 4 : s;
}
struct NarrowLds {
  size_t bufQ, ints, total;
};
__host__ __device__ inline NarrowLds narrow_lds(int fe, int maxR, int maxD, bool fold = false) {
  NarrowLds L;
  L.bufQ = 0;
  L.ints = ((size_t)maxR * narrow_ldq(fe, fold) * 4 + 15) & ~size_t(15);
  L.total = L.ints + ((2 * (size_t)maxR + 5 * (size_t)maxD) * 4 + 15 & ~size_t(15));
  return L;
}

// The EdgeBlock at Fn, Fe <= 16 (_gnn.py:200-351), tile-resident like the wide kernel: the source rows
// Q'_e = W5 edge_e + Wi node[b_e] of a tile's atoms are built once per frame in LDS (one lane per row) and reused by
// every destination edge entering those atoms; then ONE LANE PER DESTINATION EDGE d = (k -> j):
//   P'_d = W4 edge_d + Wj node[j] + Wk node[k] + bias                      (centred Linears: zero row mean)
//   c2   = LayerNorm(gate(LayerNorm(c2_linear(node[j] * node[k]))))         (does not depend on the triplets: done first,
//                                                                            so only r = edge_d + c2 lives across the loop)
//   c3   = LayerNorm(sum over the out-edges e of j, e != (j -> k), of gate(LayerNorm(P'_d + Q'_e)))
//   out  = tanh(edge_d + c2 + c3)
// The triplet loop is everything: 17 iterations of 2 Fe columns on the benchmark cell.  Per triplet and lane it issues
// (Fe = 14) 7 ds_read_b128, 14 v_pk_add (z = p + q), 14 v_pk_fma (|z|^2), rsq, and per PAIR of gates 2 v_pk_mul
// (rstd gamma), 2 v_pk_fma (+ beta), 4 v_exp, 3 packed ops + 2 v_rcp for the gate and the accumulating v_pk_fma:
// ~10 instructions per gate, 3 of them transcendental, where the one-column-per-instruction form of rounds 3-5
// issued 18.4.  The variance is the plain sum of squares of z (zero mean by construction): no cross term, no |q|^2.
// CLAMP = false when the host has proven the gate arguments small (PassW::c3_fast, api.hip refresh_pass_flags).
// FOLD (opt-in, RN_POTGNN_NARROW_FOLD=1, needs CLAMP = false): the wide kernel's folded-scale loop -- c3_norm_1's scale times
// the gate's exp2 factor multiplied into P' and Q' once per row, the variance from the cross term p.q (p / gamma^2 in 2 Fe more
// registers, |q|^2 with the row): 14 packed multiplies less per triplet, 28 registers more (three waves per SIMD).
#ifndef RN_NARROW_WAVES
#define RN_NARROW_WAVES 2  // waves per SIMD the register allocation aims at least for (2: whatever fits <= 128 VGPRs = four waves)
#endif
// STAGE (RN_POTGNN_NARROW_STAGE, tiles of at most NT destinations): the finished rows go to HBM through LDS -- every lane
// parks its row in the (by then idle) Q' buffer and the workgroup stores the tile's contiguous block 16 bytes per lane,
// whole lines per instruction, NON-TEMPORALLY: the rows an EdgeBlock writes are read next by another kernel a trajectory
// chunk later, and kept out of the XCD's L2 they leave it to the rows this frame's other tiles are about to read a second
// time (every edge row is read twice per pass: as a destination row by its b atom's tile, as a source row by its a atom's).
template <int FN, int FE, bool CLAMP, bool PADDED, int NT, bool FOLD = false, bool STAGE = false>
__global__ __launch_bounds__(NT, FOLD ? 3 : RN_NARROW_WAVES) void edge_narrow_kernel(EdgeNarrowArgs a) {
  static_assert(FE % 2 == 0, "column pairs");
  static_assert(!(FOLD && CLAMP), "the folded loop has no clamp");
  constexpr int FnP = 16, FeP = 16, LDQ = narrow_ldq(FE, FOLD), H = FE / 2, W2 = 2 * FE;
  const cptr c3WeT = as_const(a.c3WeT), c3WnT = as_const(a.c3WnT), c3shift = as_const(a.c3shift),
             c2WT = as_const(a.c2WT), c2bias = as_const(a.c2bias), c3n1gs = as_const(a.c3n1gs),
             c3n1bs = as_const(a.c3n1bs), c3n2g = as_const(a.c3n2g), c3n2b = as_const(a.c3n2b),
             c2n1gs = as_const(a.c2n1gs), c2n1bs = as_const(a.c2n1bs), c2n2g = as_const(a.c2n2g),
             c2n2b = as_const(a.c2n2b);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const NarrowLds L = narrow_lds(FE, g.max_tile_out_rows, g.max_tile_in_rows, FOLD);
  float *bufQ = reinterpret_cast<float *>(smem_raw + L.bufQ);
  int *qb = reinterpret_cast<int *>(smem_raw + L.ints);
  const int maxD = g.max_tile_in_rows;
  int *qpos = qb + g.max_tile_out_rows, *d_a = qpos + g.max_tile_out_rows, *d_bl = d_a + maxD, *d_rb = d_bl + maxD,
      *d_cnt = d_rb + maxD, *d_skip = d_cnt + maxD;
  const int tid = threadIdx.x;

  int logical = blockIdx.x;  // workgroups of one frame group share node rows: keep them on one XCD
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.num_tiles;
  const int sg = logical / g.num_tiles, nsg = gridDim.x / g.num_tiles;
  const int j0 = g.tile_begin[tile], j1 = g.tile_begin[tile + 1];
  const int eo0 = g.out_ptr[j0], rows = g.out_ptr[j1] - eo0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;

  // ---- once per launch: the tile's topology -> LDS (the graph is the same in every frame)
  for (int r = tid; r < rows; r += NT) {
    qb[r] = g.edge_b[eo0 + r];
    qpos[r] = g.in_pos[eo0 + r];
  }
  for (int i = tid; i < dcount; i += NT) {
    const int dst = g.in_edge[di0 + i];
    const int ad = g.edge_a[dst], bd = g.edge_b[dst];
    const int rb = g.out_ptr[bd] - eo0, re = g.out_ptr[bd + 1] - eo0;
    const int rev = g.rev_edge[dst];  // edge (b_d -> a_d): its triplet (i == k) is excluded
    d_a[i] = ad;
    d_bl[i] = bd - j0;
    d_rb[i] = rb;
    d_cnt[i] = (re - rb) - (rev >= 0 ? 1 : 0);
    d_skip[i] = rev >= 0 ? rev - eo0 : re;
  }
  __syncthreads();

  const float inv2n = PADDED ? 1.0f / (float)(2 * a.fe) : 1.0f / W2;
  const int fe_rt = a.fe;

  for (int s = sg; s < a.S; s += nsg) {
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
    // ================= source rows Q'_e = W5 edge_e + Wi node[b_e]  (zero row mean, padded columns zero)
    for (int r = tid; r < rows; r += NT) {
      float x[FE], nb[FN];
      load_row<FE>(a.edge_in + (erow0 + qpos[r]) * FeP, x);
      load_row<FN>(a.node + (nrow0 + qb[r]) * FnP, nb);
      v2f q[2 * H];
#pragma unroll
      for (int c = 0; c < 2 * H; ++c) q[c] = v2f{0.f, 0.f};
      gated_matvec2<FE, FE, FeP, 4 * FeP, 2 * FeP>(c3WeT, x, q);
      gated_matvec2<FN, FE, FeP, 6 * FeP, 0>(c3WnT, nb, q);
      float *row = bufQ + r * LDQ;  // [filter pairs | core pairs]: 2 FE floats
      if constexpr (FOLD) {
        row[W2] = sumsq2<2 * H>(q) * inv2n;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          q[j] *= ldw2(c3n1gs + 2 * j);
          q[H + j] *= ldw2(c3n1gs + FeP + 2 * j);
        }
      }
#pragma unroll
      for (int j = 0; j < H; ++j) {
        *reinterpret_cast<v2f *>(row + 2 * j) = q[j];
        *reinterpret_cast<v2f *>(row + FE + 2 * j) = q[H + j];
      }
    }
    __syncthreads();

    // ================= destination edges: one lane each
    // (uniform trip count: the STAGE form has workgroup barriers behind the body)
    for (int i0 = 0; i0 < dcount; i0 += NT) {
      const int i = i0 + tid;
      const bool active = i < dcount;
      const int64_t drow = erow0 + di0 + i;  // (b, a) order: the tile's destination rows are contiguous
      v2f out[H];
      if (active) {
      v2f p[2 * H], r[H];  // r = edge_d + c2
      {
        float x[FE], nj[FN], nk[FN];
        load_row<FE>(a.edge_in + drow * FeP, x);
        load_row<FN>(a.node + (nrow0 + j0 + d_bl[i]) * FnP, nj);
        load_row<FN>(a.node + (nrow0 + d_a[i]) * FnP, nk);
        load_pairs<FE, FeP, 2 * FeP>(c3shift, p);
        gated_matvec2<FE, FE, FeP, 4 * FeP, 0>(c3WeT, x, p);
        gated_matvec2<FN, FE, FeP, 6 * FeP, 2 * FeP>(c3WnT, nj, p);
        gated_matvec2<FN, FE, FeP, 6 * FeP, 4 * FeP>(c3WnT, nk, p);
        // c2: gate(LayerNorm(c2_linear(node[j] * node[k]))) -> LayerNorm   (_gnn.py:223-228)
        float z[FN];
#pragma unroll
        for (int k = 0; k < FN; ++k) z[k] = nj[k] * nk[k];
        v2f c2pre[2 * H], g2p[H];
        load_pairs<FE, FeP, 0>(c2bias, c2pre);
        gated_matvec2<FN, FE, FeP, 2 * FeP, 0>(c2WT, z, c2pre);
        ln_gate_pairs<FE, FeP, true>(c2pre, c2n1gs, c2n1bs, inv2n, g2p);
        ln_pairs<FE, PADDED>(g2p, c2n2g, c2n2b, r, fe_rt);
#pragma unroll
        for (int j = 0; j < H; ++j) r[j] += v2f{x[2 * j], x[2 * j + 1]};
      }
      v2f acc[H];
#pragma unroll
      for (int j = 0; j < H; ++j) acc[j] = v2f{0.f, 0.f};
      const int cnt = d_cnt[i];
      const float *qr = bufQ + d_rb[i] * LDQ, *qskip = bufQ + d_skip[i] * LDQ;
      if constexpr (FOLD) {
        // p -> p gamma (added to the rows' q gamma), pd = p / gamma * (2 / 2Fe) (the cross term against q gamma)
        v2f pd[2 * H];
        const float spe = fmaf(sumsq2<2 * H>(p), inv2n, 1e-5f);
#pragma unroll
        for (int j = 0; j < 2 * H; ++j) {
          const v2f gam = ldw2(c3n1gs + (j < H ? 2 * j : FeP + 2 * (j - H)));
          const v2f ig = {fast_rcp(gam.x), fast_rcp(gam.y)};
          pd[j] = p[j] * ig * (2.0f * inv2n);
          if (PADDED) {  // (rounded-up columns: gamma = 0 -> keep the exact zeros)
            if (gam.x == 0.f) pd[j].x = 0.f;
            if (gam.y == 0.f) pd[j].y = 0.f;
          }
          p[j] *= gam;
        }
        for (int t = 0; t < cnt; ++t) {
          if (qr == qskip) qr += LDQ;
          v2f zz[2 * H];
          v2f d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
          for (int j = 0; j < W2 / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(qr + 4 * j);
            d0 = fma2(pd[2 * j], v2f{v.x, v.y}, d0);
            d1 = fma2(pd[2 * j + 1], v2f{v.z, v.w}, d1);
            zz[2 * j] = v2f{v.x, v.y} + p[2 * j];
            zz[2 * j + 1] = v2f{v.z, v.w} + p[2 * j + 1];
          }
          const float qq = qr[W2];
          qr += LDQ;
          d0 += d1;
          const float ve = fmaxf((d0.x + d0.y) + (spe + qq), 1e-5f);
          const v2f rstd2 = bcast2(fast_rsq(ve));
#pragma unroll
          for (int j = 0; j < H; ++j) {
            const v2f yf = fma2(zz[j], rstd2, ldw2(c3n1bs + 2 * j));
            const v2f yc = fma2(zz[H + j], rstd2, ldw2(c3n1bs + FeP + 2 * j));
            acc[j] = gate2s<false>(yf, yc, acc[j]);
          }
        }
      } else
      for (int t = 0; t < cnt; ++t) {
        if (qr == qskip) qr += LDQ;
        v2f zz[2 * H];
#pragma unroll
        for (int j = 0; j < W2 / 4; ++j) {
          const float4 v = *reinterpret_cast<const float4 *>(qr + 4 * j);
          zz[2 * j] = v2f{v.x, v.y} + p[2 * j];
          zz[2 * j + 1] = v2f{v.z, v.w} + p[2 * j + 1];
        }
        qr += LDQ;
        const v2f rstd2 = bcast2(fast_rsq(fmaf(sumsq2<2 * H>(zz), inv2n, 1e-5f)));
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const v2f yf = fma2(zz[j], rstd2 * ldw2(c3n1gs + 2 * j), ldw2(c3n1bs + 2 * j));
          const v2f yc = fma2(zz[H + j], rstd2 * ldw2(c3n1gs + FeP + 2 * j), ldw2(c3n1bs + FeP + 2 * j));
          acc[j] = gate2s<CLAMP>(yf, yc, acc[j]);
        }
      }
      v2f c3[H];
      ln_pairs<FE, PADDED>(acc, c3n2g, c3n2b, c3, fe_rt);
#pragma unroll
      for (int j = 0; j < H; ++j) out[j] = tanh2(r[j] + c3[j]);
      }  // active
      bool staged = false;
      if constexpr (STAGE) {
        if (dcount <= NT) {    // (uniform: one destination per lane, nobody comes back to the Q' rows)
          staged = true;
          __syncthreads();     // every lane's triplet loop is done
          if (active) store_pairs<FE, FeP>(bufQ + i * FeP, out);
          __syncthreads();     // the tile's rows are parked
          float *dstblk = a.edge_out + (erow0 + di0) * FeP;
          for (int idx = tid; idx < dcount * (FeP / 4); idx += NT)
            __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(bufQ + 4 * idx), reinterpret_cast<v4f *>(dstblk + 4 * idx));
        }
      }
      if (!staged && active) store_pairs<FE, FeP>(a.edge_out + drow * FeP, out);
    }
    __syncthreads();  // bufQ may be rewritten
  }
}

// ============================================================================ readout
struct ReadoutNarrowArgs {
  const float *__restrict__ edge;   // [S*E, FeP]
  const float *__restrict__ unit4;  // [S*E, 4]
  int S;
  Graph g;
  ReadoutW<float> w;  // HP = 32
  const double *__restrict__ mean9, *__restrict__ std9;
  float *__restrict__ vec6;
  double *__restrict__ alpha, *__restrict__ alpha_raw;
  float *__restrict__ pol;  // optional [S*E, 32] (stage snapshots), or null
};

template <int FE>
__global__ __launch_bounds__(256) void readout_narrow_kernel(ReadoutNarrowArgs a) {
  constexpr int FeP = 16, HP = 32;
  const cptr W0T = as_const(a.w.W0T), scale0 = as_const(a.w.scale0), shift0 = as_const(a.w.shift0),
             W3T = as_const(a.w.W3T), b3 = as_const(a.w.b3), W5T = as_const(a.w.W5T), b5 = as_const(a.w.b5);
  __shared__ float red[4][6];
  __shared__ float fin[6];
  const int s = blockIdx.x;
  float acc[6] = {0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < a.g.E; e += 256) {
    const int64_t row = (int64_t)s * a.g.E + e;
    float x[FE], h1[FE], h2[FE], m[12];
    load_row<FE>(a.edge + row * FeP, x);
#pragma unroll
    for (int c = 0; c < FE; ++c) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < FE; ++k) v = fmaf(W0T[k * HP + c], x[k], v);
      h1[c] = ssp_fast(v * scale0[c] + shift0[c]);
    }
#pragma unroll
    for (int c = 0; c < FE; ++c) {
      float v = b3[c];
#pragma unroll
      for (int k = 0; k < FE; ++k) v = fmaf(W3T[k * HP + c], h1[k], v);
      h2[c] = ssp_fast(v);
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      float v = b5[c];
#pragma unroll
      for (int k = 0; k < FE; ++k) v = fmaf(W5T[k * 32 + c], h2[k], v);
      m[c] = v;
    }
    if (a.pol) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        *reinterpret_cast<float4 *>(a.pol + row * 32 + 4 * j) = make_float4(m[4 * j], m[4 * j + 1], m[4 * j + 2], m[4 * j + 3]);
    }
    const float4 u = *reinterpret_cast<const float4 *>(a.unit4 + row * 4);
    // closed form of R diag(p,q,q) R^-1 = q I + (p - q) u u^T, one masked component per pair
    acc[3] += (m[0] - m[1]) * (u.x * u.y);            // xy <- emb 0,1
    acc[4] += (m[2] - m[3]) * (u.x * u.z);            // xz <- emb 2,3
    acc[5] += (m[4] - m[5]) * (u.y * u.z);            // yz <- emb 4,5
    acc[0] += m[7] + (m[6] - m[7]) * (u.x * u.x);     // xx <- emb 6,7
    acc[1] += m[9] + (m[8] - m[9]) * (u.y * u.y);     // yy <- emb 8,9
    acc[2] += m[11] + (m[10] - m[11]) * (u.z * u.z);  // zz <- emb 10,11
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    v = v / (float)a.g.E;
    fin[threadIdx.x] = v;
    if (a.vec6) a.vec6[(int64_t)s * 6 + threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int map[9] = {0, 3, 4, 3, 1, 5, 4, 5, 2};  // dataset/torch/utils.py:30-37
    const double v = (double)fin[map[threadIdx.x]];
    if (a.alpha) a.alpha[(int64_t)s * 9 + threadIdx.x] = v * a.std9[threadIdx.x] + a.mean9[threadIdx.x];
    if (a.alpha_raw) a.alpha_raw[(int64_t)s * 9 + threadIdx.x] = v;
  }
}

// ---------------------------------------------------------------------------- readout on the matrix pipe
// The readout is the one dense MLP of the narrow pipeline whose rows are independent and plentiful (E per frame): three
// [E,16] x [16,16] products.  readout_narrow_kernel above does them as 560 scalar FMAs per row and is VALU-bound (0.91
// busy); here they go to the matrix pipe as EXACT float32 products (v_mfma_f32_16x16x4_f32: an fmaf chain, bit for bit),
// which leaves the VALU the two ShiftedSoftplus epilogues and the tensor arithmetic.
// Every layer is computed TRANSPOSED, H_next^T = W^T H^T, so that the result layout of one product is the operand layout of
// the next and nothing is transposed in between: for a tile of 16 rows, lane (r = l % 16, q = l / 16) holds columns
// 4q .. 4q + 3 of row r of every activation -- on input one 16-byte piece of the edge row as it sits in HBM; as the MFMA's
// B operand its element kk is H[r][4q + kk]; the A operand is W[4q + kk][r] (twelve registers, loaded once); the result
// D[4q + i][r] is H_next[r][4q + i].  The last layer's twelve outputs of a row therefore sit in lanes q = 0, 1, 2 of the row,
// two tensor components each (closed form of R diag(p,q,q) R^-1, _gnn.py:372-415), summed per q over the 16-lane row.
template <bool UNUSED = false>
__global__ __launch_bounds__(256) void readout_narrow_mfma_kernel(ReadoutNarrowArgs a) {
  constexpr int FeP = 16, HP = 32;
  __shared__ float red[4][3][2];
  __shared__ float fin[6];
  const int s = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  // A operands: W[4q + kk][r] of the three layers (W0T [FeP][HP], W3T [HP][HP], W5T [HP][32]: k-major)
  float w0[4], w3[4], w5[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    w0[kk] = a.w.W0T[(4 * q + kk) * HP + r];
    w3[kk] = a.w.W3T[(4 * q + kk) * HP + r];
    w5[kk] = a.w.W5T[(4 * q + kk) * 32 + r];
  }
  // the epilogues' per-column constants for this lane's four columns 4q + i
  f32x4_t sc0, sh0, bb3, bb5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc0[i] = a.w.scale0[4 * q + i];
    sh0[i] = a.w.shift0[4 * q + i];
    bb3[i] = a.w.b3[4 * q + i];
    bb5[i] = a.w.b5[4 * q + i];
  }
  const int E = a.g.E, ntiles = (E + 15) / 16;
  float a0 = 0.f, a1 = 0.f;
  for (int t = wv; t < ntiles; t += 4) {
    const int e = min(16 * t + r, E - 1);
    const bool valid = 16 * t + r < E;
    const int64_t row = (int64_t)s * E + e;
    const f32x4_t x = *reinterpret_cast<const f32x4_t *>(a.edge + row * FeP + 4 * q);
    const float4 u = *reinterpret_cast<const float4 *>(a.unit4 + row * 4);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[kk], x[kk], acc, 0, 0, 0);
    f32x4_t h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = ssp_fast(fmaf(acc[i], sc0[i], sh0[i]));
    acc = bb3;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[kk], h[kk], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = ssp_fast(acc[i]);
    acc = bb5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w5[kk], h[kk], acc, 0, 0, 0);
    if (a.pol && valid && q < 3) *reinterpret_cast<f32x4_t *>(a.pol + row * 32 + 4 * q) = acc;
    // q = 0: xy <- (m0 - m1) ux uy, xz <- (m2 - m3) ux uz;  q = 1: yz <- (m4 - m5) uy uz, xx <- m7 + (m6 - m7) ux ux;
    // q = 2: yy <- m9 + (m8 - m9) uy uy, zz <- m11 + (m10 - m11) uz uz
    const float c0 = q == 0 ? u.x * u.y : (q == 1 ? u.y * u.z : u.y * u.y);
    const float c1 = q == 0 ? u.x * u.z : (q == 1 ? u.x * u.x : u.z * u.z);
    const float k0 = q == 2 ? 1.f : 0.f, k1 = (q == 1 || q == 2) ? 1.f : 0.f;
    const float t0 = fmaf(acc[0] - acc[1], c0, k0 * acc[1]), t1 = fmaf(acc[2] - acc[3], c1, k1 * acc[3]);
    if (valid && q < 3) {
      a0 += t0;
      a1 += t1;
    }
  }
  a0 = lg_sum<16>(a0);  // over the 16 rows of this lane's q (one DPP row)
  a1 = lg_sum<16>(a1);
  if (r == 0 && q < 3) {
    red[wv][q][0] = a0;
    red[wv][q][1] = a1;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int qq[6] = {1, 2, 2, 0, 0, 1}, slot[6] = {1, 0, 1, 0, 1, 0};  // component (xx,yy,zz,xy,xz,yz) -> (q, which accumulator)
    const int c = threadIdx.x;
    float v = (red[0][qq[c]][slot[c]] + red[1][qq[c]][slot[c]]) + (red[2][qq[c]][slot[c]] + red[3][qq[c]][slot[c]]);
    v = v / (float)E;
    fin[c] = v;
    if (a.vec6) a.vec6[(int64_t)s * 6 + c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int map[9] = {0, 3, 4, 3, 1, 5, 4, 5, 2};  // dataset/torch/utils.py:30-37
    const double v = (double)fin[map[threadIdx.x]];
    if (a.alpha) a.alpha[(int64_t)s * 9 + threadIdx.x] = v * a.std9[threadIdx.x] + a.mean9[threadIdx.x];
    if (a.alpha_raw) a.alpha_raw[(int64_t)s * 9 + threadIdx.x] = v;
  }
}

// ============================================================================ launchers
// Exact instantiations: the documented widths and those of the reference's own tests.  Everything else with
// Fn, Fe <= 16 runs on the PADDED instantiation of its widths rounded up to multiples of four.
#define RN_NARROW_EXACT(X) X(5, 14) X(5, 5)
#define RN_NARROW_EXACT_EDGE(X) X(5, 14)  // (the EdgeBlock works on column pairs: Fe = 5 runs as 8 rounded up)
#define RN_NARROW_GRID(X) \
  X(4, 4) X(4, 8) X(4, 12) X(4, 16) X(8, 4) X(8, 8) X(8, 12) X(8, 16) X(12, 4) X(12, 8) X(12, 12) X(12, 16) \
  X(16, 4) X(16, 8) X(16, 12) X(16, 16)

bool narrow_supported(Dims d) { return d.FnP == 16 && d.FeP == 16 && d.Fn >= 1 && d.Fe >= 1; }

// (fe_kernel: the compile-time edge width of the instantiation the launcher will pick -- Fe = 14 has its own)
size_t edge_narrow_lds_bytes(int fn, int fe, int tile_out_rows, int tile_in_rows) {
  const int fe_kernel = (fn == 5 && fe == 14) ? 14 : (fe + 3) / 4 * 4;  // RN_NARROW_EXACT_EDGE
  return narrow_lds(fe_kernel, tile_out_rows, tile_in_rows).total;
}

template <int FN, int FE, bool PADDED, int D>
static void launch_node_tiled_ring(const NodeNarrowArgs &a, hipStream_t st) {
  auto kern = &node_tiled_kernel<FN, FE, PADDED, D>;
  const size_t lds = node_tiled_lds(FN, a.g.nt_max_in_rows, a.g.nt_max_nodes, D).total;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  static const int cap = getenv("RN_POTGNN_NODE_WAVES") ? std::max(1, atoi(getenv("RN_POTGNN_NODE_WAVES"))) : 16;
  per_cu = std::min(per_cu, cap);  // one-wave workgroups
  int nsg = per_cu * cus / a.g.nt_num;
  nsg = nsg < 1 ? 1 : (nsg > a.S ? a.S : nsg);
  kern<<<(unsigned)nsg * (unsigned)a.g.nt_num, 64, lds, st>>>(a);
}
template <int FN, int FE, bool PADDED>
static void launch_node_tiled_cfg(const NodeNarrowArgs &a, hipStream_t st) {
  if (node_ring() == 2) launch_node_tiled_ring<FN, FE, PADDED, 2>(a, st);
  else launch_node_tiled_ring<FN, FE, PADDED, 1>(a, st);
}
static void launch_node_tiled(const NodeNarrowArgs &a, Dims d, hipStream_t st) {
#define X(FN, FE) \
  if (d.Fn == FN && d.Fe == FE) return launch_node_tiled_cfg<FN, FE, false>(a, st);
  RN_NARROW_EXACT(X)
#undef X
  const int fnc = (d.Fn + 3) / 4 * 4, fec = (d.Fe + 3) / 4 * 4;
#define X(FN, FE) \
  if (fnc == FN && fec == FE) return launch_node_tiled_cfg<FN, FE, true>(a, st);
  RN_NARROW_GRID(X)
#undef X
}

void launch_node_narrow(const float *edge, const float *node_in, float *node_out, int S, const Graph &g, Dims d,
                        const PassW<float> &w, hipStream_t st) {
  if (S == 0 || g.N == 0) return;
  NodeNarrowArgs a{edge, node_in, node_out, S, d.Fn, g, w.c1_WnT, w.c1_WeT, w.c1_bias,
                   w.c1_norm.g, w.c1_norm.b, w.final_norm.g, w.final_norm.b,
                   w.c1_WnT_c, w.c1_WeT_c, w.c1_bias_c, w.c1_norm_s.g, w.c1_norm_s.b};
  static const bool tiled_ok = !(getenv("RN_POTGNN_NODE_TILED") && atoi(getenv("RN_POTGNN_NODE_TILED")) == 0);
  if (tiled_ok && g.nt_num > 0 && g.nt_narrow) {
    launch_node_tiled(a, d, st);
    return;
  }
  const unsigned blocks = (unsigned)(((int64_t)S * g.N + 255) / 256);
#define X(FN, FE) \
  if (d.Fn == FN && d.Fe == FE) return (void)(node_narrow_kernel<FN, FE, false><<<blocks, 256, 0, st>>>(a));
  RN_NARROW_EXACT(X)
#undef X
  const int fnc = (d.Fn + 3) / 4 * 4, fec = (d.Fe + 3) / 4 * 4;
#define X(FN, FE) \
  if (fnc == FN && fec == FE) return (void)(node_narrow_kernel<FN, FE, true><<<blocks, 256, 0, st>>>(a));
  RN_NARROW_GRID(X)
#undef X
}

template <int FN, int FE, bool CLAMP, bool PADDED, int NT, bool FOLD = false, bool STAGE = false>
static void launch_edge_cfg_nt(const EdgeNarrowArgs &a, hipStream_t st) {
  auto kern = &edge_narrow_kernel<FN, FE, CLAMP, PADDED, NT, FOLD, STAGE>;
  const size_t lds = narrow_lds(FE, a.g.max_tile_out_rows, a.g.max_tile_in_rows, FOLD).total;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, (FOLD ? 12 : (RN_NARROW_WAVES >= 5 ? 20 : 16)) * 64 / NT);  // (four waves per SIMD; three with the folded loop's registers)
  static const int wg_cap = getenv("RN_POTGNN_NARROW_WGS") ? std::max(1, atoi(getenv("RN_POTGNN_NARROW_WGS"))) : 0;  // experiment knob: workgroups per CU
  if (wg_cap > 0) per_cu = std::min(per_cu, wg_cap);
  int nsg = per_cu * cus / a.g.num_tiles;
  nsg = nsg < 1 ? 1 : (nsg > a.S ? a.S : nsg);
  kern<<<(unsigned)nsg * (unsigned)a.g.num_tiles, NT, lds, st>>>(a);
}
// Two-wave workgroups when every tile has at most 128 destination edges and 128 source rows (api.hip sizes the tiles
// so: seven atoms of the benchmark cell = 126 rows), four waves otherwise: the more and smaller workgroups a CU holds,
// the less a barrier between the source-row stage and the destination stage idles it.
template <int FN, int FE, bool CLAMP, bool PADDED>
static void launch_edge_cfg(const EdgeNarrowArgs &a, hipStream_t st) {
  if constexpr (!CLAMP && !PADDED) {
    static const bool fold = getenv("RN_POTGNN_NARROW_FOLD") && atoi(getenv("RN_POTGNN_NARROW_FOLD")) != 0;
    if (fold) {
      if (a.g.max_tile_in_rows <= 128 && a.g.max_tile_out_rows <= 128) return launch_edge_cfg_nt<FN, FE, false, false, 128, true>(a, st);
      return launch_edge_cfg_nt<FN, FE, false, false, 256, true>(a, st);
    }
  }
  static const bool stage = getenv("RN_POTGNN_NARROW_STAGE") && atoi(getenv("RN_POTGNN_NARROW_STAGE")) != 0;
  if (a.g.max_tile_in_rows <= 128 && a.g.max_tile_out_rows <= 128) {
    if (stage) launch_edge_cfg_nt<FN, FE, CLAMP, PADDED, 128, false, true>(a, st);
    else launch_edge_cfg_nt<FN, FE, CLAMP, PADDED, 128>(a, st);
  } else {
    launch_edge_cfg_nt<FN, FE, CLAMP, PADDED, 256>(a, st);
  }
}

void launch_edge_narrow(const float *edge_in, float *edge_out, const float *node, int S, const Graph &g, Dims d,
                        const PassW<float> &w, hipStream_t st) {
  if (S == 0 || g.E == 0) return;
  EdgeNarrowArgs a{edge_in, edge_out, node, S, d.Fe, g, w.c3_WeT_c, w.c3_WnT_c, w.c3_nshift_c, w.c2_WT_c, w.c2_bias_c,
                   w.c3_norm_1s.g, w.c3_norm_1s.b, w.c3_norm_2.g, w.c3_norm_2.b,
                   w.c2_norm_1s.g, w.c2_norm_1s.b, w.c2_norm_2.g, w.c2_norm_2.b};
  // the gate's overflow clamp stays in the loop unless the host has bounded the gate arguments (refresh_pass_flags)
  const bool clamp = (w.c3_fast & 1) == 0;
#define X(FN, FE)                                                 \
  if (d.Fn == FN && d.Fe == FE) {                                 \
    if (clamp) launch_edge_cfg<FN, FE, true, false>(a, st);       \
    else launch_edge_cfg<FN, FE, false, false>(a, st);            \
    return;                                                       \
  }
  RN_NARROW_EXACT_EDGE(X)
#undef X
  const int fnc = (d.Fn + 3) / 4 * 4, fec = (d.Fe + 3) / 4 * 4;
#define X(FN, FE)                                                \
  if (fnc == FN && fec == FE) {                                  \
    if (clamp) launch_edge_cfg<FN, FE, true, true>(a, st);       \
    else launch_edge_cfg<FN, FE, false, true>(a, st);            \
    return;                                                      \
  }
  RN_NARROW_GRID(X)
#undef X
}

void launch_readout_narrow(const float *edge, const float *unit4, int S, const Graph &g, Dims d,
                           const ReadoutW<float> &w, const double *mean9, const double *std9, float *vec6,
                           double *alpha, double *alpha_raw, float *pol, hipStream_t st) {
  if (S == 0) return;
  ReadoutNarrowArgs a{edge, unit4, S, g, w, mean9, std9, vec6, alpha, alpha_raw, pol};
  // the three layers on the exact-float32 MFMA (one kernel for every Fe <= 16: rounded-up rows and columns are exact zeros
  // -- zero weights, BatchNorm fold and biases, ssp(0) = 0); RN_POTGNN_READOUT_MFMA=0: the scalar-FMA kernel
  static const bool mfma = !(getenv("RN_POTGNN_READOUT_MFMA") && atoi(getenv("RN_POTGNN_READOUT_MFMA")) == 0);
  if (mfma) return (void)(readout_narrow_mfma_kernel<><<<S, 256, 0, st>>>(a));
  // (rounded-up columns are exact zeros all the way: zero weights, BatchNorm fold and biases, ssp(0) = 0)
  if (d.Fe == 14) return (void)(readout_narrow_kernel<14><<<S, 256, 0, st>>>(a));
  if (d.Fe == 5) return (void)(readout_narrow_kernel<5><<<S, 256, 0, st>>>(a));
  switch ((d.Fe + 3) / 4 * 4) {
    case 4: return (void)(readout_narrow_kernel<4><<<S, 256, 0, st>>>(a));
    case 8: return (void)(readout_narrow_kernel<8><<<S, 256, 0, st>>>(a));
    case 12: return (void)(readout_narrow_kernel<12><<<S, 256, 0, st>>>(a));
    default: return (void)(readout_narrow_kernel<16><<<S, 256, 0, st>>>(a));
  }
}

}  // namespace rn
