// Narrow-width PotGNN blocks for gfx950: one LANE per row instead of a lane group per row.
//
// The reference documents ONE set of hyper-parameters (Fn = 5, Fe = 14, P = 4:
// docs/source/notebooks/machine-learning.ipynb:187-192).  At such widths the 64-wide kernels
// (kernels_fused.hip / kernels_agg.hip) pad every row to 16 columns per half, spend DPP butterflies
// on 4-lane groups and move the padding through HBM.  Here a whole row lives in ONE lane's registers:
//   * no padding (compile-time Fn, Fe), no cross-lane reductions at all;
//   * every weight is wave-uniform with a compile-time index, so the compiler keeps the weights in
//     SGPRs (s_load_dwordx16 batches) and the projections are plain v_fma chains -- at K = 5..16 an
//     MFMA tile would be mostly padding;
//   * NodeBlock (_gnn.py:122-151): one lane per (frame, atom) walks the atom's in-edges in the
//     reference's scatter order -- a streaming gather of 64-byte edge rows, no LDS, no barrier;
//   * EdgeBlock (_gnn.py:200-351): tile-resident like the wide kernel (the source rows Q' of a
//     tile's atoms are built once per frame in LDS and reused by every destination edge entering
//     them), one lane per destination edge, two barriers per frame;
//   * readout MLP + edge tensors + per-structure mean (_gnn.py:532-539, 354-415, 658-665) in one
//     launch, one workgroup per frame, fixed summation order (deterministic).
// Embeddings keep the padded [rows][16] float layout of the other kernels (geometry, snapshots and
// the reverse pass share the buffers); padded columns are written as exact zeros.
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
    load_row<FE>(erow0 + (int64_t)a.g.in_edge[idx] * FeP, x);
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
// vector-memory path at a third of the HBM rate.  Here a workgroup owns a tile of atoms (Graph::nt_*) and
// streams the tile's in-edge rows through LDS in chunks of 256:
//   * LDS-DMA brings 16 whole rows per wave instruction (lane l: row l / 4, 16-byte piece l % 4), two
//     chunks ahead of the arithmetic, double-buffered; slot (row, p) receives global piece p ^ (row / 4 % 4),
//     which makes the row-per-lane ds_read_b128 of 16 consecutive lanes conflict-free;
//   * ONE LANE PER ROW computes W_e edge_e + (W_n node[b] + bias) -> LayerNorm(2Fn) -> gate and leaves the
//     Fn gated values in LDS (column-major, so that the per-atom pass reads without conflicts);
//   * one lane per atom then adds its in-edge rows in the reference's scatter order, LayerNorm(Fn),
//     residual tanh (_gnn.py:141-151).
// A counted vmcnt keeps the youngest chunk's requests in flight across the barrier.
struct NodeTiledLds {
  size_t stage, gated, base, ints, total;
};
__host__ __device__ inline NodeTiledLds node_tiled_lds(int fn, int maxD, int maxN) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  NodeTiledLds L;
  size_t off = 0;
  L.stage = off; off += (size_t)2 * 256 * 16 * 4;             // 2 x [256][16] edge rows, pieces swizzled
  L.gated = off; off += up((size_t)fn * maxD * 4);            // [fn][maxD] gate outputs of the tile's in-edge rows
  L.base = off; off += up((size_t)maxN * 2 * fn * 4);         // [maxN][2 fn] W_n node[b] + bias
  L.ints = off; off += up((size_t)2 * maxD * 4);              // global edge id, tile-local atom of every row
  L.total = off;
  return L;
}

template <int FN, int FE, bool PADDED>
__global__ __launch_bounds__(256) void node_tiled_kernel(NodeNarrowArgs a) {
  constexpr int FnP = 16, FeP = 16;
  const cptr WnT = as_const(a.WnT), WeT = as_const(a.WeT), bias = as_const(a.bias), c1g = as_const(a.c1g),
             c1b = as_const(a.c1b), fing = as_const(a.fing), finb = as_const(a.finb);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const int maxD = g.nt_max_in_rows;
  const NodeTiledLds L = node_tiled_lds(FN, maxD, g.nt_max_nodes);
  float *stage = reinterpret_cast<float *>(smem_raw + L.stage);
  float *gated = reinterpret_cast<float *>(smem_raw + L.gated);
  float *base = reinterpret_cast<float *>(smem_raw + L.base);
  int *d_edge = reinterpret_cast<int *>(smem_raw + L.ints), *d_bl = d_edge + maxD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int logical = blockIdx.x;  // workgroups of one frame group share node rows: keep them on one XCD
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.nt_num;
  const int sg = logical / g.nt_num, nsg = gridDim.x / g.nt_num;
  const int j0 = g.nt_begin[tile], natoms = g.nt_begin[tile + 1] - j0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j0 + natoms] - di0;
  const int nchunks = max((dcount + 255) / 256, 1);  // (a tile without in-edges still has its atoms to finish)
  for (int i = tid; i < dcount; i += 256) {
    const int e = g.in_edge[di0 + i];
    d_edge[i] = e;
    d_bl[i] = g.edge_b[e] - j0;
  }
  __syncthreads();
  if (sg >= a.S) return;
  const int nframes = (a.S - sg + nsg - 1) / nsg;  // frames of this workgroup
  const int nitems = nframes * nchunks;
  // chunk `item` (frame sg + (item / nchunks) nsg, rows 256 (item % nchunks) ..) -> stage buffer item & 1:
  // every wave issues exactly four requests (rows beyond the tile are clamped), so vmcnt can count them
  auto request = [&](int item) {
    if (item >= nitems || dcount == 0) return;
    const int f = item / nchunks, k = item - f * nchunks;
    const int64_t erow0 = (int64_t)(sg + f * nsg) * g.E;
    float *buf = stage + (item & 1) * (256 * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 64 * wave + 16 * j + (lane >> 2);           // chunk-local row of this lane's piece
      const int piece = (lane & 3) ^ ((row >> 2) & 3);
      const int i = min(256 * k + row, dcount - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(a.edge + (erow0 + d_edge[i]) * FeP + 4 * piece),
          (__attribute__((address_space(3))) void *)(buf + (64 * wave + 16 * j) * 16), 16, 0, 0);
    }
  };
  request(0);
  request(1);
  for (int item = 0; item < nitems; ++item) {
    const int f = item / nchunks, k = item - f * nchunks;
    const int s = sg + f * nsg;
    const int64_t nrow0 = (int64_t)s * g.N;
    if (k == 0) {
      // ---- frame start: W_n node[b] + bias of the tile's atoms (one lane per atom)
      if (tid < natoms) {
        float nb[FN], bs[2 * FN];
        load_row<FN>(a.node_in + (nrow0 + j0 + tid) * FnP, nb);
#pragma unroll
        for (int c = 0; c < 2 * FN; ++c) bs[c] = bias[gcol<FN, FnP>(c)];
        gated_matvec<FN, FN, FnP, 2 * FnP, 0>(WnT, nb, bs);
#pragma unroll
        for (int c = 0; c < 2 * FN; ++c) base[tid * 2 * FN + c] = bs[c];
      }
    }
    // chunk `item` must have landed; the four requests of chunk item + 1 (issued one chunk ago) may stay in
    // flight -- unless other operations are outstanding behind them (frame start: the previous frame's loads
    // and stores) or there is no chunk item + 1
    if (k == 0 || item + 1 >= nitems) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();  // chunk `item` landed for every wave; base / gated of the previous frame no longer read
    // ---- one lane per in-edge row of the chunk
    if (const int i = 256 * k + tid; i < dcount) {
      const float *row = stage + (item & 1) * (256 * 16) + tid * 16;
      const int sw = (tid >> 2) & 3;
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
      const float *bs = base + d_bl[i] * 2 * FN;
#pragma unroll
      for (int c = 0; c < 2 * FN; ++c) c1[c] = bs[c];
      gated_matvec<FE, FN, FnP, 2 * FnP, 0>(WeT, x, c1);
      float gt[FN];
      ln_gate_row<FN, FnP, PADDED>(c1, c1g, c1b, gt, a.fn);
#pragma unroll
      for (int c = 0; c < FN; ++c) gated[c * maxD + i] = gt[c];
    }
    __syncthreads();  // the chunk's stage buffer is free; after the last chunk: gated complete
    request(item + 2);
    if (k == nchunks - 1 && tid < natoms) {
      // ---- one lane per atom: sum of its in-edge rows (ascending = the reference's scatter order)
      const int b = j0 + tid;
      const int i0 = g.in_ptr[b] - di0, i1 = g.in_ptr[b + 1] - di0;
      float acc[FN];
#pragma unroll
      for (int c = 0; c < FN; ++c) acc[c] = 0.f;
      for (int i = i0; i < i1; ++i) {
#pragma unroll
        for (int c = 0; c < FN; ++c) acc[c] += gated[c * maxD + i];
      }
      float nb[FN], ln[FN], out[FN];
      load_row<FN>(a.node_in + (nrow0 + b) * FnP, nb);
      ln_row1<FN, PADDED>(acc, fing, finb, ln, a.fn);
#pragma unroll
      for (int c = 0; c < FN; ++c) out[c] = fast_tanh(nb[c] + ln[c]);
      store_row<FN, FnP>(a.node_out + (nrow0 + b) * FnP, out);
    }
  }
}

// (fn_kernel: the compile-time node width of the instantiation the launcher will pick)
size_t node_tiled_lds_bytes(int fn, int fe, int tile_in_rows, int tile_nodes) {
  int fn_kernel = (fn + 3) / 4 * 4;
  if ((fn == 5 && fe == 14) || (fn == 5 && fe == 5)) fn_kernel = fn;  // RN_NARROW_EXACT
  return node_tiled_lds(fn_kernel, tile_in_rows, tile_nodes).total;
}

// ============================================================================ EdgeBlock
struct EdgeNarrowArgs {
  const float *__restrict__ edge_in;  // [S*E, FeP]
  float *__restrict__ edge_out;
  const float *__restrict__ node;  // updated node embedding [S*N, FnP]
  int S;
  int fe;  // the model's Fe (PADDED instantiations: FE is Fe rounded up to a multiple of four)
  Graph g;
  // PassW layout (kernels.hpp): c3_WeT [FeP][4FeP] = (W4 | W5), c3_WnT [FnP][6FeP] = (Wi | Wj | Wk),
  // c3_nshift [6FeP] = (0 | bias | 0), c2_WT [FnP][2FeP]
  const float *__restrict__ c3WeT, *__restrict__ c3WnT, *__restrict__ c3shift, *__restrict__ c2WT,
      *__restrict__ c2bias;
  const float *__restrict__ c3n1gs, *__restrict__ c3n1bs;  // c3_norm_1 with the gate's exp2 scales folded in
  const float *__restrict__ c3n2g, *__restrict__ c3n2b;
  const float *__restrict__ c2n1g, *__restrict__ c2n1b, *__restrict__ c2n2g, *__restrict__ c2n2b;
};

// LDS row stride (floats) of the centred source rows: 2 FE values + |q|^2, a multiple of 4 with an
// odd number of 16-byte slots so that consecutive rows start on different bank groups
__host__ __device__ constexpr int narrow_ldq(int fe) {
  int s = (2 * fe + 1 + 3) / 4 * 4;
  return ((s / 4) % 2 == 0) ? s + 4 : s;
}
struct NarrowLds {
  size_t bufQ, ints, total;
};
__host__ __device__ inline NarrowLds narrow_lds(int fe, int maxR, int maxD) {
  NarrowLds L;
  L.bufQ = 0;
  L.ints = ((size_t)maxR * narrow_ldq(fe) * 4 + 15) & ~size_t(15);
  L.total = L.ints + (((size_t)maxR + 6 * (size_t)maxD) * 4 + 15 & ~size_t(15));
  return L;
}

// (two workgroups per SIMD keep 128 VGPRs each: enough for the exact instantiations up to 2 FE = 28 values per
//  row; the rounded-up ones -- whose weights overflow the 102 SGPRs into VGPR lanes -- and FE = 16 take one
//  workgroup per SIMD instead of spilling to scratch)
template <int FN, int FE, bool FASTG, bool PADDED>
__global__ __launch_bounds__(256, ((FE <= 14 && !PADDED) || FE <= 4 ? 2 : 1)) void edge_narrow_kernel(EdgeNarrowArgs a) {
  constexpr int FnP = 16, FeP = 16, LDQ = narrow_ldq(FE), W2 = 2 * FE;
  const cptr c3WeT = as_const(a.c3WeT), c3WnT = as_const(a.c3WnT), c3shift = as_const(a.c3shift),
             c2WT = as_const(a.c2WT), c2bias = as_const(a.c2bias), c3n1gs = as_const(a.c3n1gs),
             c3n1bs = as_const(a.c3n1bs), c3n2g = as_const(a.c3n2g), c3n2b = as_const(a.c3n2b),
             c2n1g = as_const(a.c2n1g), c2n1b = as_const(a.c2n1b), c2n2g = as_const(a.c2n2g),
             c2n2b = as_const(a.c2n2b);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const NarrowLds L = narrow_lds(FE, g.max_tile_out_rows, g.max_tile_in_rows);
  float *bufQ = reinterpret_cast<float *>(smem_raw + L.bufQ);
  int *qb = reinterpret_cast<int *>(smem_raw + L.ints);
  const int maxD = g.max_tile_in_rows;
  int *d_edge = qb + g.max_tile_out_rows, *d_a = d_edge + maxD, *d_bl = d_a + maxD, *d_rb = d_bl + maxD,
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
  for (int r = tid; r < rows; r += 256) qb[r] = g.edge_b[eo0 + r];
  for (int i = tid; i < dcount; i += 256) {
    const int dst = g.in_edge[di0 + i];
    const int ad = g.edge_a[dst], bd = g.edge_b[dst];
    const int rb = g.out_ptr[bd] - eo0, re = g.out_ptr[bd + 1] - eo0;
    const int rev = g.rev_edge[dst];  // edge (b_d -> a_d): its triplet (i == k) is excluded
    d_edge[i] = dst;
    d_a[i] = ad;
    d_bl[i] = bd - j0;
    d_rb[i] = rb;
    d_cnt[i] = (re - rb) - (rev >= 0 ? 1 : 0);
    d_skip[i] = rev >= 0 ? rev - eo0 : re;
  }
  __syncthreads();

  // c3_norm_1 with the exp2 scale of the gate folded in (wave-uniform values: SGPRs)
  auto g3 = [&](int c) { return c3n1gs[gcol<FE, FeP>(c)]; };
  auto b3 = [&](int c) { return c3n1bs[gcol<FE, FeP>(c)]; };
  const float inv2n = PADDED ? 1.0f / (float)(2 * a.fe) : 1.0f / W2;
  // PADDED: is column c of a [filter | core] row one of the rounded-up ones?  (only the last three of a half can be)
  const int fe_rt = a.fe;
  auto pad_col = [fe_rt](int c) { return PADDED && (c % FE) >= FE - 3 && (c % FE) >= fe_rt; };

  for (int s = sg; s < a.S; s += nsg) {
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
    // ================= source rows Q'_e = W5 edge_e + Wi node[b_e], centred, (x gamma), |q|^2
#if RN_NARROW_PRIO
    __builtin_amdgcn_s_setprio(RN_NARROW_PRIO);  // the short source-row stage goes ahead of other waves' triplet loops
#endif
    for (int r = tid; r < rows; r += 256) {
      float x[FE], nb[FN];
      load_row<FE>(a.edge_in + (erow0 + eo0 + r) * FeP, x);
      load_row<FN>(a.node + (nrow0 + qb[r]) * FnP, nb);
      float q[W2], sum = 0.f;
#pragma unroll
      for (int c = 0; c < W2; ++c) q[c] = 0.f;
      gated_matvec<FE, FE, FeP, 4 * FeP, 2 * FeP>(c3WeT, x, q);
      gated_matvec<FN, FE, FeP, 6 * FeP, 0>(c3WnT, nb, q);
#pragma unroll
      for (int c = 0; c < W2; ++c) sum += q[c];
      const float mean = sum * inv2n;
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < W2; ++c) {
        q[c] -= mean;
        if (pad_col(c)) q[c] = 0.f;
        ss = fmaf(q[c], q[c], ss);
      }
      if (FASTG) {
#pragma unroll
        for (int c = 0; c < W2; ++c) q[c] *= g3(c);
        ss *= inv2n;
      }
      float *row = bufQ + r * LDQ;
#pragma unroll
      for (int j = 0; j < W2 / 4; ++j)
        *reinterpret_cast<float4 *>(row + 4 * j) = make_float4(q[4 * j], q[4 * j + 1], q[4 * j + 2], q[4 * j + 3]);
#pragma unroll
      for (int c = W2 / 4 * 4; c < W2; ++c) row[c] = q[c];
      row[W2] = ss;
    }
    __syncthreads();

    // ================= destination edges: one lane each
#if RN_NARROW_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    for (int i = tid; i < dcount; i += 256) {
      const int dst = d_edge[i];
      const float *xrow = a.edge_in + (erow0 + dst) * FeP;
      const float *njrow = a.node + (nrow0 + j0 + d_bl[i]) * FnP, *nkrow = a.node + (nrow0 + d_a[i]) * FnP;
      // P'_d = W4 edge_d + Wj node[j] + Wk node[k] + bias, centred.  (The three operand rows are
      // re-read after the triplet loop instead of being kept alive across it: 24 registers.)
      float p[W2], sum = 0.f;
      {
        float x[FE], nj[FN], nk[FN];
        load_row<FE>(xrow, x);
        load_row<FN>(njrow, nj);
        load_row<FN>(nkrow, nk);
#pragma unroll
        for (int c = 0; c < W2; ++c) p[c] = c3shift[2 * FeP + gcol<FE, FeP>(c)];
        gated_matvec<FE, FE, FeP, 4 * FeP, 0>(c3WeT, x, p);
        gated_matvec<FN, FE, FeP, 6 * FeP, 2 * FeP>(c3WnT, nj, p);
        gated_matvec<FN, FE, FeP, 6 * FeP, 4 * FeP>(c3WnT, nk, p);
#pragma unroll
        for (int c = 0; c < W2; ++c) sum += p[c];
      }
      const float mean = sum * inv2n;
      float sp = 0.f;
#pragma unroll
      for (int c = 0; c < W2; ++c) {
        p[c] -= mean;
        if (pad_col(c)) p[c] = 0.f;
        sp = fmaf(p[c], p[c], sp);
      }
      float acc[FE];
#pragma unroll
      for (int k = 0; k < FE; ++k) acc[k] = 0.f;
      const int rb = d_rb[i], cnt = d_cnt[i], rskip = d_skip[i];
      if constexpr (FASTG) {
        // pd = p / gamma * (2 / 2Fe), pg = p * gamma:  var + eps = pd.qg + (|p|^2/2Fe + eps) + |q|^2/2Fe
        float pd[W2];
#pragma unroll
        for (int c = 0; c < W2; ++c) {
          const float gam = g3(c);
          pd[c] = p[c] * (2.0f * inv2n) / gam;
          p[c] *= gam;
        }
        const float spe = sp * inv2n + 1e-5f;
        for (int t = 0; t < cnt; ++t) {
          const int rq = rb + t + ((rb + t >= rskip) ? 1 : 0);
          const float *qr = bufQ + rq * LDQ;
          float q[W2];
#pragma unroll
          for (int j = 0; j < W2 / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(qr + 4 * j);
            q[4 * j] = v.x; q[4 * j + 1] = v.y; q[4 * j + 2] = v.z; q[4 * j + 3] = v.w;
          }
#pragma unroll
          for (int c = W2 / 4 * 4; c < W2; ++c) q[c] = qr[c];
          float dot0 = 0.f, dot1 = 0.f;
#pragma unroll
          for (int c = 0; c < W2; c += 2) {
            dot0 = fmaf(pd[c], q[c], dot0);
            if (c + 1 < W2) dot1 = fmaf(pd[c + 1], q[c + 1], dot1);
          }
          float ve = (dot0 + dot1) + (spe + qr[W2]);
          ve = ve > 1e-5f ? ve : 1e-5f;
          const float rstd = fast_rsq(ve);
#pragma unroll
          for (int k = 0; k < FE; ++k) {
            const float e1 = fast_exp2((p[k] + q[k]) * rstd + b3(k));
            const float e2 = fast_exp2((p[FE + k] + q[FE + k]) * rstd + b3(FE + k));
            const float t2 = 1.0f + e2;
            acc[k] = fmaf(e2 - 1.0f, fast_rcp(fmaf(e1, t2, t2)), acc[k]);
          }
        }
      } else {
        for (int t = 0; t < cnt; ++t) {
          const int rq = rb + t + ((rb + t >= rskip) ? 1 : 0);
          const float *qr = bufQ + rq * LDQ;
          float q[W2];
#pragma unroll
          for (int j = 0; j < W2 / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(qr + 4 * j);
            q[4 * j] = v.x; q[4 * j + 1] = v.y; q[4 * j + 2] = v.z; q[4 * j + 3] = v.w;
          }
#pragma unroll
          for (int c = W2 / 4 * 4; c < W2; ++c) q[c] = qr[c];
          float dot0 = 0.f, dot1 = 0.f;
#pragma unroll
          for (int c = 0; c < W2; c += 2) {
            dot0 = fmaf(p[c], q[c], dot0);
            if (c + 1 < W2) dot1 = fmaf(p[c + 1], q[c + 1], dot1);
          }
          const float var = fmaxf((sp + qr[W2] + 2.0f * (dot0 + dot1)) * inv2n, 0.0f);
          const float rstd = fast_rsq(var + 1e-5f);
#pragma unroll
          for (int k = 0; k < FE; ++k) {
            // (one scalar operand per instruction: rstd * gamma first, then an fma with beta -- the
            //  form ((p+q) rstd) gamma + beta needs two scalars in one fma, i.e. an extra v_mov each)
            const float yf = fmaf(p[k] + q[k], rstd * g3(k), b3(k));
            float yc = fmaf(p[FE + k] + q[FE + k], rstd * g3(FE + k), b3(FE + k));
            yc = fminf(fmaxf(yc, -43.28f), 43.28f);
            const float e1 = fast_exp2(yf), e2 = fast_exp2(yc);
            const float t2 = 1.0f + e2;
            acc[k] = fmaf(e2 - 1.0f, fast_rcp(fmaf(e1, t2, t2)), acc[k]);
          }
        }
      }
      float c3[FE];
      ln_row1<FE, PADDED>(acc, c3n2g, c3n2b, c3, fe_rt);
      // c2: gate(LayerNorm(c2_linear(node[j] * node[k]))) -> LayerNorm   (_gnn.py:223-228)
      float x[FE], nj[FN], nk[FN];
      load_row<FN>(njrow, nj);
      load_row<FN>(nkrow, nk);
      load_row<FE>(xrow, x);
      float c2pre[W2], z[FN];
#pragma unroll
      for (int k = 0; k < FN; ++k) z[k] = nj[k] * nk[k];
#pragma unroll
      for (int c = 0; c < W2; ++c) c2pre[c] = c2bias[gcol<FE, FeP>(c)];
      gated_matvec<FN, FE, FeP, 2 * FeP, 0>(c2WT, z, c2pre);
      float g2[FE], c2[FE], out[FE];
      ln_gate_row<FE, FeP, PADDED>(c2pre, c2n1g, c2n1b, g2, fe_rt);
      ln_row1<FE, PADDED>(g2, c2n2g, c2n2b, c2, fe_rt);
#pragma unroll
      for (int k = 0; k < FE; ++k) out[k] = fast_tanh(x[k] + c2[k] + c3[k]);
      store_row<FE, FeP>(a.edge_out + (erow0 + dst) * FeP, out);
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

// ============================================================================ launchers
// Exact instantiations: the documented widths and those of the reference's own tests.  Everything else with
// Fn, Fe <= 16 runs on the PADDED instantiation of its widths rounded up to multiples of four.
#define RN_NARROW_EXACT(X) X(5, 14) X(5, 5)
#define RN_NARROW_GRID(X) \
  X(4, 4) X(4, 8) X(4, 12) X(4, 16) X(8, 4) X(8, 8) X(8, 12) X(8, 16) X(12, 4) X(12, 8) X(12, 12) X(12, 16) \
  X(16, 4) X(16, 8) X(16, 12) X(16, 16)

bool narrow_supported(Dims d) { return d.FnP == 16 && d.FeP == 16 && d.Fn >= 1 && d.Fe >= 1; }

size_t edge_narrow_lds_bytes(int fe, int tile_out_rows, int tile_in_rows) {
  return narrow_lds((fe + 3) / 4 * 4, tile_out_rows, tile_in_rows).total;  // (an upper bound for the exact widths)
}

template <int FN, int FE, bool PADDED>
static void launch_node_tiled_cfg(const NodeNarrowArgs &a, hipStream_t st) {
  auto kern = &node_tiled_kernel<FN, FE, PADDED>;
  const size_t lds = node_tiled_lds(FN, a.g.nt_max_in_rows, a.g.nt_max_nodes).total;
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
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, 4);
  int nsg = per_cu * cus / a.g.nt_num;
  nsg = nsg < 1 ? 1 : (nsg > a.S ? a.S : nsg);
  kern<<<(unsigned)nsg * (unsigned)a.g.nt_num, 256, lds, st>>>(a);
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
                   w.c1_norm.g, w.c1_norm.b, w.final_norm.g, w.final_norm.b};
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

template <int FN, int FE, bool FASTG, bool PADDED>
static void launch_edge_cfg(const EdgeNarrowArgs &a, hipStream_t st) {
  auto kern = &edge_narrow_kernel<FN, FE, FASTG, PADDED>;
  const size_t lds = narrow_lds(FE, a.g.max_tile_out_rows, a.g.max_tile_in_rows).total;
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
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, 4);
  int nsg = per_cu * cus / a.g.num_tiles;
  nsg = nsg < 1 ? 1 : (nsg > a.S ? a.S : nsg);
  kern<<<(unsigned)nsg * (unsigned)a.g.num_tiles, 256, lds, st>>>(a);
}

void launch_edge_narrow(const float *edge_in, float *edge_out, const float *node, int S, const Graph &g, Dims d,
                        const PassW<float> &w, hipStream_t st) {
  if (S == 0 || g.E == 0) return;
  EdgeNarrowArgs a{edge_in, edge_out, node, S, d.Fe, g, w.c3_WeT, w.c3_WnT, w.c3_nshift, w.c2_WT, w.c2_bias,
                   w.c3_norm_1s.g, w.c3_norm_1s.b, w.c3_norm_2.g, w.c3_norm_2.b,
                   w.c2_norm_1.g, w.c2_norm_1.b, w.c2_norm_2.g, w.c2_norm_2.b};
  // The folded-scale triplet loop (FASTG, kernels_fused.hip) keeps 2 Fe more values per lane alive;
  // here that costs a wave per SIMD (132 vs 100 VGPRs at Fe = 14) and measured 5 % slower than the
  // general loop (4.23 vs 4.03 us per 256-atom structure); forced down to 128 VGPRs (four workgroups per
  // CU again, 16 bytes of scratch) it ties with the general loop (217 k vs 219 k structures/s) although
  // it issues 209 instead of 257 instructions per triplet.  So the general loop is always used.
#define X(FN, FE)                                  \
  if (d.Fn == FN && d.Fe == FE) {                  \
    launch_edge_cfg<FN, FE, false, false>(a, st);  \
    return;                                        \
  }
  RN_NARROW_EXACT(X)
#undef X
  const int fnc = (d.Fn + 3) / 4 * 4, fec = (d.Fe + 3) / 4 * 4;
#define X(FN, FE)                                 \
  if (fnc == FN && fec == FE) {                   \
    launch_edge_cfg<FN, FE, false, true>(a, st);  \
    return;                                       \
  }
  RN_NARROW_GRID(X)
#undef X
}

void launch_readout_narrow(const float *edge, const float *unit4, int S, const Graph &g, Dims d,
                           const ReadoutW<float> &w, const double *mean9, const double *std9, float *vec6,
                           double *alpha, double *alpha_raw, float *pol, hipStream_t st) {
  if (S == 0) return;
  ReadoutNarrowArgs a{edge, unit4, S, g, w, mean9, std9, vec6, alpha, alpha_raw, pol};
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
