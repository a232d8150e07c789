// Kernel launch interface between api.hip (host orchestration) and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace rn {

// Frozen reference graph on the device (all int32).
struct Graph {
  int N, E;
  const int *edge_a;   // [E]  source atom a of edge (a -> b); edges sorted by (a, b)
  const int *edge_b;   // [E]  destination atom b
  const int *out_ptr;  // [N+1] CSR over a: edges out_ptr[a]..out_ptr[a+1] leave a
  const int *in_ptr;   // [N+1] CSR over b
  const int *in_edge;  // [E]  edge ids entering b, ascending
  const int *in_pos;   // [E]  inverse of in_edge: position of edge e in the (b, a) order
  const int *atom_type;  // [N]
  const int *rev_edge;   // [E]  id of the reverse edge (b -> a), or -1
  // node tiles for the edge-block kernel
  int num_tiles;
  const int *tile_begin;  // [num_tiles+1] node ranges
  int max_tile_out_rows;  // LDS rows needed by the largest tile
  int max_tile_in_rows;   // most destination edges entering one tile
  int max_tile_nodes;     // most atoms in one tile
  // node tiles of the fused NodeBlock kernel (its own partition: it holds no Q' rows, so its tiles can
  // be larger than the EdgeBlock's and fill their rounds better)
  int nt_num;
  const int *nt_begin;    // [nt_num+1] node ranges
  int nt_max_in_rows, nt_max_nodes;
  int nt_narrow;          // the partition was made for node_tiled_kernel (kernels_narrow.hip)
  // atom-owning fused NodeBlock (kernels_node_atom.hip): tiles of 16 consecutive atoms, round r = their r-th in-edges
  int na_num;             // ceil(N / 16), or 0 when that kernel does not serve the graph
  int na_max_deg;         // largest in-degree
  // node tiles of the twelve-wave EdgeBlock kernel (edge_block3_kernel: one workgroup per CU, 48 destinations a round)
  int et_num;
  const int *et_begin;    // [et_num+1] node ranges
  int et_max_out_rows, et_max_in_rows, et_max_nodes;
  // node tiles of the role-specialised EdgeBlock (kernels_edge_ps.hip: one twelve-wave workgroup per CU, 16 destinations a round)
  int pt_num;
  const int *pt_begin;    // [pt_num+1] node ranges
  int pt_max_out_rows, pt_max_in_rows;
  int pt_back;            // rounds that may still read the ring when a step rewrites it (2 .. 5: edge_ps_tile_ok, the largest the ring has room for)
  int pt_gram;            // 1: the GRAM instantiation (LayerNorm cross terms on the matrix pipe; a 7-tile ring + the Gram tables)
  // node tiles of the EdgeBlock reverse kernel (edge_bwd_tile2_kernel): small enough for TWO workgroups per CU
  int bt_num;
  const int *bt_begin;    // [bt_num+1] node ranges
  int bt_max_out_rows, bt_max_in_rows, bt_max_nodes;
  // triplet enumeration
  const int *trip_off;  // [E+1] exclusive prefix of triplets per destination edge
  int64_t T;
};

// One LayerNorm's affine parameters (device pointers, padded with zeros).
template <typename T>
struct Ln {
  const T *g, *b;
};

template <typename T>
struct PassW {
  const T *c1_WnT;   // [FnP][2FnP]   node part of c1_linear, transposed, [filter|core]
  const T *c1_WeT;   // [FeP][2FnP]   edge part
  const T *c1_bias;  // [2FnP]
  Ln<T> c1_norm;     // [2FnP]
  Ln<T> final_norm;  // [FnP]
  const T *c2_WT;    // [FnP][2FeP]
  const T *c2_bias;  // [2FeP]
  Ln<T> c2_norm_1;   // [2FeP]
  Ln<T> c2_norm_2;   // [FeP]
  const T *c3_WnT;   // [FnP][6FeP]   (W_i | W_j | W_k) node parts of c3_linear
  const T *c3_nshift;  // [6FeP]      (0 | c3 bias | 0)
  const T *c3_WeT;   // [FeP][4FeP]   (W_4 | W_5) edge parts: dest-edge | source-edge
  Ln<T> c3_norm_1;   // [2FeP]
  Ln<T> c3_norm_2;   // [FeP]
  Ln<T> c3_norm_1s;  // [2FeP] c3_norm_1 times the gate's exp2 scales (-log2e on the filter half, 2 log2e on the core half)
  Ln<T> c2_norm_1s;  // [2FeP] c2_norm_1 and
  Ln<T> c1_norm_s;   // [2FnP] c1_norm in the same form (kernels_narrow.hip)
  const T *mfma_scale;  // [8] split-f16 prescales (s, 1/s) of c1_WeT | c3_WeT[:, W4] | c3_WeT[:, W5] | c2_WT (mfma_prescale)
  // c3_linear / c2_linear centred over their real output columns (LayerNorm(x) = LayerNorm(x - mean x), and the
  // mean is linear in the inputs): projections with these come out with zero row mean (kernels_edge_ps.hip)
  const T *c3_WeT_c;    // [FeP][4FeP]
  const T *c3_WnT_c;    // [FnP][6FeP]
  const T *c3_nshift_c; // [6FeP]
  const T *c2_WT_c;     // [FnP][2FeP]
  const T *c2_bias_c;   // [2FeP]
  // c1_linear centred the same way (the fused NodeBlock then needs no LayerNorm(2Fn) mean either)
  const T *c1_WnT_c;    // [FnP][2FnP]
  const T *c1_WeT_c;    // [FeP][2FnP]
  const T *c1_bias_c;   // [2FnP]
  const T *mfma_scale_c;  // [8] split-f16 prescales (s, 1/s) of c3_WeT_c[:, W4] | c3_WeT_c[:, W5] | c2_WT_c | c1_WeT_c
  int c3_fast;       // host-side decision: c3_norm_1 admits the folded-scale triplet loop
                     // (bit 0: in the fused EdgeBlock kernel, bit 1: in edge_agg_kernel)
};

template <typename T>
struct ReadoutW {
  // HP = max(FeP, 32): hidden width of the readout MLP as stored
  const T *W0T;  // [FeP][HP]
  const T *scale0, *shift0;  // [HP] BatchNorm(eval) folded with bias 0: ssp(acc*scale+shift)
  const T *W3T;  // [HP][HP]
  const T *b3;   // [HP]
  const T *W5T;  // [HP][32]  (12 real columns)
  const T *b5;   // [32]
  const T *mfma_scale;  // [8] split-f16 prescales (s, 1/s) of W0T | W3T | W5T (mfma_prescale)
};

// Power-of-two prescale of a weight matrix for the split-f16 matrix products (device_utils.hpp):
// s = 2^(12 - floor(log2 m)) for m = max |w|, so that s w lies in [2^12, 2^13) -- inside f16's
// normal range whatever the scale of the weights, with every entry down to 2^-15 of the largest
// one keeping its full 22 bits.  Exact (a power of two); the inverse is applied to the float32
// accumulator.  Same bit arithmetic on the host (pack_weights) and on the device (kernels_train.hip).
__host__ __device__ inline float mfma_prescale(float m) {
  union { float f; unsigned u; } b;
  b.f = m;
  const unsigned ex = (b.u >> 23) & 0xffu;
  if (!(m > 0.0f) || ex == 0xffu) return 1.0f;  // zero, negative, NaN, inf: leave unscaled
  int e = (int)ex - 127;
  if (e < -126) e = -126;
  int se = 12 - e;
  se = se < -126 ? -126 : (se > 126 ? 126 : se);
  b.u = (unsigned)(se + 127) << 23;
  return b.f;
}

struct Dims {
  int Fn, Fe, FnP, FeP;  // logical and padded (pow2 >= 16) embedding widths
};

// ---- launchers (all asynchronous on `st`) ----
template <typename T>
void launch_setup(const T *emb, const T *W2, const T *b2, const T *W4, const T *b4, int K,
                  Dims d, T *node_table, const T *b0, const T *bn_w, const T *bn_b,
                  const T *bn_rm, const T *bn_rv, T *scale0, T *shift0, hipStream_t st);

template <typename T>
void launch_geom_rbf(const double *pos, int S, const Graph &g, const T *lattice,
                     int lat_stride /* 0: one lattice for every frame; 9: lattice[S][9] */,
                     const T *offsets, T coef, Dims d, T *unit4, T *edge0, hipStream_t st,
                     bool in_order = false /* rows in (b, a) order: the narrow kernels' layout */);
// "Pair" edge rows (float32 evaluations on the role-specialised EdgeBlock + atom-owning NodeBlock + fused readout, FeP = 64):
// a row's 256 bytes hold, for each group m of eight columns, [f16 hi x8][f16 lo x8] with hi = f16(x), lo = f16(x - hi) --
// the split-f16 MFMA operand itself (device_utils.hpp), written once by the kernel that produces the row instead of being
// re-derived by each of its three consumers; x = hi + lo to 2^-25 absolute for |x| < 1 (tanh / Gaussian outputs).
void launch_geom_rbf_pairs(const double *pos, int S, const Graph &g, const float *lattice, int lat_stride, const float *offsets,
                           float coef, Dims d, float *unit4, float *edge0, hipStream_t st);
// The two float32 geometry kernels on positions that are float32 already (cast on the host while staging: bit-identical rows)
void launch_geom_rbf_pos32(const float *pos, int S, const Graph &g, const float *lattice, int lat_stride, const float *offsets,
                           float coef, Dims d, float *unit4, float *edge0, hipStream_t st, bool in_order);
void launch_geom_rbf_pairs_pos32(const float *pos, int S, const Graph &g, const float *lattice, int lat_stride,
                                 const float *offsets, float coef, Dims d, float *unit4, float *edge0, hipStream_t st);

template <typename T>
void launch_node_init(const T *table, int S, const Graph &g, Dims d, T *node,
                      const int *types /* [S*N] atom type per (sample, atom), or null: g.atom_type */, hipStream_t st);

// Y[M, NOUT] = act(X[M, KP] * WT[KP, NOUT] * scale + shift)
//   amode 0: X rows are read from `X` (row stride KP)
//   amode 1: X row r = node[s*N + b_e] * node[s*N + a_e]  (r = s*E + e)   -- c2 input
template <typename T>
void launch_rowgemm(const T *X, int64_t M, int KP, const T *WT, int NOUT, T *Y,
                    const T *scale, const T *shift, bool act, int amode, const T *node,
                    const Graph &g, hipStream_t st);

// float32 only: Y (+)= X[:, 0:N] * Wt[N, NOUT] on the MFMA kernel; false if the shape is unsupported
// LDS bytes of edge_bwd_tile2_kernel for a tile with these maxima (element size 4 or 8)
size_t edge_bwd_tile2_lds_bytes(int rows, int in_rows, int nodes, int FP, size_t elem);
// split-f16 row product for the reverse pass (kernels_gemm.hip: rowgemm_split_kernel); false = shape not served
bool launch_rowgemm_split(const float *X, int ldx, int K, int64_t M, const float *Wt, int NOUT, float *Y,
                          bool accumulate, const float *bias, int amode, const float *node, const Graph &g,
                          hipStream_t st);
bool launch_rowgemm_blocks(const float *X, int ldx, int N, int64_t M, const float *Wt, int NOUT,
                           float *Y, bool accumulate, const Graph &g, hipStream_t st);

template <typename T>
void launch_node_agg(const T *npc1, const T *bc1, const T *node_in, T *node_out, int S,
                     const Graph &g, Dims d, const PassW<T> &w, hipStream_t st);

template <typename T>
void launch_edge_agg(const T *pq, const T *np3, const T *c2pre, const T *edge_in,
                     T *edge_out, int S, const Graph &g, Dims d, const PassW<T> &w,
                     T *agg_out /* optional tape of the pre-LayerNorm triplet sums */,
                     hipStream_t st);

template <typename T>
void launch_readout_reduce(const T *pol, const T *unit4, int S, const Graph &g,
                           const double *mean9, const double *std9, float *vec6,
                           double *alpha, double *alpha_raw, hipStream_t st, int pol_stride = 32 /* floats between rows of pol */);

void launch_radius_graph(const double *lattice, const double *pos, int N, float cutoff,
                         unsigned char *adjacency, hipStream_t st);

void launch_enum_triplets(const Graph &g, int *idx_i, int *idx_j, int *idx_k, int *slot5,
                          int *slot6, hipStream_t st);

size_t edge_agg_lds_bytes(const Graph &g, Dims d, size_t elem);

// ---- reverse mode w.r.t. activations / positions (kernels_bwd.hip).  C cotangent
// instances, B per forward frame.
template <typename T>
void launch_gemm_nt(const T *X, int64_t R, int N, const T *W, int ldw, int K, T *Y, bool accumulate,
                    hipStream_t st);
template <typename T>
void launch_readout_bwd(const T *dout6, const T *pol, const T *unit4, int C, int B, const Graph &g,
                        T *dpol, T *dunit, hipStream_t st);
template <typename T>
void launch_ssp_bwd(T *d, const T *h, const T *scale, int64_t rows_per_frame, int width, int C,
                    int B, hipStream_t st);
template <typename T>
void launch_edge_bwd(const T *pq, const T *np3, const T *c2pre, const T *edge_next,
                     const T *agg /* taped pre-LayerNorm triplet sums [S*E, FeP] */,
                     const T *dedge_next, T *dedge_prev, T *dpq, T *dnp3, T *dc2pre, int C, int B,
                     const Graph &g, Dims d, const PassW<T> &w, const PassW<T> *grad_w,
                     hipStream_t st);  // grad_w: same layout as w inside the gradient blob, or null
template <typename T>
void launch_prod_fwd(const T *node, T *prod, int64_t rows, const Graph &g, Dims d, hipStream_t st);
template <typename T>
void launch_prod_bwd(const T *dprod, const T *node, T *dnode, int C, int B, const Graph &g, Dims d,
                     hipStream_t st);
template <typename T>
void launch_node_bwd(const T *npc1, const T *bc1, const T *node_next, const T *dnode_next,
                     T *dnode_prev, T *dbc1, T *dnpc1, int C, int B, const Graph &g, Dims d,
                     const PassW<T> &w, const PassW<T> *grad_w, hipStream_t st);
// ---- training pieces (weight gradients, BatchNorm in training mode, embedding MLP)
// Deferred reduction of the weight-gradient products (float32 MFMA path): every product stores one
// [K+1][N] slice per workgroup into `arena`; launch_tn_reduce sums the slices of all products recorded
// in `table` with one launch (and resets the record).
struct TnReduceOp {
  const float *part;
  float *dst, *dbias;
  int nblk, K, N, ldw;
};
struct TnReduceTable {
  static constexpr int MAX_OPS = 32;
  TnReduceOp ops[MAX_OPS];
  int wg_begin[MAX_OPS + 1];
  int num;
};
struct TnDeferred {
  float *arena = nullptr;
  size_t capacity = 0, used = 0;  // in floats
  TnReduceTable table{};
};
size_t tn_partial_elems(int64_t R, int K, int N);
void launch_tn_reduce(TnDeferred &df, hipStream_t st);
template <typename T>
void launch_gemm_tn(const T *X, int ldx, const T *dY, int ldy, int64_t R, int K, int N, T *dWT,
                    int ldw, T *dbias, int amode, const T *node, const Graph &g, hipStream_t st,
                    TnDeferred *defer = nullptr);
template <typename T>
void launch_bn_col_sums(const T *z, int64_t R, int W, double *stats, hipStream_t st);
template <typename T>
void launch_bn_train_apply(const T *z, int64_t R, int W, int F, const double *stats, double count,
                           const T *gamma, const T *beta, T *h, T *mv, hipStream_t st);
template <typename T>
void launch_bn_bwd_sums(const T *d, const T *z, int64_t R, int W, int F, const double *stats, double count,
                        double *sums, hipStream_t st);
template <typename T>
void launch_bn_bwd_apply(T *d, const T *z, int64_t R, int W, int F, const double *stats, double count,
                         const double *sums, const double *own_sums, const T *gamma, T *dgamma, T *dbeta,
                         hipStream_t st);
template <typename T>
void launch_node_embed_bwd(const T *dnode0, int S, const Graph &g, Dims d, int K, const T *emb,
                           const T *W2, const T *b2, const T *W4, T *demb, T *dW2, T *db2, T *dW4,
                           T *db4, T *type_sums_scratch,
                           const int *types /* [S*N] atom type per (sample, atom), or null: g.atom_type */, hipStream_t st);
template <typename T>
void launch_geom_bwd(const T *dedge0, const T *dunit, const T *unit4, const T *lat, const T *offs,
                     T coef, int C, int B, const Graph &g, Dims d, double *dpos, hipStream_t st);

// Fused EdgeBlock (kernels_fused.hip): projections + triplet aggregation in one launch.
// Fused EdgeBlock (kernels_fused.hip): float32, FnP == FeP == 64.  Two workgroups per CU
// need their LDS footprint within this budget.
constexpr size_t kFusedLdsBudget = 80 * 1024;
size_t edge_fused_lds_bytes(int tile_out_rows, int tile_in_rows, int tile_nodes);
bool edge_fused_supported(const Graph &g, Dims d);
size_t node_fused_lds_bytes(const Graph &g);
size_t node_fused_lds_bytes(int tile_in_rows, int tile_nodes);
// `f16`: matrix products as three split-f16 MFMAs (device_utils.hpp) instead of the exact-f32 MFMA
void launch_readout_fused(const float *edge, int64_t M, const ReadoutW<float> &w, float *pol, bool f16,
                          hipStream_t st, bool pair_rows = false, int pol_stride = 32 /* 16: rows of exactly the 16 columns written */);
// `centred`: npc1 was projected with c1_WnT_c / c1_bias_c and the kernel multiplies with c1_WeT_c: zero row mean, the
// LayerNorm(2Fn) in front of the gate needs the sum of squares only (split-f16 instantiations)
size_t node_atom_lds_bytes(int max_deg);
// (split-f16 products on the centred c1_linear only: PassW::c1_WeT_c, npc1 from c1_WnT_c / c1_bias_c)
void launch_node_atom(const float *edge, const float *node_in, const float *npc1, float *node_out, int S, const Graph &g,
                      Dims d, const PassW<float> &w, hipStream_t st, bool pair_rows = false);
void launch_node_fused(const float *edge, const float *node_in, const float *npc1, float *node_out, int S,
                       const Graph &g, Dims d, const PassW<float> &w, bool f16, bool centred, hipStream_t st);
// Role-specialised fused EdgeBlock (kernels_edge_ps.hip): float32, FnP == FeP == 64, split-f16 products, folded gate
// scale; needs the centred weight copies of PassW and np3 projected with c3_WnT_c / c3_nshift_c.
size_t edge_ps_lds_bytes(int tile_out_rows, int tile_in_rows, bool gram);
int edge_ps_ring_tiles(bool gram);  // ring capacity in 16-row tiles
int edge_ps_gram_window();          // source tiles a round's window may span in the GRAM instantiation
// one tile's destinations (first and end source row of each, sorted by atom): does the producers' schedule hold?
// `back`: rounds g - back .. g - 1 may still be reading the ring when the step of round g rewrites it (2 .. 5 in product handles); `ring`: its
// capacity in tiles; `window` (optional, out): the most source tiles one round's window spans
bool edge_ps_tile_ok(const int *rb, const int *re, int D, int back, int ring, int *window);
// `fail`: device int, set to a nonzero code if a bounded spin wait inside the kernel ran out (never in a correct run)
void launch_edge_ps(const float *edge_in, float *edge_out, const float *node, const float *np3, float *agg_out, int S,
                    const Graph &g, Dims d, const PassW<float> &w, int *fail, hipStream_t st, bool pair_rows = false,
                    bool f16 = true /* false: exact-f32 MFMA products (float32 rows, no Gram tables) */);

// Opt-in experiment kernels (experiments/kernels_fused_experiments.hip): compiled and reachable only with
// -DRN_EXPERIMENTS=1; the product build has neither the kernels nor the RN_POTGNN_EDGE2 / EDGE3 / NODE_WAVE knobs.
#ifndef RN_EXPERIMENTS
#define RN_EXPERIMENTS 0
#endif
#if RN_EXPERIMENTS
// Per-frame fused EdgeBlock (edge_block_fused_kernel, retired from the product build in round 5: experiments/kernels_edge_frame.hip)
size_t edge_fused_lds_bytes(const Graph &g);
// `agg_out` (taped runs, else null): the pre-LayerNorm triplet sums per destination edge
void launch_edge_fused(const float *edge_in, float *edge_out, const float *node, const float *np3,
                       float *agg_out, int S, const Graph &g, Dims d, const PassW<float> &w, bool f16,
                       hipStream_t st);
// Twelve-wave EdgeBlock (edge_block3_kernel + edge_c2_kernel): ONE 768-thread workgroup per CU with the CU's LDS.
#ifndef RN_E3_WAVES
#define RN_E3_WAVES 12
#endif
constexpr size_t kEdge3LdsBudget = RN_E3_WAVES <= 4 ? 54272 : (RN_E3_WAVES <= 6 ? 80 : 158) * 1024;
size_t edge3_lds_bytes(int rows, int in_rows, int nodes);
inline int edge3_dests_per_round() { return 4 * RN_E3_WAVES; }
bool edge3_supported(const Graph &g, Dims d);
bool edge3_applicable(const PassW<float> &w, bool f16);
void launch_edge3(const float *edge_in, float *edge_out, const float *np3, const float *c2, float *agg_out, int S,
                  const Graph &g, Dims d, const PassW<float> &w, hipStream_t st);
bool node_fused_wave_tiles();  // the split-f16 NodeBlock runs wave-autonomous: 16-row tiles, four per workgroup step
size_t node_wave_lds_bytes(int tile_in_rows, int tile_nodes);
// Frame-pipelined EdgeBlock (kernels_fused.hip: edge_block2_kernel) and the c2 branch as its own
// streaming kernel: float32, FnP == FeP == 64, same LDS budget for two workgroups per CU.
size_t edge2_lds_bytes(int tile_out_rows, int tile_in_rows, int tile_nodes);
bool edge2_supported(const Graph &g, Dims d);
// c2[S*E, FeP] = LayerNorm(gate(LayerNorm(c2_linear(node[b] * node[a]))))  (_gnn.py:200-228)
void launch_edge_c2(const float *node, float *c2, int S, const Graph &g, Dims d, const PassW<float> &w, bool f16,
                    hipStream_t st);
void launch_edge2(const float *edge_in, float *edge_out, const float *np3, const float *c2, float *agg_out, int S,
                  const Graph &g, Dims d, const PassW<float> &w, bool f16, hipStream_t st);
#endif  // RN_EXPERIMENTS

// Device-resident optimisation step (kernels_train.hip); offsets index the packed weight blob.
struct DerivedOp {
  int kind;        // 0: transposed copy of a [K][N] block, 1: scaled copy of K values,
                   // 2: dst[0..1] = (s, 1/s), s = mfma_prescale(max |src[k * ld + n]|), ld = (int)scale
                   // 3: dst[k][n] = src[k][n] - mean over the REAL columns of n's [filter | core] block (2 FeP wide, the
                   //    first Fe of each half real; padded columns stay 0), rows of N = ld columns: the centred copies
                   //    of c3_linear / c2_linear (kernels_edge_ps.hip)
  int K, N;
  float scale;
  size_t src, dst;
  int Fe = 0, FeP = 0;  // kind 3
};
// (kind 2 entries that read a kind 3 result go into a second launch: blocks of one launch run concurrently)
void launch_adam(float *w, const float *g, float *m, float *v, const unsigned char *trainable, size_t n,
                 double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                 hipStream_t st, const int *poisoned = nullptr /* device word: non-zero = apply nothing */);
void launch_refresh_derived(float *w, const DerivedOp *ops, int num_ops, hipStream_t st);
// staging[dst .. dst + n) = w[src .. src + n) for every (src, dst, n) of `seg` (kernels_train.hip)
void launch_gather_segments(const float *w, const long long *seg, int nseg, float *staging, hipStream_t st);
void launch_bn_running(float *running_mean, float *running_var, const float *batch_mean, const float *batch_var,
                       int F, double momentum, double unbias, hipStream_t st);

// Narrow-width kernels (kernels_narrow.hip): one lane per row, compile-time (Fn, Fe) <= 16, float32.
bool narrow_supported(Dims d);
size_t edge_narrow_lds_bytes(int fn, int fe, int tile_out_rows, int tile_in_rows);
size_t node_tiled_lds_bytes(int fn, int fe, int tile_in_rows, int tile_nodes);
void launch_node_narrow(const float *edge, const float *node_in, float *node_out, int S, const Graph &g, Dims d,
                        const PassW<float> &w, hipStream_t st);
void launch_edge_narrow(const float *edge_in, float *edge_out, const float *node, int S, const Graph &g, Dims d,
                        const PassW<float> &w, hipStream_t st);
void launch_readout_narrow(const float *edge, const float *unit4, int S, const Graph &g, Dims d,
                           const ReadoutW<float> &w, const double *mean9, const double *std9, float *vec6,
                           double *alpha, double *alpha_raw, float *pol /* optional [S*E,32] */, hipStream_t st);

}  // namespace rn
