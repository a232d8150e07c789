// Host orchestration + C ABI (include/rn_potgnn.h) for the gfx950 PotGNN evaluator.
//
// A handle owns: the frozen graph (CSR both ways, node tiles, triplet offsets), the
// weights re-laid-out for the kernels (transposed, [filter|core] halves padded to a
// power-of-two width, concatenated Linear layers split into per-operand blocks), and
// per-"lane" device workspaces (a lane = a HIP stream).  An evaluation walks the frames in
// chunks (hundreds to thousands of frames: large launches amortise launch gaps and tails).
// The fused pipeline (kernels_fused.hip) uses one lane; the unfused one alternates chunks
// between two lanes so that the HBM-bound projections of one chunk overlap the VALU-bound
// aggregation of the other (run_pair).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rn_potgnn.h"
#include "kernels.hpp"

using namespace rn;

namespace {

std::string g_create_error;

int pad_pow2(int f) {
  int p = 16;
  while (p < f) p *= 2;
  return p;
}

// Which padded widths a model runs at.  Fe padded to 64 with a narrower Fn (e.g. Fn = 32 / Fe = 64, Fn = 20 / Fe = 48): padding
// Fn to 64 as well puts the model on the fused MFMA kernels (their masked LayerNorms handle F < FP) instead of the unfused
// per-stage chain (round 3: 15.4 -> 12.4 us per 128-atom structure).  Since round 5 the same holds for Fe in 17..32 (below).
// RN_POTGNN_WIDEN=0: minimal power-of-two padding everywhere; =1: round 4's policy; =3: experiment.
void widen_for_fused(rn::Dims &d) {
  static const int widen = getenv("RN_POTGNN_WIDEN") ? atoi(getenv("RN_POTGNN_WIDEN")) : 2;
  if (widen == 0) return;
  if (d.FeP == 64 && d.FnP < 64) d.FnP = 64;
  // round 5: with the role-specialised EdgeBlock at 3.2 us per structure and pass the 64-wide fused kernels (7.8 us per
  // 128-atom structure whatever the real widths) are level with or ahead of the unfused chain for every Fe in 17..32 (7.5-7.9 us
  // at Fn <= 32, 9.1 at Fn in 33..64: profiles/r05/width_sweep.txt), so those pad to 64 x 64 as well: one kernel family
  // for every edge width in 17..64.  Fe <= 16 with a wide Fn stays unfused (6.0 us).  RN_POTGNN_WIDEN=1 keeps round 4's
  // policy (only Fe in 33..64 widens Fn).
  if (widen >= 2 && d.FeP == 32 && d.FnP >= 32 && d.FnP <= 64) d.FnP = d.FeP = 64;  // Fe in 17..32, Fn in 17..64 (Fn <= 16: 6.8 unfused against 7.7)
  if (widen >= 3 && d.FeP <= 16 && d.FnP >= 32 && d.FnP <= 64) d.FnP = d.FeP = 64;  // (experiment) Fe <= 16 with Fn in 17..64
}

struct HipError {
  hipError_t code;
  const char *what;
};

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) throw HipError{_e, #expr};   \
  } while (0)

struct DeviceBuf {
  void *p = nullptr;
  size_t bytes = 0;
  ~DeviceBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  void ensure(size_t n) {
    if (n <= bytes) return;
    release();
    HIP_TRY(hipMalloc(&p, n));
    bytes = n;
  }
  template <typename T>
  T *as() const { return reinterpret_cast<T *>(p); }
};

enum KernelId {
  K_GEOM = 0, K_NODE_INIT, K_PROJ_NODE, K_PROJ_EDGE_C1, K_NODE_AGG, K_PROJ_EDGE_C3, K_PROJ_C2,
  K_EDGE_AGG, K_READOUT_MLP, K_READOUT_REDUCE, K_COUNT
};
const char *kKernelNames[K_COUNT] = {
    "geom_rbf", "node_init", "proj_node", "proj_edge_c1", "node_agg", "proj_edge_c3",
    "proj_c2", "edge_agg", "readout_mlp", "readout_reduce"};

// Host-side packed weights, one flat array + offsets; uploaded per precision.
struct PackedLayout {
  // setup inputs (unpadded)
  size_t emb, W2, b2, W4, b4, b0, bn_w, bn_b, bn_rm, bn_rv;
  size_t offsets;  // [FeP]
  struct Pass {
    size_t c1_WnT, c1_WeT, c1_bias, c1n_g, c1n_b, fin_g, fin_b;
    size_t c2_WT, c2_bias, c2n1_g, c2n1_b, c2n2_g, c2n2_b;
    size_t c3_WnT, c3_nshift, c3_WeT, c3n1_g, c3n1_b, c3n2_g, c3n2_b;
    size_t c3n1_gs, c3n1_bs;  // c3_norm_1 with the gate's exp2 scale folded in (-log2e | 2 log2e): narrow kernels
    size_t c2n1_gs, c2n1_bs, c1n_gs, c1n_bs;  // the same for c2_norm_1 and c1_norm (narrow kernels)
    size_t mfma_scale;        // [8] split-f16 prescales (s, 1/s): c1_WeT | W4 | W5 | c2_WT (kernels.hpp: mfma_prescale)
    size_t t_c3We, t_c3Wn, t_c2W, t_c1We, t_c1Wn;  // transposed copies [N][K] (reverse pass)
    // copies of c3_linear / c2_linear centred over their real output columns (kernels_edge_ps.hip) and the
    // split-f16 prescales (s, 1/s) of W4 | W5 | c2 in that form
    size_t c3_WeT_c, c3_WnT_c, c3_nshift_c, c2_WT_c, c2_bias_c, mfma_scale_c;
    size_t c1_WnT_c, c1_WeT_c, c1_bias_c;  // c1_linear centred over its 2 Fn output columns (fused NodeBlock)
  };
  std::vector<Pass> pass;
  size_t W0T, W3T, b3, W5T, b5, ones, b0p, t_W0, t_W3, t_W5;
  size_t ro_mfma_scale;  // [8] split-f16 prescales (s, 1/s): W0T | W3T | W5T
  // device-computed
  size_t node_table, scale0, shift0;
  size_t total = 0;
  size_t take(size_t n) {
    size_t o = total;
    total += (n + 3) & ~size_t(3);  // keep 16-byte alignment for float4 loads
    return o;
  }
};

template <typename T>
struct Lane {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  DeviceBuf unit4, node[2], edge[2], npc1, np3, bufA, bufB;
  DeviceBuf c2;  // frame-pipelined EdgeBlock: the finished c2 embedding of every edge [S*E, FeP]
  DeviceBuf fbA, fbB;  // a block of frames' c2 / c3 edge projections for a pass the fused pipeline hands to the unfused EdgeBlock
};

template <typename T>
struct Precision {
  bool ready = false;
  DeviceBuf weights;     // packed, type T
  DeviceBuf lattice;     // [9] T
  std::vector<PassW<T>> pass;
  ReadoutW<T> ro{};
  const T *offsets = nullptr, *node_table = nullptr, *ones = nullptr;
  Lane<T> lanes[2];
  // stage snapshots (debug)
  std::vector<DeviceBuf> snap_node, snap_edge;
  // forward tape + cotangent workspace of the reverse pass (Jacobian d alpha / d r)
  std::vector<DeviceBuf> tape_node, tape_edge, tape_agg;
  DeviceBuf bw[18];
  hipStream_t side = nullptr;  // weight-gradient products of the reverse pass run here, beside the cotangent chain
  hipEvent_t ev_side[3] = {nullptr, nullptr, nullptr};
  DeviceBuf tn_arena;  // partial sums of the weight-gradient products (reverse_pass)
  DeviceBuf tape_z1, bn_stats, grad, seeds, mv, type_sums;  // training: pre-BatchNorm activations, batch sums, gradient blob
  bool tape_on = false;
};

struct TimedLaunch {
  int kid;
  hipEvent_t a, b;
};

}  // namespace

struct rn_potgnn {
  mutable std::recursive_mutex lock;  // serialises the calls on this handle (guarded(), the small getters / setters, set_error)
  rn_potgnn_config cfg{};
  Dims d{};
  int chunk = 1;
  size_t f64_budget = 0;  // workspace bytes the float64 lanes may take (0: what the float32 lanes were given)
  int num_lanes = 2;
  bool keep_stages = false;
  bool debug_sync = false;  // RN_POTGNN_DEBUG_SYNC=1: synchronise + check after every kernel
  // graph
  std::vector<int> edge_a, edge_b, out_ptr, in_ptr, in_edge, atom_type, tile_begin, trip_off, rev_edge, nt_begin, et_begin, bt_begin, pt_begin, in_pos;
  bool use_fused = false;
  bool use_edge2 = false;  // fused EdgeBlock in its frame-pipelined form (edge_block2_kernel + edge_c2_kernel)
  bool split_projections = true;  // RN_POTGNN_SPLIT_PROJ=0: the forward's stand-alone projections on the exact-f32 MFMA kernel
  bool use_edge3 = false;  // fused EdgeBlock on twelve waves, one workgroup per CU (edge_block3_kernel + edge_c2_kernel)
  bool use_ps = false;     // role-specialised fused EdgeBlock (kernels_edge_ps.hip) on its own atom tiles (Graph::pt_*)
  bool use_node_fused = false;  // fused NodeBlock (only together with the fused EdgeBlock)
  bool want_pair_rows = true;   // RN_POTGNN_PAIR_ROWS at create time (ForwardRun::pair_rows decides per run)
  DeviceBuf step_seg, step_stage;  // segments / staging of the one download after a device-resident Adam step
  float *step_host = nullptr;      // its pinned host image
  bool tape_ps = true;          // RN_POTGNN_TAPE_PS: taped float32 runs on the role-specialised EdgeBlock / atom-owning NodeBlock
  bool use_readout_fused = false;  // readout MLP in one launch (same condition)
  bool use_narrow = false;  // narrow-width kernels (kernels_narrow.hip): Fn, Fe <= 16, one lane per row
  bool mfma_f16 = true;  // fused kernels: split-f16 MFMA products are in use (requested and inside the safe range)
  bool mfma_f16_requested = true;   // RN_POTGNN_MFMA=f32 at create time: exact-f32 MFMA everywhere
  bool mfma_range_fallback = false; // split-f16 was requested but the range guard (mfma_f16_range_ok) refused it
  Graph g{};
  DeviceBuf g_ints;
  DeviceBuf ps_fail;  // [1] int: set by edge_block_ps_kernel when one of its bounded waits ran out
  double lattice[9], mean[9], stdv[9];
  DeviceBuf d_mean_std;  // [18] double
  std::vector<float> packed;  // host packed weights (float master copy)
  PackedLayout lay;
  Precision<float> f32;
  Precision<double> f64;
  // cached I/O staging for the host entry points
  DeviceBuf io_pos, io_alpha, io_vec6, io_lat, io_types;
  // pipelined host entry (rn_potgnn_calc_polarizabilities_async): two staging slots
  struct Slot {
    DeviceBuf pos, alpha;
    hipEvent_t copied = nullptr, done = nullptr;
    bool busy = false;
  } slots[2];
  hipStream_t copy_stream = nullptr, exec_stream = nullptr;
  int next_slot = 0;
  // host entry rn_potgnn_calc_polarizabilities: two page-locked buffers the caller's float64 positions are cast into
  // (float32) piece by piece, each with the event of its host-to-device copy
  struct HostStage {
    float *pin[2] = {nullptr, nullptr};
    size_t elems = 0;  // floats per buffer
    hipEvent_t copied[2] = {nullptr, nullptr};
    hipEvent_t done = nullptr;
    char *out_pin = nullptr;  // the result (+ the EdgeBlock's time-out word) of the synchronous host entry
    size_t out_bytes = 0;
  } hstage;
  int last_chunk_structs = 0;
  int train_S = 0;  // frames of the pending train_forward (0 = none)
  int train_prec = 4;  // sizeof of the precision it ran in
  bool train_lat = false, train_types = false;  // it ran with per-sample lattices / atom types (io_lat / io_types)
  bool tape_fused = true;
  // device-resident optimisation (rn_potgnn_adam_step): gradients stay in f32.grad, Adam moments and
  // the trainable mask live next to the weights; the host copy `packed` is refreshed on demand
  bool device_training = false;  // BatchNorm running statistics are updated on the device
  bool grads_on_device = false;  // f32.grad holds the gradients of the last backward
  bool host_stale = false;       // the device weights are ahead of `packed`
  DeviceBuf adam_m, adam_v, trainable_mask, derived_ops;
  int num_derived_ops = 0, derived_first_stage = 0;  // (ops after the first stage read what it wrote: second launch)
  double bn_count = 0;  // rows the pending step's BatchNorm statistics cover (all ranks)
  rn_potgnn_reduce_fn reducer = nullptr;  // data-parallel training: sums doubles over ranks
  void *reducer_ctx = nullptr;
  bool last_was_f64 = false;
  bool last_in_order = false;  // the last run kept its edge rows in (b, a) order (narrow kernels)
  // profiling
  int profiling = 0;
  std::vector<TimedLaunch> timed;
  double k_ms[K_COUNT] = {0};
  int64_t k_launches[K_COUNT] = {0};
  std::string error;
  hipEvent_t ev_start = nullptr;
  hipEvent_t ev_g[2] = {nullptr, nullptr};  // "projection stage done" per lane (run_pair)
  bool interleave = true;
};

namespace {

void set_error(rn_potgnn *h, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) {  // (entry points report argument errors before they enter guarded(): the error text is handle state too)
    std::lock_guard<std::recursive_mutex> hold(h->lock);
    h->error = buf;
  } else {
    g_create_error = buf;
  }
}

// ----------------------------------------------------------------------------- packing
// Column of the padded [filter | core] layout for original output row r of a Linear /
// LayerNorm of logical width 2F (first F rows = filter, last F = core; _gnn.py:143).
inline int gated_col(int r, int F, int FP) { return r < F ? r : FP + (r - F); }

// Largest out-degree the tiled kernels take: the rows of one atom's outgoing edges, 2 FeP float64
// values (+ an index) each, have to fit a 150 KB LDS tile.  590 for FeP = 16, 149 for 64, 74 for 128.
inline size_t max_out_degree(int FeP) { return (size_t)150 * 1024 / ((size_t)2 * FeP * sizeof(double) + 4); }

// The weight blocks that enter a split-f16 matrix product and where their (s, 1/s) pair lives.
struct MfmaScaleOp {
  size_t src;
  int K, N, ld;
  size_t dst;
};
std::vector<MfmaScaleOp> mfma_scale_ops(const rn_potgnn *h) {
  const PackedLayout &L = h->lay;
  const int FnP = h->d.FnP, FeP = h->d.FeP, HP = std::max(FeP, 32);
  std::vector<MfmaScaleOp> ops;
  for (const auto &q : L.pass) {
    ops.push_back({q.c1_WeT, FeP, 2 * FnP, 2 * FnP, q.mfma_scale});
    ops.push_back({q.c3_WeT, FeP, 2 * FeP, 4 * FeP, q.mfma_scale + 2});            // W4: destination-edge part
    ops.push_back({q.c3_WeT + 2 * FeP, FeP, 2 * FeP, 4 * FeP, q.mfma_scale + 4});  // W5: source-edge part
    ops.push_back({q.c2_WT, FnP, 2 * FeP, 2 * FeP, q.mfma_scale + 6});
  }
  ops.push_back({L.W0T, FeP, HP, HP, L.ro_mfma_scale});
  ops.push_back({L.W3T, HP, HP, HP, L.ro_mfma_scale + 2});
  ops.push_back({L.W5T, HP, 32, 32, L.ro_mfma_scale + 4});
  return ops;
}

// The centred copies: (source, destination, rows, columns) of every matrix / bias vector whose output columns feed a
// LayerNorm over a [filter | core] row in the role-specialised EdgeBlock.  LayerNorm(x) = LayerNorm(x - mean x), and
// the row mean of a Linear's output is itself linear in the input, so subtracting from every weight row (and from
// the bias) its mean over the real output columns makes the projections come out with zero row mean.
struct CentreOp {
  size_t src, dst;
  int K, N;
  int F, FP;  // the [filter | core] blocks are 2 FP columns wide, the first F of each half real
};
std::vector<CentreOp> centre_ops(const rn_potgnn *h) {
  const int FnP = h->d.FnP, FeP = h->d.FeP;
  std::vector<CentreOp> ops;
  for (const auto &q : h->lay.pass) {
    const int Fn = h->d.Fn, Fe = h->d.Fe;
    ops.push_back({q.c3_WeT, q.c3_WeT_c, FeP, 4 * FeP, Fe, FeP});
    ops.push_back({q.c3_WnT, q.c3_WnT_c, FnP, 6 * FeP, Fe, FeP});
    ops.push_back({q.c3_nshift, q.c3_nshift_c, 1, 6 * FeP, Fe, FeP});
    ops.push_back({q.c2_WT, q.c2_WT_c, FnP, 2 * FeP, Fe, FeP});
    ops.push_back({q.c2_bias, q.c2_bias_c, 1, 2 * FeP, Fe, FeP});
    ops.push_back({q.c1_WnT, q.c1_WnT_c, FnP, 2 * FnP, Fn, FnP});
    ops.push_back({q.c1_WeT, q.c1_WeT_c, FeP, 2 * FnP, Fn, FnP});
    ops.push_back({q.c1_bias, q.c1_bias_c, 1, 2 * FnP, Fn, FnP});
  }
  return ops;
}
std::vector<MfmaScaleOp> centred_scale_ops(const rn_potgnn *h) {
  const int FnP = h->d.FnP, FeP = h->d.FeP;
  std::vector<MfmaScaleOp> ops;
  for (const auto &q : h->lay.pass) {
    ops.push_back({q.c3_WeT_c, FeP, 2 * FeP, 4 * FeP, q.mfma_scale_c});
    ops.push_back({q.c3_WeT_c + 2 * FeP, FeP, 2 * FeP, 4 * FeP, q.mfma_scale_c + 2});
    ops.push_back({q.c2_WT_c, FnP, 2 * FeP, 2 * FeP, q.mfma_scale_c + 4});
    ops.push_back({q.c1_WeT_c, FeP, 2 * FnP, 2 * FnP, q.mfma_scale_c + 6});
  }
  return ops;
}

void pack_weights(rn_potgnn *h, const float *w) {
  const int K = h->cfg.num_atom_types, Fn = h->d.Fn, Fe = h->d.Fe, FnP = h->d.FnP,
            FeP = h->d.FeP, P = h->cfg.num_message_passes;
  PackedLayout &L = h->lay;
  L = PackedLayout();
  L.emb = L.take((size_t)K * Fn);
  L.W2 = L.take((size_t)Fn * Fn);
  L.b2 = L.take(Fn);
  L.W4 = L.take((size_t)Fn * Fn);
  L.b4 = L.take(Fn);
  L.offsets = L.take(FeP);
  L.pass.resize(P);
  for (auto &p : L.pass) {
    p.c1_WnT = L.take((size_t)FnP * 2 * FnP);
    p.c1_WeT = L.take((size_t)FeP * 2 * FnP);
    p.c1_bias = L.take(2 * FnP);
    p.c1n_g = L.take(2 * FnP);
    p.c1n_b = L.take(2 * FnP);
    p.fin_g = L.take(FnP);
    p.fin_b = L.take(FnP);
    p.c2_WT = L.take((size_t)FnP * 2 * FeP);
    p.c2_bias = L.take(2 * FeP);
    p.c2n1_g = L.take(2 * FeP);
    p.c2n1_b = L.take(2 * FeP);
    p.c2n2_g = L.take(FeP);
    p.c2n2_b = L.take(FeP);
    p.c3_WnT = L.take((size_t)FnP * 6 * FeP);
    p.c3_nshift = L.take(6 * FeP);
    p.c3_WeT = L.take((size_t)FeP * 4 * FeP);
    p.c3n1_g = L.take(2 * FeP);
    p.c3n1_b = L.take(2 * FeP);
    p.c3n2_g = L.take(FeP);
    p.c3n2_b = L.take(FeP);
    p.c3n1_gs = L.take(2 * FeP);
    p.c3n1_bs = L.take(2 * FeP);
    p.c2n1_gs = L.take(2 * FeP);
    p.c2n1_bs = L.take(2 * FeP);
    p.c1n_gs = L.take(2 * FnP);
    p.c1n_bs = L.take(2 * FnP);
    p.mfma_scale = L.take(8);
    p.mfma_scale_c = L.take(8);  // (right behind mfma_scale: one copy fetches both after a device-resident step)
    p.t_c3We = L.take((size_t)4 * FeP * FeP);
    p.t_c3Wn = L.take((size_t)6 * FeP * FnP);
    p.t_c2W = L.take((size_t)2 * FeP * FnP);
    p.t_c1We = L.take((size_t)2 * FnP * FeP);
    p.t_c1Wn = L.take((size_t)2 * FnP * FnP);
    p.c3_WeT_c = L.take((size_t)FeP * 4 * FeP);
    p.c3_WnT_c = L.take((size_t)FnP * 6 * FeP);
    p.c3_nshift_c = L.take(6 * FeP);
    p.c2_WT_c = L.take((size_t)FnP * 2 * FeP);
    p.c2_bias_c = L.take(2 * FeP);
    p.c1_WnT_c = L.take((size_t)FnP * 2 * FnP);
    p.c1_WeT_c = L.take((size_t)FeP * 2 * FnP);
    p.c1_bias_c = L.take(2 * FnP);
  }
  const int HP = std::max(FeP, 32);  // readout hidden width: projections emit 32-column tiles
  L.W0T = L.take((size_t)FeP * HP);
  L.b0 = L.take(Fe);
  L.bn_w = L.take(Fe);
  L.bn_b = L.take(Fe);
  L.bn_rm = L.take(Fe);
  L.bn_rv = L.take(Fe);
  L.W3T = L.take((size_t)HP * HP);
  L.b3 = L.take(HP);
  L.W5T = L.take((size_t)HP * 32);
  L.b5 = L.take(32);
  L.ones = L.take(HP);
  L.b0p = L.take(HP);  // bias of readout Linear 0, padded (training-mode forward)
  L.t_W0 = L.take((size_t)HP * FeP);
  L.t_W3 = L.take((size_t)HP * HP);
  L.t_W5 = L.take((size_t)32 * HP);
  L.ro_mfma_scale = L.take(8);
  L.node_table = L.take((size_t)K * FnP);
  L.scale0 = L.take(HP);
  L.shift0 = L.take(HP);

  std::vector<float> &o = h->packed;
  o.assign(L.total, 0.0f);
  const float *c = w;  // cursor over the state_dict-ordered blob
  auto copy = [&](size_t dst, size_t n) {
    std::memcpy(&o[dst], c, n * sizeof(float));
    c += n;
  };
  copy(L.emb, (size_t)K * Fn);
  copy(L.W2, (size_t)Fn * Fn);
  copy(L.b2, Fn);
  copy(L.W4, (size_t)Fn * Fn);
  copy(L.b4, Fn);
  copy(L.offsets, Fe);  // "_edge_embedding.offset"
  // node blocks (all passes) come first in the state dict, then edge blocks
  for (int p = 0; p < P; ++p) {
    auto &q = L.pass[p];
    const float *W = c;  // c1_linear.weight [2Fn, Fn+Fe]
    for (int r = 0; r < 2 * Fn; ++r) {
      const int col = gated_col(r, Fn, FnP);
      for (int k = 0; k < Fn; ++k) o[q.c1_WnT + (size_t)k * 2 * FnP + col] = W[r * (Fn + Fe) + k];
      for (int k = 0; k < Fe; ++k)
        o[q.c1_WeT + (size_t)k * 2 * FnP + col] = W[r * (Fn + Fe) + Fn + k];
    }
    c += (size_t)2 * Fn * (Fn + Fe);
    for (int r = 0; r < 2 * Fn; ++r) o[q.c1_bias + gated_col(r, Fn, FnP)] = c[r];
    c += 2 * Fn;
    for (int r = 0; r < 2 * Fn; ++r) o[q.c1n_g + gated_col(r, Fn, FnP)] = c[r];
    c += 2 * Fn;
    for (int r = 0; r < 2 * Fn; ++r) o[q.c1n_b + gated_col(r, Fn, FnP)] = c[r];
    c += 2 * Fn;
    copy(q.fin_g, Fn);
    copy(q.fin_b, Fn);
  }
  for (int p = 0; p < P; ++p) {
    auto &q = L.pass[p];
    const float *W2 = c;  // c2_linear.weight [2Fe, Fn]
    for (int r = 0; r < 2 * Fe; ++r) {
      const int col = gated_col(r, Fe, FeP);
      for (int k = 0; k < Fn; ++k) o[q.c2_WT + (size_t)k * 2 * FeP + col] = W2[r * Fn + k];
    }
    c += (size_t)2 * Fe * Fn;
    for (int r = 0; r < 2 * Fe; ++r) o[q.c2_bias + gated_col(r, Fe, FeP)] = c[r];
    c += 2 * Fe;
    const float *W3 = c;  // c3_linear.weight [2Fe, 3Fn+2Fe]: [n_i | n_j | n_k | e_slot5 | e_slot6]
    const int ld = 3 * Fn + 2 * Fe;
    for (int r = 0; r < 2 * Fe; ++r) {
      const int col = gated_col(r, Fe, FeP);
      for (int blk = 0; blk < 3; ++blk)
        for (int k = 0; k < Fn; ++k)
          o[q.c3_WnT + (size_t)k * 6 * FeP + blk * 2 * FeP + col] = W3[r * ld + blk * Fn + k];
      for (int blk = 0; blk < 2; ++blk)
        for (int k = 0; k < Fe; ++k)
          o[q.c3_WeT + (size_t)k * 4 * FeP + blk * 2 * FeP + col] =
              W3[r * ld + 3 * Fn + blk * Fe + k];
    }
    c += (size_t)2 * Fe * ld;
    for (int r = 0; r < 2 * Fe; ++r) o[q.c3_nshift + 2 * FeP + gated_col(r, Fe, FeP)] = c[r];
    c += 2 * Fe;
    for (int r = 0; r < 2 * Fe; ++r) o[q.c2n1_g + gated_col(r, Fe, FeP)] = c[r];
    c += 2 * Fe;
    for (int r = 0; r < 2 * Fe; ++r) o[q.c2n1_b + gated_col(r, Fe, FeP)] = c[r];
    c += 2 * Fe;
    for (int r = 0; r < 2 * Fe; ++r) o[q.c3n1_g + gated_col(r, Fe, FeP)] = c[r];
    c += 2 * Fe;
    for (int r = 0; r < 2 * Fe; ++r) o[q.c3n1_b + gated_col(r, Fe, FeP)] = c[r];
    c += 2 * Fe;
    copy(q.c2n2_g, Fe);
    copy(q.c2n2_b, Fe);
    copy(q.c3n2_g, Fe);
    copy(q.c3n2_b, Fe);
    for (int k = 0; k < FeP; ++k) {  // sigmoid(f) tanh(c) through exp2: exp2(-log2e f), exp2(2 log2e c)
      o[q.c3n1_gs + k] = -1.4426950408889634f * o[q.c3n1_g + k];
      o[q.c3n1_bs + k] = -1.4426950408889634f * o[q.c3n1_b + k];
      o[q.c3n1_gs + FeP + k] = 2.0f * 1.4426950408889634f * o[q.c3n1_g + FeP + k];
      o[q.c3n1_bs + FeP + k] = 2.0f * 1.4426950408889634f * o[q.c3n1_b + FeP + k];
      o[q.c2n1_gs + k] = -1.4426950408889634f * o[q.c2n1_g + k];
      o[q.c2n1_bs + k] = -1.4426950408889634f * o[q.c2n1_b + k];
      o[q.c2n1_gs + FeP + k] = 2.0f * 1.4426950408889634f * o[q.c2n1_g + FeP + k];
      o[q.c2n1_bs + FeP + k] = 2.0f * 1.4426950408889634f * o[q.c2n1_b + FeP + k];
    }
    for (int k = 0; k < FnP; ++k) {
      o[q.c1n_gs + k] = -1.4426950408889634f * o[q.c1n_g + k];
      o[q.c1n_bs + k] = -1.4426950408889634f * o[q.c1n_b + k];
      o[q.c1n_gs + FnP + k] = 2.0f * 1.4426950408889634f * o[q.c1n_g + FnP + k];
      o[q.c1n_bs + FnP + k] = 2.0f * 1.4426950408889634f * o[q.c1n_b + FnP + k];
    }
  }
  {  // readout
    const float *W0 = c;
    for (int r = 0; r < Fe; ++r)
      for (int k = 0; k < Fe; ++k) o[L.W0T + (size_t)k * HP + r] = W0[r * Fe + k];
    c += (size_t)Fe * Fe;
    copy(L.b0, Fe);
    std::memcpy(&o[L.b0p], &o[L.b0], Fe * sizeof(float));
    copy(L.bn_w, Fe);
    copy(L.bn_b, Fe);
    copy(L.bn_rm, Fe);
    copy(L.bn_rv, Fe);
    const float *W3 = c;
    for (int r = 0; r < Fe; ++r)
      for (int k = 0; k < Fe; ++k) o[L.W3T + (size_t)k * HP + r] = W3[r * Fe + k];
    c += (size_t)Fe * Fe;
    copy(L.b3, Fe);
    const float *W5 = c;
    for (int r = 0; r < 12; ++r)
      for (int k = 0; k < Fe; ++k) o[L.W5T + (size_t)k * 32 + r] = W5[r * Fe + k];
    c += (size_t)12 * Fe;
    copy(L.b5, 12);
  }
  for (int i = 0; i < HP; ++i) o[L.ones + i] = 1.0f;
  for (const MfmaScaleOp &m : mfma_scale_ops(h)) {  // power-of-two prescales of the split-f16 products
    float mx = 0.0f;
    for (int k = 0; k < m.K; ++k)
      for (int n = 0; n < m.N; ++n) mx = std::max(mx, std::fabs(o[m.src + (size_t)k * m.ld + n]));
    const float sc = mfma_prescale(mx);
    o[m.dst] = sc;
    o[m.dst + 1] = 1.0f / sc;
  }
  for (const CentreOp &c : centre_ops(h)) {  // row-centred copies (float64 means; padded columns stay zero)
    const int bw = 2 * c.FP;
    for (int k = 0; k < c.K; ++k)
      for (int b = 0; b < c.N / bw; ++b) {
        const size_t base = (size_t)k * c.N + (size_t)b * bw;
        double sum = 0;
        for (int hh = 0; hh < 2; ++hh)
          for (int col = 0; col < c.F; ++col) sum += (double)o[c.src + base + hh * c.FP + col];
        const double mean = sum / (2.0 * c.F);
        for (int hh = 0; hh < 2; ++hh)
          for (int col = 0; col < c.FP; ++col)
            o[c.dst + base + hh * c.FP + col] = col < c.F ? (float)((double)o[c.src + base + hh * c.FP + col] - mean) : 0.0f;
      }
  }
  for (const MfmaScaleOp &m : centred_scale_ops(h)) {
    float mx = 0.0f;
    for (int k = 0; k < m.K; ++k)
      for (int n = 0; n < m.N; ++n) mx = std::max(mx, std::fabs(o[m.src + (size_t)k * m.ld + n]));
    const float sc = mfma_prescale(mx);
    o[m.dst] = sc;
    o[m.dst + 1] = 1.0f / sc;
  }
  // transposed copies: src [K][N] (row stride N) -> dst [N][K]
  auto transpose = [&](size_t src, int Kd, int Nd, size_t dst) {
    for (int k = 0; k < Kd; ++k)
      for (int n = 0; n < Nd; ++n) o[dst + (size_t)n * Kd + k] = o[src + (size_t)k * Nd + n];
  };
  for (auto &q : L.pass) {
    transpose(q.c3_WeT, FeP, 4 * FeP, q.t_c3We);
    transpose(q.c3_WnT, FnP, 6 * FeP, q.t_c3Wn);
    transpose(q.c2_WT, FnP, 2 * FeP, q.t_c2W);
    transpose(q.c1_WeT, FeP, 2 * FnP, q.t_c1We);
    transpose(q.c1_WnT, FnP, 2 * FnP, q.t_c1Wn);
  }
  transpose(L.W0T, FeP, HP, L.t_W0);
  transpose(L.W3T, HP, HP, L.t_W3);
  transpose(L.W5T, HP, 32, L.t_W5);
}

// Frames per device work chunk in precision T.  The workspace budget is counted in bytes, so
// the float64 lanes (created on first use, next to the float32 ones) take half the frames.
bool lean_workspace(const rn_potgnn *h);
size_t per_structure_elems(const rn_potgnn *h, bool lean);
// The float32 chunk may have been sized from the LEAN workspace of the fused / narrow pipelines; float64 always runs the
// unfused chain on the full-width buffers, so its chunk is counted from that layout (8 bytes per element) against its own
// allowance (rn_potgnn::f64_budget; with an explicit chunk: the bytes the float32 lanes were given).
template <typename T>
int chunk_frames(const rn_potgnn *h) {
  if (sizeof(T) == 4) return h->chunk;
  const size_t budget = std::max(h->f64_budget, (size_t)h->chunk * per_structure_elems(h, lean_workspace(h)) * sizeof(float));
  const size_t per64 = per_structure_elems(h, false) * sizeof(double);
  return (int)std::max<size_t>(1, std::min<size_t>((size_t)h->chunk, budget / std::max<size_t>(per64, 1)));
}

template <typename T>
Precision<T> &prec(rn_potgnn *h);
template <>
Precision<float> &prec<float>(rn_potgnn *h) { return h->f32; }
template <>
Precision<double> &prec<double>(rn_potgnn *h) { return h->f64; }

// Sum `n` device doubles over the ranks of a data-parallel training run (no-op without a reducer).
void reduce_over_ranks(rn_potgnn *h, double *dev, size_t n, hipStream_t st) {
  if (!h->reducer) return;
  std::vector<double> host(n);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipMemcpy(host.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost));
  if (h->reducer(host.data(), (int64_t)n, h->reducer_ctx) != 0)
    throw HipError{hipErrorUnknown, "the statistics reducer (data-parallel training) failed"};
  HIP_TRY(hipMemcpy(dev, host.data(), n * sizeof(double), hipMemcpyHostToDevice));
}

// The fused and the narrow pipelines keep no per-edge projection in HBM: of the two projection buffers only
// the [S*E, 32] readout output (bufA) is left.  Taped runs (training, Jacobian) size the full-width buffers
// for their own, much smaller, batch (ensure_tape).
bool lean_workspace(const rn_potgnn *h) {
  return (h->use_fused && h->use_node_fused && h->use_readout_fused) || h->use_narrow;
}
size_t bufA_width(const rn_potgnn *h, bool lean) {
  return lean ? 32 : std::max<size_t>(std::max(2 * h->d.FnP, 2 * h->d.FeP), 32);
}
size_t per_structure_elems(const rn_potgnn *h, bool lean) {
  const size_t N = h->cfg.num_atoms, E = h->cfg.num_edges;
  const size_t FnP = h->d.FnP, FeP = h->d.FeP;
  return E * 4 + 2 * N * FnP + 2 * E * FeP + N * 2 * FnP + N * 6 * FeP + E * bufA_width(h, lean) +
         (lean ? 0 : E * 4 * FeP) + ((h->use_edge2 || h->use_edge3) ? E * FeP : 0);
}

// May the fused kernels run their matrix products as split-f16 MFMAs (device_utils.hpp)?
// Weights: always -- each block is prescaled by a power of two into f16's normal range (mfma_prescale),
// so only a non-finite weight refuses.  Activations are split unscaled, which is exact to 22 bits while
// they stay well inside f16's range: edge rows (Gaussian basis, then tanh outputs), updated node rows
// and their products are bounded by 1 by construction; the two hidden layers of the readout MLP
// (shifted softplus, unbounded above) are bounded here from the weights, for |edge| <= 1:
//   |h1_n| <= |scale0_n| sum_k |W0[n][k]| + |shift0_n|,   |h2_n| <= B1 sum_k |W3[n][k]| + |b3_n|.
// Beyond 3e4 (f16 overflows at 65504) the handle falls back to the exact-f32 MFMA instantiations.
bool mfma_f16_range_ok(const rn_potgnn *h) {
  const PackedLayout &L = h->lay;
  const float *o = h->packed.data();
  const int Fe = h->d.Fe, HP = std::max(h->d.FeP, 32);
  // While the device weights are ahead of `packed` (device-resident training) only the readout block, c3_norm_1 and the
  // prescale pairs have been fetched: the finiteness of the weight blocks is then read off the pairs, which the device
  // refresh sets to NaN for a block with a non-finite entry (kernels_train.hip, kind 2)
  for (const MfmaScaleOp &m : mfma_scale_ops(h)) {
    if (!std::isfinite(o[m.dst]) || !std::isfinite(o[m.dst + 1])) return false;
    if (h->host_stale) continue;
    for (int k = 0; k < m.K; ++k)
      for (int n = 0; n < m.N; ++n)
        if (!std::isfinite(o[m.src + (size_t)k * m.ld + n])) return false;
  }
  for (const MfmaScaleOp &m : centred_scale_ops(h))
    if (!std::isfinite(o[m.dst]) || !std::isfinite(o[m.dst + 1])) return false;
  const double ln2 = 0.6931471805599453;
  double b1 = ln2, b2 = ln2;
  for (int n = 0; n < Fe; ++n) {  // BatchNorm(eval) folded as setup_kernel does
    const double sc = (double)o[L.bn_w + n] / std::sqrt((double)o[L.bn_rv + n] + 1e-5);
    const double sh = ((double)o[L.b0 + n] - (double)o[L.bn_rm + n]) * sc + (double)o[L.bn_b + n];
    double sum = 0;
    for (int k = 0; k < Fe; ++k) sum += std::fabs((double)o[L.W0T + (size_t)k * HP + n]);
    b1 = std::max(b1, std::fabs(sc) * sum + std::fabs(sh));
  }
  for (int n = 0; n < Fe; ++n) {
    double sum = 0;
    for (int k = 0; k < Fe; ++k) sum += std::fabs((double)o[L.W3T + (size_t)k * HP + n]);
    b2 = std::max(b2, sum * b1 + std::fabs((double)o[L.b3 + n]));
  }
  return std::isfinite(b1) && std::isfinite(b2) && b1 <= 3.0e4 && b2 <= 3.0e4;
}
void refresh_mfma_mode(rn_potgnn *h) {
  const bool ok = mfma_f16_range_ok(h);
  h->mfma_f16 = h->mfma_f16_requested && ok;
  h->mfma_range_fallback = h->mfma_f16_requested && !ok;
}

// Per pass: may the EdgeBlock's triplet loop fold c3_norm_1's scale into its operands and
// drop the gate's overflow clamp (edge_agg_kernel, FASTG)?  Needs every gamma of a real
// column away from zero (the loop divides by it) and the gate arguments provably small:
// a LayerNorm output is at most sqrt(2Fe - 1) in magnitude.
template <typename T>
void refresh_pass_flags(rn_potgnn *h) {
  Precision<T> &P = prec<T>(h);
  const PackedLayout &L = h->lay;
  const int Fe = h->d.Fe, FeP = h->d.FeP;
  const bool off = getenv("RN_POTGNN_NO_FASTG") && atoi(getenv("RN_POTGNN_NO_FASTG")) != 0;
  for (size_t p = 0; p < L.pass.size() && p < P.pass.size(); ++p) {
    const float *gam = h->packed.data() + L.pass[p].c3n1_g, *bet = h->packed.data() + L.pass[p].c3n1_b;
    const double xmax = std::sqrt(2.0 * Fe) * 1.02;
    bool ok = !off;
    for (int half = 0; half < 2 && ok; ++half)
      for (int k = 0; k < Fe && ok; ++k) {
        const double g = std::fabs((double)gam[half * FeP + k]), b = std::fabs((double)bet[half * FeP + k]);
        const double arg = (g * xmax + b) * 2.0 * 1.4426950408889634;
        ok = std::isfinite(g) && std::isfinite(b) && g >= 1e-5 && arg < 60.0;
      }
    // bit 0: fused EdgeBlock kernel; bit 1: unfused edge_agg_kernel -- there only with a
    // single lane: its 170 registers per lane shut the other lane's projection workgroups
    // out of the CU (150 do not), which costs more than the shorter loop gains
    P.pass[p].c3_fast = ok ? (h->num_lanes == 1 ? 3 : 1) : 0;
  }
}

template <typename T>
void upload_weights(rn_potgnn *h) {  // (re)upload the packed weights and redo the device precompute
  Precision<T> &P = prec<T>(h);
  const PackedLayout &L = h->lay;
  std::vector<T> host(h->packed.size());
  for (size_t i = 0; i < host.size(); ++i) host[i] = (T)h->packed[i];
  P.weights.ensure(host.size() * sizeof(T));
  HIP_TRY(hipMemcpy(P.weights.p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  T *w = P.weights.template as<T>();
  launch_setup<T>(w + L.emb, w + L.W2, w + L.b2, w + L.W4, w + L.b4, h->cfg.num_atom_types, h->d,
                  w + L.node_table, w + L.b0, w + L.bn_w, w + L.bn_b, w + L.bn_rm, w + L.bn_rv,
                  w + L.scale0, w + L.shift0, P.lanes[0].stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(P.lanes[0].stream));
  refresh_pass_flags<T>(h);
}

template <typename T>
void ensure_precision(rn_potgnn *h) {
  Precision<T> &P = prec<T>(h);
  if (P.ready) return;
  const PackedLayout &L = h->lay;
  std::vector<T> host(h->packed.size());
  for (size_t i = 0; i < host.size(); ++i) host[i] = (T)h->packed[i];
  P.weights.ensure(host.size() * sizeof(T));
  HIP_TRY(hipMemcpy(P.weights.p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  T lat[9];
  for (int i = 0; i < 9; ++i) lat[i] = (T)h->lattice[i];
  P.lattice.ensure(sizeof(lat));
  HIP_TRY(hipMemcpy(P.lattice.p, lat, sizeof(lat), hipMemcpyHostToDevice));
  T *w = P.weights.template as<T>();
  P.pass.resize(L.pass.size());
  for (size_t p = 0; p < L.pass.size(); ++p) {
    const auto &q = L.pass[p];
    PassW<T> &o = P.pass[p];
    o.c1_WnT = w + q.c1_WnT;
    o.c1_WeT = w + q.c1_WeT;
    o.c1_bias = w + q.c1_bias;
    o.c1_norm = {w + q.c1n_g, w + q.c1n_b};
    o.final_norm = {w + q.fin_g, w + q.fin_b};
    o.c2_WT = w + q.c2_WT;
    o.c2_bias = w + q.c2_bias;
    o.c2_norm_1 = {w + q.c2n1_g, w + q.c2n1_b};
    o.c2_norm_2 = {w + q.c2n2_g, w + q.c2n2_b};
    o.c3_WnT = w + q.c3_WnT;
    o.c3_nshift = w + q.c3_nshift;
    o.c3_WeT = w + q.c3_WeT;
    o.c3_norm_1 = {w + q.c3n1_g, w + q.c3n1_b};
    o.c3_norm_2 = {w + q.c3n2_g, w + q.c3n2_b};
    o.c3_norm_1s = {w + q.c3n1_gs, w + q.c3n1_bs};
    o.c2_norm_1s = {w + q.c2n1_gs, w + q.c2n1_bs};
    o.c1_norm_s = {w + q.c1n_gs, w + q.c1n_bs};
    o.mfma_scale = w + q.mfma_scale;
    o.c3_WeT_c = w + q.c3_WeT_c;
    o.c3_WnT_c = w + q.c3_WnT_c;
    o.c3_nshift_c = w + q.c3_nshift_c;
    o.c2_WT_c = w + q.c2_WT_c;
    o.c2_bias_c = w + q.c2_bias_c;
    o.mfma_scale_c = w + q.mfma_scale_c;
    o.c1_WnT_c = w + q.c1_WnT_c;
    o.c1_WeT_c = w + q.c1_WeT_c;
    o.c1_bias_c = w + q.c1_bias_c;
  }
  refresh_pass_flags<T>(h);
  P.ro = {w + L.W0T, w + L.scale0, w + L.shift0, w + L.W3T, w + L.b3, w + L.W5T, w + L.b5, w + L.ro_mfma_scale};
  P.offsets = w + L.offsets;
  P.node_table = w + L.node_table;
  P.ones = w + L.ones;

  const size_t N = h->cfg.num_atoms, E = h->cfg.num_edges, S = chunk_frames<T>(h);
  const size_t FnP = h->d.FnP, FeP = h->d.FeP;
  const bool lean = sizeof(T) == 4 && lean_workspace(h);
  const size_t bufA = bufA_width(h, lean);
  for (int l = 0; l < h->num_lanes; ++l) {
    Lane<T> &ln = P.lanes[l];
    HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    ln.unit4.ensure(S * E * 4 * sizeof(T));
    for (int i = 0; i < 2; ++i) {
      ln.node[i].ensure(S * N * FnP * sizeof(T));
      ln.edge[i].ensure(S * E * FeP * sizeof(T));
    }
    ln.npc1.ensure(S * N * 2 * FnP * sizeof(T));
    ln.np3.ensure(S * N * 6 * FeP * sizeof(T));
    ln.bufA.ensure(S * E * bufA * sizeof(T));
    if (!lean) ln.bufB.ensure(S * E * 4 * FeP * sizeof(T));
    if (sizeof(T) == 4 && (h->use_edge2 || h->use_edge3)) ln.c2.ensure(S * E * FeP * sizeof(T));
  }
  if (h->keep_stages) {
    const int np = h->cfg.num_message_passes + 1;
    P.snap_node.resize(np);
    P.snap_edge.resize(np);
    for (int i = 0; i < np; ++i) {
      P.snap_node[i].ensure(S * N * FnP * sizeof(T));
      P.snap_edge[i].ensure(S * E * FeP * sizeof(T));
    }
  }
  // frame-independent device precompute
  launch_setup<T>(w + L.emb, w + L.W2, w + L.b2, w + L.W4, w + L.b4, h->cfg.num_atom_types, h->d,
                  w + L.node_table, w + L.b0, w + L.bn_w, w + L.bn_b, w + L.bn_rm, w + L.bn_rv,
                  w + L.scale0, w + L.shift0, P.lanes[0].stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(P.lanes[0].stream));
  P.ready = true;
}

struct Timer {
  rn_potgnn *h;
  hipStream_t st;
  int kid;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  Timer(rn_potgnn *h_, hipStream_t st_, int kid_) : h(h_), st(st_), kid(kid_) {
    // 1 = every kernel, 100 + k = kernel k only, 1000 + mask = the kernels whose bit is set
    on = h->profiling == 1 || (h->profiling >= 100 && h->profiling < 1000 && h->profiling - 100 == kid) ||
         (h->profiling >= 1000 && (((h->profiling - 1000) >> kid) & 1));
    if (on) {
      HIP_TRY(hipEventCreate(&a));
      HIP_TRY(hipEventCreate(&b));
      HIP_TRY(hipEventRecord(a, st));
    }
  }
  ~Timer() noexcept(false) {
    if (on) {
      (void)hipEventRecord(b, st);
      h->timed.push_back({kid, a, b});
    }
    if (h->debug_sync) {
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) throw HipError{e, kKernelNames[kid]};
    }
  }
};

void resolve_timers(rn_potgnn *h) {
  for (auto &t : h->timed) {
    float ms = 0;
    if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
      h->k_ms[t.kid] += ms;
      h->k_launches[t.kid] += 1;
    }
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  h->timed.clear();
}

// One chunk of `S` frames on one lane, split into stages so that two chunks can be
// interleaved (see forward_device).  Mirrors PotGNN.forward (_gnn.py:641-665).
template <typename T>
struct ChunkRun {
  rn_potgnn *h;
  Lane<T> *ln;
  const double *d_pos;
  const float *d_pos32 = nullptr;  // float32 evaluations: the chunk's positions as float32 (then d_pos is null)
  int S;
  double *d_alpha;
  float *d_vec6;
  double *d_alpha_raw;
  const T *d_lat = nullptr;  // per-frame lattices [S][9] of this chunk, or null: the reference structure's
  const int *d_types = nullptr;  // per-frame atom types [S][N] of this chunk, or null: the reference structure's
  int cur = 0;
  T *node[2], *edge[2], *unit4, *npc1, *np3, *bufA, *bufB, *c2;
  int64_t MN, ME;

  ChunkRun(rn_potgnn *h_, Lane<T> &l, const double *pos, int S_, double *alpha, float *vec6,
           double *alpha_raw)
      : h(h_), ln(&l), d_pos(pos), S(S_), d_alpha(alpha), d_vec6(vec6), d_alpha_raw(alpha_raw) {
    node[0] = l.node[0].template as<T>();
    node[1] = l.node[1].template as<T>();
    edge[0] = l.edge[0].template as<T>();
    edge[1] = l.edge[1].template as<T>();
    unit4 = l.unit4.template as<T>();
    npc1 = l.npc1.template as<T>();
    np3 = l.np3.template as<T>();
    bufA = l.bufA.template as<T>();
    bufB = l.bufB.template as<T>();
    c2 = l.c2.template as<T>();
    MN = (int64_t)S * h->g.N;
    ME = (int64_t)S * h->g.E;
  }
  hipStream_t st() const { return ln->stream; }
  // With the tape on, the embeddings after pass p are written straight into tape slot p+1 (the
  // ping-pong pair is re-pointed pass by pass), so recording costs no copies.
  void target_tape(int slot, int which) {
    Precision<T> &P = prec<T>(h);
    if (!P.tape_on) return;
    node[which] = P.tape_node[slot].template as<T>();
    edge[which] = P.tape_edge[slot].template as<T>();
  }
  void snapshot(int p) {
    Precision<T> &P = prec<T>(h);
    if (!h->keep_stages) return;
    HIP_TRY(hipMemcpyAsync(P.snap_node[p].p, node[cur], (size_t)MN * h->d.FnP * sizeof(T),
                           hipMemcpyDeviceToDevice, st()));
    HIP_TRY(hipMemcpyAsync(P.snap_edge[p].p, edge[cur], (size_t)ME * h->d.FeP * sizeof(T),
                           hipMemcpyDeviceToDevice, st()));
  }

  // geometry + radial basis, initial node embedding
  void begin() {
    Precision<T> &P = prec<T>(h);
    target_tape(0, 0);
    {
      Timer t(h, st(), K_GEOM);
      bool done = false;
      if constexpr (sizeof(T) == 4) {
        if (d_pos32) {
          const float *lat = d_lat ? d_lat : P.lattice.template as<T>();
          if (pair_rows()) launch_geom_rbf_pairs_pos32(d_pos32, S, h->g, lat, d_lat ? 9 : 0, P.offsets, (T)h->cfg.gauss_coefficient, h->d, unit4, edge[0], st());
          else launch_geom_rbf_pos32(d_pos32, S, h->g, lat, d_lat ? 9 : 0, P.offsets, (T)h->cfg.gauss_coefficient, h->d, unit4, edge[0], st(), narrow());
          done = true;
        } else if (pair_rows()) {
          launch_geom_rbf_pairs(d_pos, S, h->g, d_lat ? d_lat : P.lattice.template as<T>(), d_lat ? 9 : 0, P.offsets,
                                (T)h->cfg.gauss_coefficient, h->d, unit4, edge[0], st());
          done = true;
        }
      }
      if (!done)
        launch_geom_rbf<T>(d_pos, S, h->g, d_lat ? d_lat : P.lattice.template as<T>(), d_lat ? 9 : 0,
                           P.offsets, (T)h->cfg.gauss_coefficient, h->d, unit4, edge[0], st(), narrow());
      h->last_in_order = narrow();  // (kernels_narrow.hip: edge rows in (b, a) order; rn_potgnn_debug_stage undoes it)
    }
    {
      Timer t(h, st(), K_NODE_INIT);
      launch_node_init<T>(P.node_table, S, h->g, h->d, node[0], d_types, st());
    }
    cur = 0;
    snapshot(0);
  }

  // "G" stage of pass p: NodeBlock + every dense projection the EdgeBlock needs
  // (matrix pipe / HBM bound)
  void stage_project(int p) {
    const PassW<T> &w = prec<T>(h).pass[p];
    const Graph &g = h->g;
    const Dims d = h->d;
    const int nxt = cur ^ 1;
    target_tape(p + 1, nxt);
    // Y = X W (+ bias): float32 with the split-f16 products enabled -> rowgemm_split_kernel where it serves the shape
    // (K a multiple of 64), else the exact-f32 MFMA projection kernel
    auto project = [&](const T *X, int64_t R, int K, const T *WT, int NOUT, T *Y, const T *bias, int amode, const T *nd) {
      if constexpr (sizeof(T) == 4) {
        if (h->mfma_f16 && h->split_projections &&
            launch_rowgemm_split(X, K, K, R, WT, NOUT, Y, false, bias, amode, nd, g, st()))
          return;
      }
      launch_rowgemm<T>(X, R, K, WT, NOUT, Y, nullptr, bias, false, amode, nd, g, st());
    };
    if constexpr (sizeof(T) == 4) {
      if (narrow()) {  // the whole NodeBlock, projections included, in one launch
        Timer t(h, st(), K_NODE_AGG);
        launch_node_narrow(edge[cur], node[cur], node[nxt], S, g, d, w, st());
        return;
      }
    }
    {
      Timer t(h, st(), K_PROJ_NODE);
      if (node_centred()) project(node[cur], MN, d.FnP, w.c1_WnT_c, 2 * d.FnP, npc1, w.c1_bias_c, 0, nullptr);  // zero row mean
      else project(node[cur], MN, d.FnP, w.c1_WnT, 2 * d.FnP, npc1, w.c1_bias, 0, nullptr);
    }
    bool node_fused = false;
    if constexpr (sizeof(T) == 4) {
      if (fused() && h->use_node_fused) {  // c1 edge projection + aggregation in one launch
        Timer t(h, st(), K_NODE_AGG);
        if (node_centred() && g.na_num > 0) launch_node_atom(edge[cur], node[cur], npc1, node[nxt], S, g, d, w, st(), pair_rows());
        else launch_node_fused(edge[cur], node[cur], npc1, node[nxt], S, g, d, w, h->mfma_f16, node_centred(), st());
        node_fused = true;
      }
    }
    if (!node_fused) {
      {
        Timer t(h, st(), K_PROJ_EDGE_C1);
        project(edge[cur], ME, d.FeP, w.c1_WeT, 2 * d.FnP, bufA, nullptr, 0, nullptr);
      }
      {
        Timer t(h, st(), K_NODE_AGG);
        launch_node_agg<T>(npc1, bufA, node[cur], node[nxt], S, g, d, w, st());
      }
    }
    {  // EdgeBlock uses the UPDATED node embedding (_gnn.py:649-650)
      Timer t(h, st(), K_PROJ_NODE);
      if (role_split(w)) project(node[nxt], MN, d.FnP, w.c3_WnT_c, 6 * d.FeP, np3, w.c3_nshift_c, 0, nullptr);  // zero row mean
      else project(node[nxt], MN, d.FnP, w.c3_WnT, 6 * d.FeP, np3, w.c3_nshift, 0, nullptr);
    }
    if constexpr (sizeof(T) == 4) {
#if RN_EXPERIMENTS
      if (fused() && (h->use_edge2 || (h->use_edge3 && edge3_applicable(w, h->mfma_f16)))) {  // c2 branch of the EdgeBlock, one finished row per edge
        Timer t(h, st(), K_PROJ_C2);
        launch_edge_c2(node[nxt], c2, S, g, d, w, h->mfma_f16, st());
      }
#endif
    }
    if (!fused()) {
      {
        Timer t(h, st(), K_PROJ_EDGE_C3);
        project(edge[cur], ME, d.FeP, w.c3_WeT, 4 * d.FeP, bufB, nullptr, 0, nullptr);
      }
      {
        Timer t(h, st(), K_PROJ_C2);
        project(nullptr, ME, d.FnP, w.c2_WT, 2 * d.FeP, bufA, w.c2_bias, 1, node[nxt]);
      }
    }
  }

  // "V" stage of pass p: triplet aggregation of the EdgeBlock (VALU bound)
  void stage_aggregate(int p) {
    const PassW<T> &w = prec<T>(h).pass[p];
    const int nxt = cur ^ 1;
    {
      Timer t(h, st(), K_EDGE_AGG);
      if constexpr (sizeof(T) == 4) {
        if (narrow()) launch_edge_narrow(edge[cur], edge[nxt], node[nxt], S, h->g, h->d, w, st());
#if RN_EXPERIMENTS
        else if (fused() && h->use_edge3 && edge3_applicable(w, h->mfma_f16))
          launch_edge3(edge[cur], edge[nxt], np3, c2, tape_agg(p), S, h->g, h->d, w, st());
        else if (fused() && h->use_edge2)
          launch_edge2(edge[cur], edge[nxt], np3, c2, tape_agg(p), S, h->g, h->d, w, h->mfma_f16, st());
#endif
        else if (role_split(w)) launch_edge_ps(edge[cur], edge[nxt], node[nxt], np3, tape_agg(p), S, h->g, h->d, w, h->ps_fail.as<int>(), st(), pair_rows(), h->mfma_f16);
#if RN_EXPERIMENTS
        else if (fused() && getenv("RN_POTGNN_EDGE_FRAME") && atoi(getenv("RN_POTGNN_EDGE_FRAME")) != 0)  // the retired per-frame kernel
          launch_edge_fused(edge[cur], edge[nxt], node[nxt], np3, tape_agg(p), S, h->g, h->d, w, h->mfma_f16, st());
#endif
        else if (fused()) edge_unfused_in_blocks(p);
        else launch_edge_agg<T>(bufB, np3, bufA, edge[cur], edge[nxt], S, h->g, h->d, w, tape_agg(p), st());
      } else {
        launch_edge_agg<T>(bufB, np3, bufA, edge[cur], edge[nxt], S, h->g, h->d, w, tape_agg(p), st());
      }
    }
    cur = nxt;
    snapshot(p + 1);
  }

  // A pass of the FUSED pipeline that the role-specialised EdgeBlock does not serve (a graph its ring refuses, a pass
  // without the folded gate scale, exact-f32 products, RN_POTGNN_EDGE_PS=0 / RN_POTGNN_TAPE_PS=0): the unfused EdgeBlock --
  // the two per-edge projections + edge_agg_kernel, what the float64 instantiation always runs -- block of frames by block
  // of frames, because the fused pipeline's lean workspace holds no per-edge projection buffers (6 FeP floats per edge).
  // Through round 4 these passes took the per-frame fused kernel (now experiments/kernels_edge_frame.hip).
  void edge_unfused_in_blocks(int p) {
    const PassW<T> &w = prec<T>(h).pass[p];
    const Graph &g = h->g;
    const Dims d = h->d;
    const int nxt = cur ^ 1;
    const size_t per_frame = (size_t)g.E * (size_t)(6 * d.FeP) * sizeof(T);
    int block = (int)std::max<size_t>(1, std::min<size_t>((size_t)S, ((size_t)1 << 30) / std::max<size_t>(per_frame, 1)));
    if (const char *e = getenv("RN_POTGNN_UNFUSED_BLOCK_FRAMES")) block = std::max(1, std::min(block, atoi(e)));  // test knob: several blocks on a small batch
    ln->fbA.ensure((size_t)block * g.E * 2 * d.FeP * sizeof(T));
    ln->fbB.ensure((size_t)block * g.E * 4 * d.FeP * sizeof(T));
    T *fa = ln->fbA.template as<T>(), *fb = ln->fbB.template as<T>();
    T *agg = tape_agg(p);
    for (int s0 = 0; s0 < S; s0 += block) {
      const int sb = std::min(block, S - s0);
      const int64_t rows = (int64_t)sb * g.E;
      const T *e_in = edge[cur] + (size_t)s0 * g.E * d.FeP, *nd = node[nxt] + (size_t)s0 * g.N * d.FnP;
      auto project = [&](const T *X, int K, const T *WT, int NOUT, T *Y, const T *bias, int amode, const T *ndp) {
        if constexpr (sizeof(T) == 4) {
          if (h->mfma_f16 && h->split_projections && launch_rowgemm_split(X, K, K, rows, WT, NOUT, Y, false, bias, amode, ndp, g, st())) return;
        }
        launch_rowgemm<T>(X, rows, K, WT, NOUT, Y, nullptr, bias, false, amode, ndp, g, st());
      };
      project(e_in, d.FeP, w.c3_WeT, 4 * d.FeP, fb, nullptr, 0, nullptr);
      project(nullptr, d.FnP, w.c2_WT, 2 * d.FeP, fa, w.c2_bias, 1, nd);
      launch_edge_agg<T>(fb, np3 + (size_t)s0 * g.N * 6 * d.FeP, fa, e_in, edge[nxt] + (size_t)s0 * g.E * d.FeP, sb, g, d, w,
                         agg ? agg + (size_t)s0 * g.E * d.FeP : nullptr, st());
    }
  }

  // readout MLP (_gnn.py:532-539): bufA <- ssp(BN(L0 edge)), bufB <- ssp(L3 .), bufA <- L5 .
  void finish() {
    Precision<T> &P = prec<T>(h);
    const Graph &g = h->g;
    const Dims d = h->d;
    if constexpr (sizeof(T) == 4) {
      if (narrow()) {  // readout MLP + edge tensors + per-frame mean in one launch
        Timer t(h, st(), K_READOUT_MLP);
        const double *ms = h->d_mean_std.as<double>();
        launch_readout_narrow(edge[cur], unit4, S, g, d, P.ro, ms, ms + 9, d_vec6, d_alpha, d_alpha_raw,
                              h->keep_stages ? bufA : nullptr, st());
        HIP_TRY(hipGetLastError());
        return;
      }
    }
    int pol_stride = 32;
    {
      Timer t(h, st(), K_READOUT_MLP);
      const int HP = std::max(d.FeP, 32);
      bool done = false;
      if constexpr (sizeof(T) == 4) {
        if (fused() && h->use_readout_fused) {  // the three layers in one launch; whole 64-byte rows of the 16 columns it writes
          pol_stride = h->keep_stages ? 32 : 16;  // (the stage snapshots read the unfused chain's 32-column layout)
          launch_readout_fused(edge[cur], ME, P.ro, bufA, h->mfma_f16, st(), pair_rows(), pol_stride);
          done = true;
        }
      }
      if (!done) {
        launch_rowgemm<T>(edge[cur], ME, d.FeP, P.ro.W0T, HP, bufA, P.ro.scale0, P.ro.shift0, true, 0,
                          nullptr, g, st());
        launch_rowgemm<T>(bufA, ME, HP, P.ro.W3T, HP, bufB, P.ones, P.ro.b3, true, 0, nullptr, g, st());
        launch_rowgemm<T>(bufB, ME, HP, P.ro.W5T, 32, bufA, nullptr, P.ro.b5, false, 0, nullptr, g,
                          st());
      }
    }
    {
      Timer t(h, st(), K_READOUT_REDUCE);
      const double *ms = h->d_mean_std.as<double>();
      launch_readout_reduce<T>(bufA, unit4, S, g, ms, ms + 9, d_vec6, d_alpha, d_alpha_raw, st(), pol_stride);
    }
    HIP_TRY(hipGetLastError());
  }
  // (the fused kernels also serve taped float32 runs: the EdgeBlock kernel records the one extra
  //  array the reverse pass needs; RN_POTGNN_TAPE_FUSED=0 keeps those runs on the unfused kernels)
  bool fused() const { return sizeof(T) == 4 && h->use_fused && (!prec<T>(h).tape_on || h->tape_fused); }
  bool narrow() const { return sizeof(T) == 4 && h->use_narrow && !prec<T>(h).tape_on; }
  // the fused NodeBlock on the centred copy of c1_linear (evaluation runs, split-f16 products)
  bool node_centred() const {
    static const bool on = !(getenv("RN_POTGNN_NODE_CENTRED") && atoi(getenv("RN_POTGNN_NODE_CENTRED")) == 0);
    return sizeof(T) == 4 && on && fused() && h->use_node_fused && h->mfma_f16 && (!prec<T>(h).tape_on || h->tape_ps);
  }
  // the role-specialised EdgeBlock (kernels_edge_ps.hip): split-f16 products and the folded gate scale only.  Taped runs
  // take it too (round 4: it records the pre-LayerNorm sums like the per-frame kernel, and the reverse pass recomputes
  // every projection from the taped node / edge rows, so it never sees the centred copies); RN_POTGNN_TAPE_PS=0 keeps
  // them on the per-frame kernel and the row-ordered NodeBlock
  bool role_split(const PassW<T> &w) const {
    // (round 5: also with exact-f32 products -- RN_POTGNN_MFMA=f32, the range guard's fallback -- on the kernel's F16 = false form)
    return sizeof(T) == 4 && fused() && h->use_ps && (w.c3_fast & 1) != 0 && (!prec<T>(h).tape_on || h->tape_ps);
  }
  // Edge rows as split-f16 pairs (kernels.hpp: launch_geom_rbf_pairs): when EVERY kernel that touches them in this run is one
  // that speaks the format -- the role-specialised EdgeBlock in every pass, the atom-owning NodeBlock, the fused readout --
  // and nothing else looks at them (no tape, no stage snapshots).  RN_POTGNN_PAIR_ROWS=0 keeps plain float32 rows.
  bool pair_rows() const {
    if (sizeof(T) != 4 || !h->want_pair_rows || h->keep_stages || prec<T>(h).tape_on || !node_centred() || h->g.na_num <= 0 || !h->use_readout_fused) return false;
    for (const auto &w : prec<T>(h).pass)
      if (!role_split(w)) return false;
    return true;
  }
  T *tape_agg(int p) {  // where the EdgeBlock's pre-LayerNorm sums are recorded (taped runs only)
    Precision<T> &P = prec<T>(h);
    return P.tape_on ? P.tape_agg[p].template as<T>() : nullptr;
  }
};

template <typename T>
void run_chunk(rn_potgnn *h, Lane<T> &ln, const double *d_pos, int S, double *d_alpha,
               float *d_vec6, double *d_alpha_raw, const T *d_lat, const int *d_types, const float *d_pos32 = nullptr) {
  ChunkRun<T> c(h, ln, d_pos, S, d_alpha, d_vec6, d_alpha_raw);
  c.d_lat = d_lat;
  c.d_types = d_types;
  c.d_pos32 = d_pos32;
  c.begin();
  for (int p = 0; p < h->cfg.num_message_passes; ++p) {
    c.stage_project(p);
    c.stage_aggregate(p);
  }
  c.finish();
}

// Two chunks A, B on the two lanes with their stages forced to alternate:
//   lane A:  G_A(0)  V_A(0)  G_A(1)  V_A(1) ...
//   lane B:          G_B(0)  V_B(0)  G_B(1) ...
// G_B(p) waits for G_A(p) and G_A(p+1) waits for G_B(p), so the two G stages never run
// together while each V stage (VALU-bound triplet loop) has the other chunk's G stage
// (matrix pipe + HBM streams) as its only companion on the chip.
template <typename T>
void run_pair(rn_potgnn *h, ChunkRun<T> &a, ChunkRun<T> &b) {
  const int P = h->cfg.num_message_passes;
  a.begin();
  b.begin();
  for (int p = 0; p < P; ++p) {
    if (p > 0) HIP_TRY(hipStreamWaitEvent(a.st(), h->ev_g[1], 0));  // after G_B(p-1)
    a.stage_project(p);
    HIP_TRY(hipEventRecord(h->ev_g[0], a.st()));
    a.stage_aggregate(p);
    HIP_TRY(hipStreamWaitEvent(b.st(), h->ev_g[0], 0));             // after G_A(p)
    b.stage_project(p);
    HIP_TRY(hipEventRecord(h->ev_g[1], b.st()));
    b.stage_aggregate(p);
  }
  a.finish();
  b.finish();
}

// The role-specialised EdgeBlock bounds its spin waits: a wait that runs out sets a word in HBM (and poisons the rows the
// workgroup stores from then on with NaN) instead of hanging the GPU.  Every entry point that synchronises with the device
// anyway looks at the word here, right after its synchronisation -- evaluation, the pipelined host entry's rn_potgnn_wait,
// the taped forward / backward of a training step, the Jacobian, the Adam step -- and clears it, so a timeout is reported by
// the call it happened in (or the first synchronising call behind it), never by an unrelated later one.  Entry points that do
// not synchronise (device buffers with synchronize = 0) hand back NaN rows in that case.
void check_ps_fail(rn_potgnn *h) {
  if (!h->use_ps || !h->ps_fail.p) return;
  int fail = 0;
  HIP_TRY(hipMemcpy(&fail, h->ps_fail.p, sizeof(int), hipMemcpyDeviceToHost));
#ifdef RN_PS_TIMING
  {
    long long t[40];
    HIP_TRY(hipMemcpy(t, (const char *)h->ps_fail.p + 64, sizeof(t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset((char *)h->ps_fail.p + 64, 0, sizeof(t)));
    static const char *names[20] = {"P sched+dma_wait", "P loads+split", "P split sync", "P seeds", "P request", "P guard",
                                    "P P'+c2", "P Q' (product, loads landed + finish)", "P norm sync+publish", "P ring guard", "C ready wait", "C prologue", "C loop",
                                    "C epilogue+done", "", "", "C(set B) ready wait", "C(set B) prologue", "C(set B) loop", "C(set B) epilogue+done"};
    for (int i = 0; i < 20; ++i)
      if (t[2 * i + 1]) fprintf(stderr, "[ps timing] %-22s %10.1f cycles x %lld\n", names[i], (double)t[2 * i] / (double)t[2 * i + 1], t[2 * i + 1]);
  }
#endif
#ifdef RN_PS_GRAM_DEBUG
  {
    int cnt = 0;
    HIP_TRY(hipMemcpy(&cnt, (const char *)h->ps_fail.p + 127 * 4, sizeof(int), hipMemcpyDeviceToHost));
    float dbg[144];
    HIP_TRY(hipMemcpy(dbg, (const char *)h->ps_fail.p + 128 * 4, sizeof(dbg), hipMemcpyDeviceToHost));
    fprintf(stderr, "[gram debug] %d mismatches\n", cnt);
    for (int k = 0; k < std::min(cnt, 12); ++k)
      fprintf(stderr, "  ref %g tab %g parts %g %g %g %g  d %g er %g tl %g urow %g r %g gr %g\n", dbg[12 * k], dbg[12 * k + 1], dbg[12 * k + 2],
              dbg[12 * k + 3], dbg[12 * k + 4], dbg[12 * k + 5], dbg[12 * k + 6], dbg[12 * k + 7], dbg[12 * k + 8], dbg[12 * k + 9],
              dbg[12 * k + 10], dbg[12 * k + 11]);
    HIP_TRY(hipMemset((char *)h->ps_fail.p + 127 * 4, 0, 4 + sizeof(dbg)));
  }
#endif
  if (fail == 0) return;
  HIP_TRY(hipMemset(h->ps_fail.p, 0, sizeof(int)));
  // (which wait: 1 = the producers' split sync, 2 = a producer waiting for round g - 2's P' / c2 rows to be taken,
  //  3 = a producer waiting for round g - back - 1 to be finished, 4 = a consumer waiting for its round,
  //  6 = a consumer waiting for the earlier arrivals of its round parity before it adds its own to c_rd)
  static const char *const which[7] = {
      "role-specialised EdgeBlock: a wait between producer and consumer waves timed out",
      "role-specialised EdgeBlock: the producers' split sync timed out",
      "role-specialised EdgeBlock: a producer's wait for the consumers to take the rows of an earlier round timed out",
      "role-specialised EdgeBlock: a producer's wait for the consumers to finish an earlier round (ring rows) timed out",
      "role-specialised EdgeBlock: a consumer's wait for its round timed out",
      "role-specialised EdgeBlock: a wait between producer and consumer waves timed out",
      "role-specialised EdgeBlock: a consumer's round-ordered wait before releasing its ring rows (c_rd) timed out"};
  throw HipError{hipErrorLaunchFailure, which[fail >= 0 && fail <= 6 ? fail : 0]};
}

template <typename T>
void forward_device(rn_potgnn *h, const double *d_pos, int64_t S, double *d_alpha, float *d_vec6,
                    double *d_alpha_raw, hipStream_t user, bool sync, const T *d_lat = nullptr,
                    const int *d_types = nullptr, const float *d_pos32 = nullptr /* instead of d_pos: float32 positions */) {
  ensure_precision<T>(h);
  Precision<T> &P = prec<T>(h);
  const int N = h->cfg.num_atoms;
  HIP_TRY(hipEventRecord(h->ev_start, user));
  const int lanes = h->num_lanes;
  for (int l = 0; l < lanes; ++l) HIP_TRY(hipStreamWaitEvent(P.lanes[l].stream, h->ev_start, 0));
  auto make = [&](int lane, int64_t first, int s) {
    ChunkRun<T> c(h, P.lanes[lane], d_pos ? d_pos + first * N * 3 : nullptr, s,
                  d_alpha ? d_alpha + first * 9 : nullptr, d_vec6 ? d_vec6 + first * 6 : nullptr,
                  d_alpha_raw ? d_alpha_raw + first * 9 : nullptr);
    c.d_pos32 = d_pos32 ? d_pos32 + first * N * 3 : nullptr;
    c.d_lat = d_lat ? d_lat + first * 9 : nullptr;
    c.d_types = d_types ? d_types + first * N : nullptr;
    return c;
  };
  int64_t done = 0;
  // Work chunks of EQUAL size: as many as the workspace demands, the frames split evenly among them (10 000 frames at a
  // capacity of 2344: five launches of 2000 instead of four of 2344 and one of 624, whose persistent workgroups idle through
  // a tail as long as a full launch's).  Results do not depend on the chunking (bit-identical: tests).
  const int capacity = chunk_frames<T>(h);
  const int64_t pieces = (S + capacity - 1) / capacity;
  const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(capacity, (S + pieces - 1) / std::max<int64_t>(pieces, 1)));
  h->train_S = 0;  // lane 0's workspace is about to be overwritten: a pending train_forward is void
  while (done < S) {
    const int64_t left = S - done;
    if (lanes == 2 && h->interleave && !h->keep_stages && left > 1) {
      // split what is left of this round evenly over the two lanes
      const int64_t both = std::min<int64_t>(left, 2 * (int64_t)chunk);
      const int sa = (int)((both + 1) / 2), sb = (int)(both - sa);
      ChunkRun<T> a = make(0, done, sa), b = make(1, done + sa, sb);
      run_pair<T>(h, a, b);
      h->last_chunk_structs = sa;
      done += both;
    } else {
      const int s = (int)std::min<int64_t>(chunk, left);
      run_chunk<T>(h, P.lanes[0], d_pos ? d_pos + done * N * 3 : nullptr, s, d_alpha ? d_alpha + done * 9 : nullptr,
                   d_vec6 ? d_vec6 + done * 6 : nullptr, d_alpha_raw ? d_alpha_raw + done * 9 : nullptr,
                   d_lat ? d_lat + done * 9 : nullptr, d_types ? d_types + done * N : nullptr,
                   d_pos32 ? d_pos32 + done * N * 3 : nullptr);
      h->last_chunk_structs = s;
      done += s;
    }
  }
  h->last_was_f64 = sizeof(T) == 8;
  for (int l = 0; l < lanes; ++l) {
    HIP_TRY(hipEventRecord(P.lanes[l].done, P.lanes[l].stream));
    HIP_TRY(hipStreamWaitEvent(user, P.lanes[l].done, 0));
  }
  if (sync) {
    HIP_TRY(hipStreamSynchronize(user));
    resolve_timers(h);
    check_ps_fail(h);
  }
}

// Reverse pass over a taped forward of S frames with B cotangents per frame.
//   d_dout6   device [S*B, 6]  cotangents of the standardised 6-vectors
//   d_dpos    device f64 [S*B, N, 3] or null   -> d / d(fractional positions)
//   grad      device packed-layout gradient blob or null -> parameter gradients (training)
//   train_bn  readout BatchNorm used batch statistics (z1 / bn_stats hold the taped values)
template <typename T>
struct Reverse {
  int S, B;
  const T *d_dout6;
  double *d_dpos;
  T *grad;
  bool train_bn;
};

enum { DE0, DE1, DN0, DN1, DNX, DPQ, DNP3, DC2, DPROD, DBC1, DNPC1, DPOL, DUNIT, DH, POL, DOUT, DPROD2, BWN };

template <typename T>
void reverse_pass(rn_potgnn *h, ChunkRun<T> &c, const Reverse<T> &rv) {
  Precision<T> &P = prec<T>(h);
  const PackedLayout &L = h->lay;
  const Graph &g = h->g;
  const Dims d = h->d;
  const int N = g.N, E = g.E, NP = h->cfg.num_message_passes;
  const int S = rv.S, B = rv.B, C = S * B;
  const int HP = std::max(d.FeP, 32);
  hipStream_t st = c.st();
  const size_t ce = (size_t)C * E, cn = (size_t)C * N, fe = (size_t)S * E, fn = (size_t)S * N;
  const size_t sizes[BWN] = {ce * d.FeP, ce * d.FeP, cn * d.FnP, cn * d.FnP, cn * d.FnP,
                             ce * 4 * d.FeP, cn * 6 * d.FeP, ce * std::max(2 * d.FeP, HP), ce * d.FnP,
                             ce * 2 * d.FnP, cn * 2 * d.FnP, ce * 32, ce * 4, ce * HP, fe * 32,
                             (size_t)C * 6, rv.grad ? fe * d.FnP : 0};
  T *b[BWN];
  for (int i = 0; i < BWN; ++i) {
    P.bw[i].ensure(sizes[i] * sizeof(T));
    b[i] = P.bw[i].template as<T>();
  }
  T *bufA = c.bufA, *bufB = c.bufB, *unit4 = c.unit4;
  T *G = rv.grad;
  T *Wd = P.weights.template as<T>();
  // dX[R, K] (+)= dY[R, N] * W^T with W = the forward's [K][N] matrix at `w_off`, its
  // transposed copy at `t_off`: MFMA projection kernel in float32, simple kernel otherwise
  // (float32 with the split-f16 products enabled: rowgemm_split_kernel where the shape allows)
  const bool split_gemm = sizeof(T) == 4 && h->mfma_f16;
  auto back_gemm = [&](const T *dY, int64_t R, int N, size_t w_off, size_t t_off, int K, T *dX,
                       bool accumulate) {
    if constexpr (sizeof(T) == 4) {
      if (split_gemm && launch_rowgemm_split(dY, N, N, R, Wd + t_off, K, dX, accumulate, nullptr, 0, nullptr, g, st)) return;
      if (launch_rowgemm_blocks(dY, N, N, R, Wd + t_off, K, dX, accumulate, g, st)) return;
    }
    launch_gemm_nt<T>(dY, R, N, Wd + w_off, N, K, dX, accumulate, st);
  };
  // Y = X W (+ bias): a recomputed forward projection of the pass
  auto project = [&](const T *X, int64_t R, int K, const T *WT, int NOUT, T *Y, const T *bias, int amode, const T *node) {
    if constexpr (sizeof(T) == 4) {
      if (split_gemm && launch_rowgemm_split(X, K, K, R, WT, NOUT, Y, false, bias, amode, node, g, st)) return;
    }
    launch_rowgemm<T>(X, R, K, WT, NOUT, Y, nullptr, bias, false, amode, node, g, st);
  };

  // weight-gradient products (float32): per-workgroup partial sums into an arena, ONE reduction at the end
  TnDeferred tn_store, *tn = nullptr;
  if constexpr (sizeof(T) == 4) {
    static const bool tn_atomic = getenv("RN_POTGNN_TN_ATOMIC") && atoi(getenv("RN_POTGNN_TN_ATOMIC")) != 0;
    if (G && !tn_atomic) {
      size_t need = tn_partial_elems(ce, HP, 32) + tn_partial_elems(ce, HP, HP) + tn_partial_elems(ce, d.FeP, HP);
      const size_t per_pass = tn_partial_elems(ce, d.FeP, 4 * d.FeP) + tn_partial_elems(cn, d.FnP, 6 * d.FeP) +
                              tn_partial_elems(ce, d.FnP, 2 * d.FeP) + tn_partial_elems(ce, d.FeP, 2 * d.FnP) +
                              tn_partial_elems(cn, d.FnP, 2 * d.FnP);
      need += per_pass * (size_t)NP;
      P.tn_arena.ensure(need * sizeof(float));
      tn_store.arena = P.tn_arena.template as<float>();
      tn_store.capacity = need;
      tn = &tn_store;
    }
  }

  // The weight-gradient products dW^T = X^T dY of a pass hang off the cotangent chain: nothing reads them before the final
  // reduction.  They run on a SIDE stream, ordered behind the kernel that produced their dY and joined before the next pass
  // overwrites it (and before the reduction) -- small latency-bound launches under the chain's small latency-bound launches.
  // (RN_POTGNN_TN_SIDE=0: in line on the chain's stream, as through round 5.)
  static const bool want_side = !(getenv("RN_POTGNN_TN_SIDE") && atoi(getenv("RN_POTGNN_TN_SIDE")) == 0);
  const bool overlap = G != nullptr && sizeof(T) == 4 && want_side;
  hipStream_t sd = st;
  if (overlap) {
    if (!P.side) {
      HIP_TRY(hipStreamCreateWithFlags(&P.side, hipStreamNonBlocking));
      for (auto &e : P.ev_side) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    sd = P.side;
  }
  auto side_after_chain = [&](int which) {  // what the chain has enqueued so far precedes what the side stream gets next
    if (!overlap) return;
    HIP_TRY(hipEventRecord(P.ev_side[which], st));
    HIP_TRY(hipStreamWaitEvent(sd, P.ev_side[which], 0));
  };
  auto chain_after_side = [&]() {  // ... and the other way round
    if (!overlap) return;
    HIP_TRY(hipEventRecord(P.ev_side[2], sd));
    HIP_TRY(hipStreamWaitEvent(st, P.ev_side[2], 0));
  };

  // ---- readout: recompute h1 (bufA), h2 (bufB), pol; then reverse
  const T *edgeP = P.tape_edge[NP].template as<T>();
  if (rv.train_bn) {
    // (the column sums of train_forward are still in bn_stats, already reduced over ranks)
    launch_bn_train_apply<T>(P.tape_z1.template as<T>(), fe, HP, d.Fe, P.bn_stats.template as<double>(),
                             h->bn_count, Wd + L.bn_w, Wd + L.bn_b, bufA, b[DH] /*scratch for mean/var*/, st);
  } else {
    launch_rowgemm<T>(edgeP, fe, d.FeP, P.ro.W0T, HP, bufA, P.ro.scale0, P.ro.shift0, true, 0, nullptr, g, st);
  }
  launch_rowgemm<T>(bufA, fe, HP, P.ro.W3T, HP, bufB, P.ones, P.ro.b3, true, 0, nullptr, g, st);
  launch_rowgemm<T>(bufB, fe, HP, P.ro.W5T, 32, b[POL], nullptr, P.ro.b5, false, 0, nullptr, g, st);
  launch_readout_bwd<T>(rv.d_dout6, b[POL], unit4, C, B, g, b[DPOL], b[DUNIT], st);
  if (G) launch_gemm_tn<T>(bufB, HP, b[DPOL], 32, ce, HP, 32, G + L.W5T, 32, G + L.b5, 0, nullptr, g, st, tn);
  back_gemm(b[DPOL], ce, 32, L.W5T, L.t_W5, HP, b[DH], false);                     // d h2
  launch_ssp_bwd<T>(b[DH], bufB, nullptr, E, HP, C, B, st);                        // d z2
  if (G) launch_gemm_tn<T>(bufA, HP, b[DH], HP, ce, HP, HP, G + L.W3T, HP, G + L.b3, 0, nullptr, g, st, tn);
  back_gemm(b[DH], ce, HP, L.W3T, L.t_W3, HP, b[DC2], false);                      // d h1
  if (rv.train_bn) {
    launch_ssp_bwd<T>(b[DC2], bufA, nullptr, E, HP, C, B, st);                     // d (BN output)
    double *stats = P.bn_stats.template as<double>(), *sums = stats + 2 * HP, *own = stats + 4 * HP;
    launch_bn_bwd_sums<T>(b[DC2], P.tape_z1.template as<T>(), fe, HP, d.Fe, stats, h->bn_count, sums, st);
    HIP_TRY(hipMemcpyAsync(own, sums, sizeof(double) * 2 * HP, hipMemcpyDeviceToDevice, st));
    reduce_over_ranks(h, sums, (size_t)2 * HP, st);
    launch_bn_bwd_apply<T>(b[DC2], P.tape_z1.template as<T>(), fe, HP, d.Fe, stats, h->bn_count, sums, own,
                           Wd + L.bn_w, G + L.bn_w, G + L.bn_b, st);               // d z1
    launch_gemm_tn<T>(edgeP, d.FeP, b[DC2], HP, ce, d.FeP, HP, G + L.W0T, HP, G + L.b0p, 0, nullptr, g, st, tn);
  } else {
    launch_ssp_bwd<T>(b[DC2], bufA, P.ro.scale0, E, HP, C, B, st);                 // d acc1
  }
  back_gemm(b[DC2], ce, HP, L.W0T, L.t_W0, d.FeP, b[DE0], false);                  // d edge_P
  HIP_TRY(hipMemsetAsync(b[DN0], 0, cn * d.FnP * sizeof(T), st));                  // d node_P = 0

  int cur = 0;  // b[DE0 + cur], b[DN0 + cur] hold the cotangents of (edge, node)_{p+1}
  for (int p = NP - 1; p >= 0; --p) {
    const PassW<T> &w = P.pass[p];
    PassW<T> gw = w;  // the same offsets inside the gradient blob
    if (G) {
      const auto &q = L.pass[p];
      gw.c1_norm = {G + q.c1n_g, G + q.c1n_b};
      gw.final_norm = {G + q.fin_g, G + q.fin_b};
      gw.c2_norm_1 = {G + q.c2n1_g, G + q.c2n1_b};
      gw.c2_norm_2 = {G + q.c2n2_g, G + q.c2n2_b};
      gw.c3_norm_1 = {G + q.c3n1_g, G + q.c3n1_b};
      gw.c3_norm_2 = {G + q.c3n2_g, G + q.c3n2_b};
    }
    const T *node0 = P.tape_node[p].template as<T>(), *node1 = P.tape_node[p + 1].template as<T>();
    const T *edge0 = P.tape_edge[p].template as<T>(), *edge1 = P.tape_edge[p + 1].template as<T>();
    T *de_next = b[DE0 + cur], *de_prev = b[DE0 + (cur ^ 1)];
    T *dn_next = b[DN0 + cur], *dn_prev = b[DN0 + (cur ^ 1)];
    // recompute this pass's projections from the tape
    project(node0, fn, d.FnP, w.c1_WnT, 2 * d.FnP, c.npc1, w.c1_bias, 0, nullptr);
    project(node1, fn, d.FnP, w.c3_WnT, 6 * d.FeP, c.np3, w.c3_nshift, 0, nullptr);
    project(edge0, fe, d.FeP, w.c3_WeT, 4 * d.FeP, bufB, nullptr, 0, nullptr);
    project(nullptr, fe, d.FnP, w.c2_WT, 2 * d.FeP, bufA, w.c2_bias, 1, node1);
    // EdgeBlock
    chain_after_side();  // (the previous pass's products have read the dY buffers this pass overwrites)
    launch_edge_bwd<T>(bufB, c.np3, bufA, edge1, P.tape_agg[p].template as<T>(), de_next, de_prev, b[DPQ],
                       b[DNP3], b[DC2], C, B, g, d, w, G ? &gw : nullptr, st);
    side_after_chain(0);
    back_gemm(b[DPQ], ce, 4 * d.FeP, L.pass[p].c3_WeT, L.pass[p].t_c3We, d.FeP, de_prev, true);
    // node_{p+1} cotangent: incoming + projections + c2 operand
    // (accumulated in place: dn_next is dead once this pass's NodeBlock has consumed it)
    back_gemm(b[DNP3], cn, 6 * d.FeP, L.pass[p].c3_WnT, L.pass[p].t_c3Wn, d.FnP, dn_next, true);
    back_gemm(b[DC2], ce, 2 * d.FeP, L.pass[p].c2_WT, L.pass[p].t_c2W, d.FnP, b[DPROD], false);
    launch_prod_bwd<T>(b[DPROD], node1, dn_next, C, B, g, d, st);
    if (G) {
      const auto &q = L.pass[p];
      launch_gemm_tn<T>(edge0, d.FeP, b[DPQ], 4 * d.FeP, ce, d.FeP, 4 * d.FeP, G + q.c3_WeT, 4 * d.FeP, nullptr, 0, nullptr, g, sd, tn);
      launch_gemm_tn<T>(node1, d.FnP, b[DNP3], 6 * d.FeP, cn, d.FnP, 6 * d.FeP, G + q.c3_WnT, 6 * d.FeP, G + q.c3_nshift, 0, nullptr, g, sd, tn);
      // the c2 operand node[b]*node[a], read as plain rows: into d prod's buffer once that is consumed -- or, beside the chain,
      // into a buffer of its own
      T *prod = overlap ? b[DPROD2] : b[DPROD];
      launch_prod_fwd<T>(node1, prod, fe, g, d, sd);
      launch_gemm_tn<T>(prod, d.FnP, b[DC2], 2 * d.FeP, ce, d.FnP, 2 * d.FeP, G + q.c2_WT, 2 * d.FeP, G + q.c2_bias, 0, nullptr, g, sd, tn);
    }
    // NodeBlock (needs bc1 = We edge_p, recomputed into bufA now that c2pre is consumed)
    project(edge0, fe, d.FeP, w.c1_WeT, 2 * d.FnP, bufA, nullptr, 0, nullptr);
    launch_node_bwd<T>(c.npc1, bufA, node1, dn_next, dn_prev, b[DBC1], b[DNPC1], C, B, g, d, w,
                       G ? &gw : nullptr, st);
    side_after_chain(1);
    back_gemm(b[DBC1], ce, 2 * d.FnP, L.pass[p].c1_WeT, L.pass[p].t_c1We, d.FeP, de_prev, true);
    back_gemm(b[DNPC1], cn, 2 * d.FnP, L.pass[p].c1_WnT, L.pass[p].t_c1Wn, d.FnP, dn_prev, true);
    if (G) {
      const auto &q = L.pass[p];
      launch_gemm_tn<T>(edge0, d.FeP, b[DBC1], 2 * d.FnP, ce, d.FeP, 2 * d.FnP, G + q.c1_WeT, 2 * d.FnP, nullptr, 0, nullptr, g, sd, tn);
      launch_gemm_tn<T>(node0, d.FnP, b[DNPC1], 2 * d.FnP, cn, d.FnP, 2 * d.FnP, G + q.c1_WnT, 2 * d.FnP, G + q.c1_bias, 0, nullptr, g, sd, tn);
    }
    cur ^= 1;
  }
  chain_after_side();
  if (tn) launch_tn_reduce(*tn, st);
  if (G) {
    P.type_sums.ensure((size_t)h->cfg.num_atom_types * d.Fn * sizeof(T));
    launch_node_embed_bwd<T>(b[DN0 + cur], S, g, d, h->cfg.num_atom_types, Wd + L.emb, Wd + L.W2, Wd + L.b2,
                             Wd + L.W4, G + L.emb, G + L.W2, G + L.b2, G + L.W4, G + L.b4,
                             P.type_sums.template as<T>(), c.d_types, st);
  }
  if (rv.d_dpos)
    launch_geom_bwd<T>(b[DE0 + cur], b[DUNIT], unit4, P.lattice.template as<T>(), P.offsets,
                       (T)h->cfg.gauss_coefficient, C, B, g, d, rv.d_dpos, st);
  HIP_TRY(hipGetLastError());
}

template <typename T>
void ensure_tape(rn_potgnn *h, int S) {
  Precision<T> &P = prec<T>(h);
  const int NP = h->cfg.num_message_passes;
  P.tape_node.resize(NP + 1);
  P.tape_edge.resize(NP + 1);
  P.tape_agg.resize(NP + 1);
  for (int p = 0; p <= NP; ++p) {
    P.tape_node[p].ensure((size_t)S * h->g.N * h->d.FnP * sizeof(T));
    P.tape_edge[p].ensure((size_t)S * h->g.E * h->d.FeP * sizeof(T));
    P.tape_agg[p].ensure((size_t)S * h->g.E * h->d.FeP * sizeof(T));
  }
  // the reverse pass and the training-mode readout work on full-width projections of the taped batch
  Lane<T> &ln = P.lanes[0];
  ln.bufA.ensure((size_t)S * h->g.E * bufA_width(h, false) * sizeof(T));
  ln.bufB.ensure((size_t)S * h->g.E * 4 * h->d.FeP * sizeof(T));
}

// forward of S frames on lane 0 with the per-pass embeddings recorded
template <typename T>
ChunkRun<T> taped_forward(rn_potgnn *h, const double *d_pos, int S, const T *d_lat = nullptr, const int *d_types = nullptr) {
  Precision<T> &P = prec<T>(h);
  ensure_tape<T>(h, S);
  P.tape_on = true;
  ChunkRun<T> c(h, P.lanes[0], d_pos, S, nullptr, nullptr, nullptr);
  c.d_lat = d_lat;      // [S][9] per-sample lattices or null (the reference structure's)
  c.d_types = d_types;  // [S][N] per-sample atom types or null
  try {
    c.begin();
    for (int p = 0; p < h->cfg.num_message_passes; ++p) {
      c.stage_project(p);
      c.stage_aggregate(p);
    }
  } catch (...) {
    P.tape_on = false;
    throw;
  }
  P.tape_on = false;
  return c;
}

// d(standardised 6-vector)/d(fractional positions) at ONE structure: six cotangents (one
// per component) pushed back through readout, P x (EdgeBlock, NodeBlock), radial basis and
// geometry.
template <typename T>
void jacobian(rn_potgnn *h, const double *host_pos, double *host_jac /*[6][N*3]*/) {
  ensure_precision<T>(h);
  Precision<T> &P = prec<T>(h);
  const int N = h->g.N;
  h->train_S = 0;  // the tape and lane 0 are reused: a pending train_forward is void
  h->io_pos.ensure((size_t)N * 3 * sizeof(double));
  HIP_TRY(hipMemcpy(h->io_pos.p, host_pos, (size_t)N * 3 * sizeof(double), hipMemcpyHostToDevice));
  ChunkRun<T> c = taped_forward<T>(h, h->io_pos.as<double>(), 1);
  hipStream_t st = c.st();
  DeviceBuf dposbuf, seeds;
  dposbuf.ensure((size_t)6 * N * 3 * sizeof(double));
  seeds.ensure(36 * sizeof(T));
  HIP_TRY(hipMemsetAsync(dposbuf.p, 0, (size_t)6 * N * 3 * sizeof(double), st));
  std::vector<T> eye(36, (T)0);
  for (int k = 0; k < 6; ++k) eye[k * 6 + k] = (T)1;
  HIP_TRY(hipMemcpyAsync(seeds.p, eye.data(), sizeof(T) * 36, hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  Reverse<T> rv{1, 6, seeds.as<T>(), dposbuf.as<double>(), nullptr, false};
  reverse_pass<T>(h, c, rv);
  HIP_TRY(hipStreamSynchronize(st));
  check_ps_fail(h);
  HIP_TRY(hipMemcpy(host_jac, dposbuf.p, (size_t)6 * N * 3 * sizeof(double), hipMemcpyDeviceToHost));
  (void)P;
}

// ---- device-resident optimisation step: which entries of the packed blob are parameters, which are
// functions of parameters (and how to recompute them on the device)

std::vector<DerivedOp> derived_ops(const rn_potgnn *h, int *first_stage = nullptr) {
  const PackedLayout &L = h->lay;
  const int FnP = h->d.FnP, FeP = h->d.FeP, Fe = h->d.Fe, HP = std::max(FeP, 32);
  std::vector<DerivedOp> ops;
  auto transpose = [&](size_t src, int K, int N, size_t dst) { ops.push_back({0, K, N, 1.0f, src, dst}); };
  auto scaled = [&](size_t src, int n, float sc, size_t dst) { ops.push_back({1, n, 1, sc, src, dst}); };
  for (const auto &q : L.pass) {
    transpose(q.c3_WeT, FeP, 4 * FeP, q.t_c3We);
    transpose(q.c3_WnT, FnP, 6 * FeP, q.t_c3Wn);
    transpose(q.c2_WT, FnP, 2 * FeP, q.t_c2W);
    transpose(q.c1_WeT, FeP, 2 * FnP, q.t_c1We);
    transpose(q.c1_WnT, FnP, 2 * FnP, q.t_c1Wn);
    scaled(q.c3n1_g, FeP, -1.4426950408889634f, q.c3n1_gs);
    scaled(q.c3n1_b, FeP, -1.4426950408889634f, q.c3n1_bs);
    scaled(q.c3n1_g + FeP, FeP, 2.0f * 1.4426950408889634f, q.c3n1_gs + FeP);
    scaled(q.c3n1_b + FeP, FeP, 2.0f * 1.4426950408889634f, q.c3n1_bs + FeP);
    scaled(q.c2n1_g, FeP, -1.4426950408889634f, q.c2n1_gs);
    scaled(q.c2n1_b, FeP, -1.4426950408889634f, q.c2n1_bs);
    scaled(q.c2n1_g + FeP, FeP, 2.0f * 1.4426950408889634f, q.c2n1_gs + FeP);
    scaled(q.c2n1_b + FeP, FeP, 2.0f * 1.4426950408889634f, q.c2n1_bs + FeP);
    scaled(q.c1n_g, FnP, -1.4426950408889634f, q.c1n_gs);
    scaled(q.c1n_b, FnP, -1.4426950408889634f, q.c1n_bs);
    scaled(q.c1n_g + FnP, FnP, 2.0f * 1.4426950408889634f, q.c1n_gs + FnP);
    scaled(q.c1n_b + FnP, FnP, 2.0f * 1.4426950408889634f, q.c1n_bs + FnP);
  }
  transpose(L.W0T, FeP, HP, L.t_W0);
  transpose(L.W3T, HP, HP, L.t_W3);
  transpose(L.W5T, HP, 32, L.t_W5);
  scaled(L.b0p, Fe, 1.0f, L.b0);  // the bias of readout Linear 0 lives twice (eval fold / training forward)
  for (const MfmaScaleOp &m : mfma_scale_ops(h)) ops.push_back({2, m.K, m.N, (float)m.ld, m.src, m.dst});
  for (const CentreOp &c : centre_ops(h)) ops.push_back({3, c.K, c.N, 1.0f, c.src, c.dst, c.F, c.FP});
  if (first_stage) *first_stage = (int)ops.size();
  // second launch: the prescales of the centred copies read what the first launch wrote
  for (const MfmaScaleOp &m : centred_scale_ops(h)) ops.push_back({2, m.K, m.N, (float)m.ld, m.src, m.dst});
  return ops;
}

// 1 where a packed entry is a trainable parameter: pack a state dict of ones, then drop the buffers
// and every derived range
std::vector<unsigned char> trainable_mask(rn_potgnn *h) {
  std::vector<float> ones(rn_potgnn_weight_count(&h->cfg), 1.0f);
  const std::vector<float> keep = h->packed;
  pack_weights(h, ones.data());
  std::vector<unsigned char> mask(h->packed.size());
  for (size_t i = 0; i < mask.size(); ++i) mask[i] = h->packed[i] != 0.0f;
  h->packed = keep;
  const PackedLayout &L = h->lay;
  const int HP = std::max(h->d.FeP, 32);
  auto clear = [&](size_t o, size_t n) { std::fill(mask.begin() + o, mask.begin() + o + n, (unsigned char)0); };
  for (const DerivedOp &op : derived_ops(h))
    clear(op.dst, op.kind == 2 ? 2 : (size_t)op.K * ((op.kind == 0 || op.kind == 3) ? op.N : 1));
  clear(L.ones, HP);
  clear(L.offsets, h->d.FeP);  // buffers of the state dict: Gaussian offsets, BatchNorm running statistics
  clear(L.bn_rm, h->d.Fe);
  clear(L.bn_rv, h->d.Fe);
  return mask;
}

// refresh `packed` (and the float64 copy, when it exists) from the device weights
void sync_host(rn_potgnn *h) {
  if (!h->host_stale) return;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->packed.data(), h->f32.weights.p, h->packed.size() * sizeof(float), hipMemcpyDeviceToHost));
  h->host_stale = false;
  if (h->f64.ready) upload_weights<double>(h);
}

// ---- training: forward with batch-statistics BatchNorm, then parameter gradients (float32 for
// the product path; float64 for validating the reverse pass against float64 autograd)
// The taped forward + the training-mode readout (batch-statistics BatchNorm) over the S structures whose positions (and, per
// h->train_lat / h->train_types, lattices and atom types) sit in h->io_pos / io_lat / io_types; everything is enqueued on lane
// 0's stream, the standardised 6-vectors end up in h->io_vec6 / io_alpha and the batch statistics in P.mv.
template <typename T>
ChunkRun<T> train_forward_core(rn_potgnn *h, int S) {
  Precision<T> &P = prec<T>(h);
  const PackedLayout &L = h->lay;
  const Graph &g = h->g;
  const Dims d = h->d;
  const int HP = std::max(d.FeP, 32);
  ChunkRun<T> c = taped_forward<T>(h, h->io_pos.as<double>(), S, h->train_lat ? h->io_lat.as<T>() : nullptr,
                                   h->train_types ? h->io_types.as<int>() : nullptr);
  hipStream_t st = c.st();
  const int64_t R = (int64_t)S * g.E;
  T *Wd = P.weights.template as<T>();
  P.tape_z1.ensure((size_t)R * HP * sizeof(T));
  P.bn_stats.ensure(sizeof(double) * 6 * HP);  // forward sums (+ row count during the reduction) | backward sums | own copy
  DeviceBuf &mv = P.mv;
  mv.ensure(sizeof(T) * 2 * HP);
  const T *edgeP = P.tape_edge[h->cfg.num_message_passes].template as<T>();
  // z1 = edge W0^T + b0 ; h1 = ssp(BN_batch(z1)) ; then the rest of the readout as in eval
  launch_rowgemm<T>(edgeP, R, d.FeP, P.ro.W0T, HP, P.tape_z1.template as<T>(), nullptr, Wd + L.b0p, false, 0,
                    nullptr, g, st);
  {
    // batch statistics over every row of the batch -- of all ranks in a data-parallel run
    double *stats = P.bn_stats.template as<double>();
    launch_bn_col_sums<T>(P.tape_z1.template as<T>(), R, HP, stats, st);
    h->bn_count = (double)R;
    if (h->reducer) {
      const double rows = (double)R;  // slot 2HP is outside the range col_sums clears
      HIP_TRY(hipMemcpy(stats + 2 * HP, &rows, sizeof(double), hipMemcpyHostToDevice));
      reduce_over_ranks(h, stats, (size_t)2 * HP + 1, st);  // [sum z | sum z^2 | rows]
      HIP_TRY(hipMemcpy(&h->bn_count, stats + 2 * HP, sizeof(double), hipMemcpyDeviceToHost));
    }
    launch_bn_train_apply<T>(P.tape_z1.template as<T>(), R, HP, d.Fe, stats, h->bn_count, Wd + L.bn_w, Wd + L.bn_b,
                             c.bufA, mv.template as<T>(), st);
  }
  launch_rowgemm<T>(c.bufA, R, HP, P.ro.W3T, HP, c.bufB, P.ones, P.ro.b3, true, 0, nullptr, g, st);
  launch_rowgemm<T>(c.bufB, R, HP, P.ro.W5T, 32, c.bufA, nullptr, P.ro.b5, false, 0, nullptr, g, st);
  h->io_vec6.ensure((size_t)S * 6 * sizeof(float));
  h->io_alpha.ensure((size_t)S * 9 * sizeof(double));
  const double *ms = h->d_mean_std.as<double>();
  launch_readout_reduce<T>(c.bufA, c.unit4, S, g, ms, ms + 9, h->io_vec6.as<float>(), nullptr,
                           h->io_alpha.as<double>() /* standardised 3x3 in double */, st);
  HIP_TRY(hipGetLastError());
  return c;
}
// running statistics where the weights live (torch: momentum 0.1, unbiased variance) and everything derived from them
inline void train_running_stats(rn_potgnn *h, hipStream_t st) {
  Precision<float> &P = h->f32;
  const PackedLayout &L = h->lay;
  const Dims d = h->d;
  const int HP = std::max(d.FeP, 32);
  float *Wd = P.weights.as<float>();
  const double rows = h->bn_count;
  launch_bn_running(Wd + L.bn_rm, Wd + L.bn_rv, P.mv.as<float>(), P.mv.as<float>() + HP, d.Fe, 0.1,
                    rows / std::max(rows - 1.0, 1.0), st);
  launch_setup<float>(Wd + L.emb, Wd + L.W2, Wd + L.b2, Wd + L.W4, Wd + L.b4, h->cfg.num_atom_types, h->d,
                      Wd + L.node_table, Wd + L.b0, Wd + L.bn_w, Wd + L.bn_b, Wd + L.bn_rm, Wd + L.bn_rv,
                      Wd + L.scale0, Wd + L.shift0, st);
  h->host_stale = true;
}

template <typename T>
void train_forward(rn_potgnn *h, const double *host_pos, int S, T *vec6, T *batch_mean, T *batch_var,
                   const double *host_lat = nullptr /* [S][9] or null */, const int32_t *host_types = nullptr /* [S][N] or null */) {
  ensure_precision<T>(h);
  Precision<T> &P = prec<T>(h);
  const Graph &g = h->g;
  const Dims d = h->d;
  const int HP = std::max(d.FeP, 32);
  const size_t pb = (size_t)S * g.N * 3 * sizeof(double);
  h->io_pos.ensure(pb);
  HIP_TRY(hipMemcpy(h->io_pos.p, host_pos, pb, hipMemcpyHostToDevice));
  // per-sample lattices (in the arithmetic of the run, as the reference's forward casts them) and atom types: the graph
  // topology stays the reference structure's (_gnn.py:603-611, 541-557)
  if (host_lat) {
    std::vector<T> lat((size_t)S * 9);
    for (size_t i = 0; i < lat.size(); ++i) lat[i] = (T)host_lat[i];
    h->io_lat.ensure(lat.size() * sizeof(T));
    HIP_TRY(hipMemcpy(h->io_lat.p, lat.data(), lat.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  if (host_types) {
    h->io_types.ensure((size_t)S * g.N * sizeof(int32_t));
    HIP_TRY(hipMemcpy(h->io_types.p, host_types, (size_t)S * g.N * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  h->train_lat = host_lat != nullptr;
  h->train_types = host_types != nullptr;
  ChunkRun<T> c = train_forward_core<T>(h, S);
  hipStream_t st = c.st();
  DeviceBuf &mv = P.mv;
  HIP_TRY(hipStreamSynchronize(st));
  check_ps_fail(h);
  if constexpr (sizeof(T) == 4) {
    HIP_TRY(hipMemcpy(vec6, h->io_vec6.p, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToHost));
  } else {  // (xx,yy,zz,xy,xz,yz) of the standardised tensor, at full precision
    std::vector<double> raw((size_t)S * 9);
    HIP_TRY(hipMemcpy(raw.data(), h->io_alpha.p, raw.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int pick[6] = {0, 4, 8, 1, 2, 5};
    for (int s = 0; s < S; ++s)
      for (int k = 0; k < 6; ++k) vec6[(size_t)s * 6 + k] = (T)raw[(size_t)s * 9 + pick[k]];
  }
  if constexpr (sizeof(T) == 4) {
    if (h->device_training) {
      train_running_stats(h, st);
      HIP_TRY(hipStreamSynchronize(st));
    }
  }
  std::vector<T> mvh(2 * HP);
  HIP_TRY(hipMemcpy(mvh.data(), mv.p, sizeof(T) * 2 * HP, hipMemcpyDeviceToHost));
  for (int k = 0; k < d.Fe; ++k) {
    batch_mean[k] = mvh[k];
    batch_var[k] = mvh[HP + k];
  }
  h->train_S = S;
  h->train_prec = (int)sizeof(T);
}

// The same step with everything where it already is (device-resident training, float32): positions [S][N][3] float64,
// optional lattices [S][9] float32 and atom types [S][N] int32, and the [S][6] result are DEVICE buffers; work is ordered
// after `user` and `user` is made to wait for it; no host round trip, no synchronisation.
inline void train_forward_device(rn_potgnn *h, const double *d_pos, int S, const float *d_lat, const int32_t *d_types,
                                 float *d_vec6, hipStream_t user) {
  ensure_precision<float>(h);
  Precision<float> &P = h->f32;
  const Graph &g = h->g;
  hipStream_t st = P.lanes[0].stream;
  HIP_TRY(hipEventRecord(h->ev_start, user));
  HIP_TRY(hipStreamWaitEvent(st, h->ev_start, 0));
  const size_t pb = (size_t)S * g.N * 3 * sizeof(double);
  h->io_pos.ensure(pb);  // (the reverse pass reads the positions again)
  HIP_TRY(hipMemcpyAsync(h->io_pos.p, d_pos, pb, hipMemcpyDeviceToDevice, st));
  if (d_lat) {
    h->io_lat.ensure((size_t)S * 9 * sizeof(float));
    HIP_TRY(hipMemcpyAsync(h->io_lat.p, d_lat, (size_t)S * 9 * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  if (d_types) {
    h->io_types.ensure((size_t)S * g.N * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(h->io_types.p, d_types, (size_t)S * g.N * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  }
  h->train_lat = d_lat != nullptr;
  h->train_types = d_types != nullptr;
  (void)train_forward_core<float>(h, S);
  HIP_TRY(hipMemcpyAsync(d_vec6, h->io_vec6.p, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToDevice, st));
  train_running_stats(h, st);
  HIP_TRY(hipEventRecord(P.lanes[0].done, st));
  HIP_TRY(hipStreamWaitEvent(user, P.lanes[0].done, 0));
  h->train_S = S;
  h->train_prec = 4;
}

template <typename T>
void unpack_grads(const rn_potgnn *h, const T *gp, T *out, bool buffers);

template <typename T>
void train_backward(rn_potgnn *h, const T *dvec6, T *grads /* null: leave the gradients on the device */) {
  Precision<T> &P = prec<T>(h);
  const int S = h->train_S;
  if (S <= 0 || h->train_prec != (int)sizeof(T))
    throw HipError{hipErrorInvalidValue, "train_backward without a train_forward of the same precision"};
  ChunkRun<T> c(h, P.lanes[0], h->io_pos.as<double>(), S, nullptr, nullptr, nullptr);
  c.d_lat = h->train_lat ? h->io_lat.as<T>() : nullptr;      // what the pending train_forward ran with
  c.d_types = h->train_types ? h->io_types.as<int>() : nullptr;
  hipStream_t st = c.st();
  DeviceBuf &seeds = P.seeds;
  seeds.ensure((size_t)S * 6 * sizeof(T));
  HIP_TRY(hipMemcpy(seeds.p, dvec6, (size_t)S * 6 * sizeof(T), hipMemcpyHostToDevice));
  P.grad.ensure(h->lay.total * sizeof(T));
  HIP_TRY(hipMemsetAsync(P.grad.p, 0, h->lay.total * sizeof(T), st));
  Reverse<T> rv{S, 1, seeds.template as<T>(), nullptr, P.grad.template as<T>(), true};
  reverse_pass<T>(h, c, rv);
  HIP_TRY(hipStreamSynchronize(st));
  h->train_S = 0;  // (the pending forward is consumed whether or not the check below throws)
  check_ps_fail(h);
  if (sizeof(T) == 4) h->grads_on_device = true;
  if (!grads) return;
  std::vector<T> gp(h->lay.total);
  HIP_TRY(hipMemcpy(gp.data(), P.grad.p, gp.size() * sizeof(T), hipMemcpyDeviceToHost));
  unpack_grads<T>(h, gp.data(), grads, false);
}

// Reverse pass of a pending train_forward(_device) with the cotangents [S][6] in a DEVICE buffer; the gradients stay in HBM
// (device-resident training), nothing is synchronised: `user` waits for the lane, the Adam step runs on the lane's stream.
inline void train_backward_device(rn_potgnn *h, const float *d_dvec6, hipStream_t user) {
  Precision<float> &P = h->f32;
  const int S = h->train_S;
  if (S <= 0 || h->train_prec != 4)
    throw HipError{hipErrorInvalidValue, "train_backward without a train_forward of the same precision"};
  ChunkRun<float> c(h, P.lanes[0], h->io_pos.as<double>(), S, nullptr, nullptr, nullptr);
  c.d_lat = h->train_lat ? h->io_lat.as<float>() : nullptr;
  c.d_types = h->train_types ? h->io_types.as<int>() : nullptr;
  hipStream_t st = c.st();
  HIP_TRY(hipEventRecord(h->ev_start, user));
  HIP_TRY(hipStreamWaitEvent(st, h->ev_start, 0));
  P.seeds.ensure((size_t)S * 6 * sizeof(float));
  HIP_TRY(hipMemcpyAsync(P.seeds.p, d_dvec6, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToDevice, st));
  P.grad.ensure(h->lay.total * sizeof(float));
  HIP_TRY(hipMemsetAsync(P.grad.p, 0, h->lay.total * sizeof(float), st));
  Reverse<float> rv{S, 1, P.seeds.as<float>(), nullptr, P.grad.as<float>(), true};
  reverse_pass<float>(h, c, rv);
  HIP_TRY(hipEventRecord(P.lanes[0].done, st));
  HIP_TRY(hipStreamWaitEvent(user, P.lanes[0].done, 0));
  h->train_S = 0;
  h->grads_on_device = true;
}

// inverse of pack_weights for a gradient blob in the packed layout -> state_dict order
template <typename T>
void unpack_grads(const rn_potgnn *h, const T *gp, T *out, bool buffers /* true: gp holds WEIGHTS */) {
  const int K = h->cfg.num_atom_types, Fn = h->d.Fn, Fe = h->d.Fe, FnP = h->d.FnP, FeP = h->d.FeP,
            P = h->cfg.num_message_passes;
  const int HP = std::max(FeP, 32);
  const PackedLayout &L = h->lay;
  T *c = out;
  auto copy = [&](size_t src, size_t n) {
    std::memcpy(c, gp + src, n * sizeof(T));
    c += n;
  };
  auto zeros = [&](size_t n) {
    std::memset(c, 0, n * sizeof(T));
    c += n;
  };
  auto buffer = [&](size_t src, size_t n) {  // a non-trainable entry: zero gradient / its value
    if (buffers) copy(src, n);
    else zeros(n);
  };
  copy(L.emb, (size_t)K * Fn);
  copy(L.W2, (size_t)Fn * Fn);
  copy(L.b2, Fn);
  copy(L.W4, (size_t)Fn * Fn);
  copy(L.b4, Fn);
  buffer(L.offsets, Fe);  // "_edge_embedding.offset" is a buffer
  for (int p = 0; p < P; ++p) {
    const auto &q = L.pass[p];
    for (int r = 0; r < 2 * Fn; ++r) {
      const int col = gated_col(r, Fn, FnP);
      for (int k = 0; k < Fn; ++k) *c++ = gp[q.c1_WnT + (size_t)k * 2 * FnP + col];
      for (int k = 0; k < Fe; ++k) *c++ = gp[q.c1_WeT + (size_t)k * 2 * FnP + col];
    }
    for (int r = 0; r < 2 * Fn; ++r) *c++ = gp[q.c1_bias + gated_col(r, Fn, FnP)];
    for (int r = 0; r < 2 * Fn; ++r) *c++ = gp[q.c1n_g + gated_col(r, Fn, FnP)];
    for (int r = 0; r < 2 * Fn; ++r) *c++ = gp[q.c1n_b + gated_col(r, Fn, FnP)];
    copy(q.fin_g, Fn);
    copy(q.fin_b, Fn);
  }
  for (int p = 0; p < P; ++p) {
    const auto &q = L.pass[p];
    for (int r = 0; r < 2 * Fe; ++r) {
      const int col = gated_col(r, Fe, FeP);
      for (int k = 0; k < Fn; ++k) *c++ = gp[q.c2_WT + (size_t)k * 2 * FeP + col];
    }
    for (int r = 0; r < 2 * Fe; ++r) *c++ = gp[q.c2_bias + gated_col(r, Fe, FeP)];
    for (int r = 0; r < 2 * Fe; ++r) {
      const int col = gated_col(r, Fe, FeP);
      for (int blk = 0; blk < 3; ++blk)
        for (int k = 0; k < Fn; ++k) *c++ = gp[q.c3_WnT + (size_t)k * 6 * FeP + blk * 2 * FeP + col];
      for (int blk = 0; blk < 2; ++blk)
        for (int k = 0; k < Fe; ++k) *c++ = gp[q.c3_WeT + (size_t)k * 4 * FeP + blk * 2 * FeP + col];
    }
    for (int r = 0; r < 2 * Fe; ++r) *c++ = gp[q.c3_nshift + 2 * FeP + gated_col(r, Fe, FeP)];
    for (int r = 0; r < 2 * Fe; ++r) *c++ = gp[q.c2n1_g + gated_col(r, Fe, FeP)];
    for (int r = 0; r < 2 * Fe; ++r) *c++ = gp[q.c2n1_b + gated_col(r, Fe, FeP)];
    for (int r = 0; r < 2 * Fe; ++r) *c++ = gp[q.c3n1_g + gated_col(r, Fe, FeP)];
    for (int r = 0; r < 2 * Fe; ++r) *c++ = gp[q.c3n1_b + gated_col(r, Fe, FeP)];
    copy(q.c2n2_g, Fe);
    copy(q.c2n2_b, Fe);
    copy(q.c3n2_g, Fe);
    copy(q.c3n2_b, Fe);
  }
  for (int r = 0; r < Fe; ++r)
    for (int k = 0; k < Fe; ++k) *c++ = gp[L.W0T + (size_t)k * HP + r];
  copy(L.b0p, Fe);
  copy(L.bn_w, Fe);
  copy(L.bn_b, Fe);
  buffer(L.bn_rm, Fe);  // running_mean
  buffer(L.bn_rv, Fe);  // running_var
  for (int r = 0; r < Fe; ++r)
    for (int k = 0; k < Fe; ++k) *c++ = gp[L.W3T + (size_t)k * HP + r];
  copy(L.b3, Fe);
  for (int r = 0; r < 12; ++r)
    for (int k = 0; k < Fe; ++k) *c++ = gp[L.W5T + (size_t)k * 32 + r];
  copy(L.b5, 12);
}

// Every entry point that touches a handle runs inside guarded(): calls on ONE handle are serialised by the handle's own
// lock (they share its workspaces, streams and tape), so a handle may be used from several threads; distinct handles are
// independent.  No exception leaves the C ABI.
int guarded(rn_potgnn *h, const std::function<void()> &fn) {
  std::unique_lock<std::recursive_mutex> lock;
  if (h) lock = std::unique_lock<std::recursive_mutex>(h->lock);
  try {
    if (h) HIP_TRY(hipSetDevice(h->cfg.device));
    fn();
    return RN_OK;
  } catch (const HipError &e) {
    set_error(h, "HIP error %d (%s) in %s", (int)e.code, hipGetErrorString(e.code), e.what);
    return e.code == hipErrorOutOfMemory ? RN_ERR_OUT_OF_MEMORY : RN_ERR_HIP;
  } catch (const std::bad_alloc &) {
    set_error(h, "host allocation failed");
    return RN_ERR_OUT_OF_MEMORY;
  } catch (const std::exception &e) {
    set_error(h, "internal error: %s", e.what());
    return RN_ERR_HIP;
  } catch (...) {
    set_error(h, "internal error: unknown exception");
    return RN_ERR_HIP;
  }
}
}  // namespace

// ================================================================================ C ABI
extern "C" {

const char *rn_potgnn_version(void) { return "ramannoodle_amd-potgnn 0.1 (gfx950)"; }

size_t rn_potgnn_weight_count(const rn_potgnn_config *c) {
  if (!c) return 0;
  const size_t K = c->num_atom_types, Fn = c->size_node_embedding, Fe = c->size_edge_embedding,
               P = c->num_message_passes;
  size_t n = K * Fn + 2 * (Fn * Fn + Fn) + Fe;
  n += P * (2 * Fn * (Fn + Fe) + 2 * Fn + 2 * (2 * Fn) + 2 * Fn);
  n += P * (2 * Fe * Fn + 2 * Fe + 2 * Fe * (3 * Fn + 2 * Fe) + 2 * Fe + 4 * (2 * Fe) + 4 * Fe);
  n += (Fe * Fe + Fe) + 4 * Fe + (Fe * Fe + Fe) + (12 * Fe + 12);
  return n;
}

const char *rn_potgnn_last_error(const rn_potgnn *h) {
  return h ? h->error.c_str() : g_create_error.c_str();
}

int rn_potgnn_create(const rn_potgnn_config *cfg, const int32_t *edge_a, const int32_t *edge_b,
                     const int32_t *atom_types, const double *lattice, const float *weights,
                     size_t num_weights, const double *mean, const double *stddev,
                     rn_potgnn **out) {
  if (!out) return RN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!cfg || !atom_types || !lattice || !weights || !mean || !stddev ||
      (cfg->num_edges > 0 && (!edge_a || !edge_b))) {
    set_error(nullptr, "null argument");
    return RN_ERR_INVALID_ARGUMENT;
  }
  const int N = cfg->num_atoms, E = cfg->num_edges;
  if (N <= 0 || E < 0 || cfg->num_atom_types <= 0 || cfg->size_node_embedding <= 0 ||
      cfg->size_edge_embedding <= 0 || cfg->num_message_passes <= 0) {
    set_error(nullptr, "invalid configuration (non-positive size)");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (E == 0) {
    set_error(nullptr, "reference graph has no edges: the per-structure mean over edges "
                       "(_gnn.py:662-664) is undefined");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (cfg->size_node_embedding > 128 || cfg->size_edge_embedding > 128) {
    set_error(nullptr, "embedding sizes above 128 are not supported (Fn=%d, Fe=%d)",
              cfg->size_node_embedding, cfg->size_edge_embedding);
    return RN_ERR_UNSUPPORTED;
  }
  if (num_weights != rn_potgnn_weight_count(cfg)) {
    set_error(nullptr, "weights has %zu floats, expected %zu", num_weights,
              rn_potgnn_weight_count(cfg));
    return RN_ERR_INVALID_ARGUMENT;
  }
  for (int e = 0; e < E; ++e) {
    if (edge_a[e] < 0 || edge_a[e] >= N || edge_b[e] < 0 || edge_b[e] >= N ||
        edge_a[e] == edge_b[e]) {
      set_error(nullptr, "edge %d = (%d,%d) is out of range or a self loop", e, edge_a[e],
                edge_b[e]);
      return RN_ERR_INVALID_ARGUMENT;
    }
    if (e > 0 && (edge_a[e] < edge_a[e - 1] ||
                  (edge_a[e] == edge_a[e - 1] && edge_b[e] <= edge_b[e - 1]))) {
      set_error(nullptr, "edges must be strictly sorted by (a, b); violated at edge %d", e);
      return RN_ERR_INVALID_ARGUMENT;
    }
  }
  for (int n = 0; n < N; ++n)
    if (atom_types[n] < 0 || atom_types[n] >= cfg->num_atom_types) {
      set_error(nullptr, "atom %d has type %d outside [0,%d)", n, atom_types[n],
                cfg->num_atom_types);
      return RN_ERR_INVALID_ARGUMENT;
    }

  {  // an atom's outgoing-edge rows (in float64) must fit one workgroup's LDS tile
    const size_t cap = max_out_degree(pad_pow2(cfg->size_edge_embedding));
    std::vector<int> deg(N, 0);
    for (int e = 0; e < E; ++e) deg[edge_a[e]]++;
    for (int n = 0; n < N; ++n)
      if ((size_t)deg[n] > cap) {
        set_error(nullptr, "atom %d has %d outgoing edges; more than %zu per atom is unsupported "
                           "for size_edge_embedding=%d", n, deg[n], cap, cfg->size_edge_embedding);
        return RN_ERR_UNSUPPORTED;
      }
  }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 ||
      cfg->device >= ndev) {
    set_error(nullptr, "no usable HIP device (count=%d, requested=%d)", ndev, cfg->device);
    return RN_ERR_NO_DEVICE;
  }

  std::unique_ptr<rn_potgnn> h(new rn_potgnn());
  h->cfg = *cfg;
  h->d = {cfg->size_node_embedding, cfg->size_edge_embedding, pad_pow2(cfg->size_node_embedding),
          pad_pow2(cfg->size_edge_embedding)};
  widen_for_fused(h->d);
  std::memcpy(h->lattice, lattice, sizeof(h->lattice));
  std::memcpy(h->mean, mean, sizeof(h->mean));
  std::memcpy(h->stdv, stddev, sizeof(h->stdv));
  h->keep_stages = getenv("RN_POTGNN_KEEP_STAGES") && atoi(getenv("RN_POTGNN_KEEP_STAGES")) != 0;
  h->debug_sync = getenv("RN_POTGNN_DEBUG_SYNC") && atoi(getenv("RN_POTGNN_DEBUG_SYNC")) != 0;
  if (const char *e = getenv("RN_POTGNN_INTERLEAVE")) h->interleave = atoi(e) != 0;
  if (const char *e = getenv("RN_POTGNN_LANES")) h->num_lanes = std::max(1, std::min(2, atoi(e)));
  if (const char *e = getenv("RN_POTGNN_MFMA")) h->mfma_f16_requested = !(e[0] == 'f' && e[1] == '3');
  h->mfma_f16 = h->mfma_f16_requested;
  if (const char *e = getenv("RN_POTGNN_TAPE_FUSED")) h->tape_fused = atoi(e) != 0;
  if (const char *e = getenv("RN_POTGNN_TAPE_PS")) h->tape_ps = atoi(e) != 0;

  // ---- graph: CSR over a (edges are already grouped), CSR over b, tiles, triplet offsets
  h->edge_a.assign(edge_a, edge_a + E);
  h->edge_b.assign(edge_b, edge_b + E);
  h->atom_type.assign(atom_types, atom_types + N);
  h->out_ptr.assign(N + 1, 0);
  h->in_ptr.assign(N + 1, 0);
  for (int e = 0; e < E; ++e) {
    h->out_ptr[edge_a[e] + 1]++;
    h->in_ptr[edge_b[e] + 1]++;
  }
  for (int n = 0; n < N; ++n) {
    h->out_ptr[n + 1] += h->out_ptr[n];
    h->in_ptr[n + 1] += h->in_ptr[n];
  }
  h->in_edge.assign(E, 0);
  {
    std::vector<int> fill(h->in_ptr.begin(), h->in_ptr.end() - 1);
    for (int e = 0; e < E; ++e) h->in_edge[fill[edge_b[e]]++] = e;  // ascending edge id per b
  }
  const Dims d = h->d;
  // node tiles: consecutive atoms whose outgoing-edge rows fit an LDS budget (counted in
  // float32 rows; the float64 path uses twice the bytes for the same tiles).  A workgroup
  // serves its tile's destination edges G at a time (G lane groups), so the budget is chosen
  // to waste as few lane groups in the last round as possible (18 in-edges per atom and
  // G = 16: 4 atoms per tile idle 10 % of the groups, 6 atoms 4 %).
  const size_t row_bytes = (size_t)2 * d.FeP * sizeof(float);  // (the out-degree cap was checked before the device probe)
  auto build_tiles = [&](size_t budget_rows, std::vector<int> &tb) {
    tb.assign(1, 0);
    int rows = 0, max_rows = 0;
    for (int n = 0; n < N; ++n) {
      const int deg = h->out_ptr[n + 1] - h->out_ptr[n];
      if (rows > 0 && (size_t)(rows + deg) > budget_rows) {
        tb.push_back(n);
        rows = 0;
      }
      rows += deg;
      max_rows = std::max(max_rows, rows);
    }
    tb.push_back(N);
    return max_rows;
  };
  const bool vpl8 = getenv("RN_POTGNN_VPL") && atoi(getenv("RN_POTGNN_VPL")) == 8;
  // fused EdgeBlock (kernels_fused.hip): its LDS footprint bounds the tile instead
  const bool want_fused = getenv("RN_POTGNN_FUSED") ? atoi(getenv("RN_POTGNN_FUSED")) != 0 : true;
  const bool fused_mode = want_fused && d.FnP == 64 && d.FeP == 64;
  // Experiment builds (-DRN_EXPERIMENTS=1) only -- RN_POTGNN_EDGE2=1: the frame-pipelined form of the fused EdgeBlock
  // (edge_block2_kernel + edge_c2_kernel), measured level with the per-frame form (profiles/r03/edge2_experiment.txt).
#if RN_EXPERIMENTS
  const bool want_edge2 = getenv("RN_POTGNN_EDGE2") ? atoi(getenv("RN_POTGNN_EDGE2")) != 0 : false;
#else
  const bool want_edge2 = false;
  (void)want_edge2;
#endif
  // narrow-width kernels (kernels_narrow.hip): one lane per destination edge, so a tile should bring
  // about one workgroup's worth (256) of destination edges and keep its LDS rows within ~40 KiB
  const bool want_narrow = getenv("RN_POTGNN_NARROW") ? atoi(getenv("RN_POTGNN_NARROW")) != 0 : true;
  const bool narrow_mode = want_narrow && narrow_supported(d);
  int max_rows = 0;
  if (narrow_mode && !getenv("RN_POTGNN_TILE_KB")) {
    const size_t per_row = edge_narrow_lds_bytes(d.Fn, d.Fe, 1024, 1024) / 1024 + 1;
    // 128 rows = one two-wave workgroup per tile (kernels_narrow.hip launch_edge_cfg); an atom with more out-edges gets a
    // tile of its own and the four-wave form
    size_t budget = std::min<size_t>(128, (size_t)40 * 1024 / per_row);
    if (const char *e = getenv("RN_POTGNN_NARROW_TILE_ROWS")) budget = (size_t)std::max(1, atoi(e));  // experiment knob
    max_rows = build_tiles(std::max<size_t>(1, budget), h->tile_begin);
    // The greedy partition fills every tile but the last (256 atoms of degree 18: eighteen tiles of 14 atoms and one
    // of 4).  The same NUMBER of tiles with boundaries at equal shares of the edge list (14, 13, 14, 13, ...) costs
    // the same lane slots and keeps the workgroups of a frame in step, as long as no tile exceeds the budget.
    if (h->tile_begin.size() > 2) {
      const int T = (int)h->tile_begin.size() - 1;
      std::vector<int> tb(1, 0);
      for (int t = 1; t < T; ++t) {
        const int64_t want = ((int64_t)E * t + T - 1) / T;
        int n = tb.back();
        while (n < N && h->out_ptr[n] < want) ++n;
        tb.push_back(std::max(n, tb.back()));
      }
      tb.push_back(N);
      int worst_out = 0, worst_in = 0, greedy_in = 0;
      bool ok = true;
      for (int t = 0; t < T; ++t) {
        ok = ok && tb[t + 1] > tb[t];
        worst_out = std::max(worst_out, h->out_ptr[tb[t + 1]] - h->out_ptr[tb[t]]);
        worst_in = std::max(worst_in, h->in_ptr[tb[t + 1]] - h->in_ptr[tb[t]]);
        greedy_in = std::max(greedy_in, h->in_ptr[h->tile_begin[t + 1]] - h->in_ptr[h->tile_begin[t]]);
      }
      if (ok && (size_t)worst_out <= budget && worst_in <= std::max(greedy_in, (int)budget)) {
        h->tile_begin = tb;
        max_rows = worst_out;
      }
    }
  } else if (getenv("RN_POTGNN_TILE_KB") || vpl8) {
    const size_t tile_kb = getenv("RN_POTGNN_TILE_KB") ? (size_t)atoi(getenv("RN_POTGNN_TILE_KB")) : 64;
    max_rows = build_tiles(std::max<size_t>(1, tile_kb * 1024 / row_bytes), h->tile_begin);
  } else {
    // relative cost of one frame = (rounds of the slowest tile) x (tiles sharing the chip),
    // among budgets whose whole LDS footprint stays within 64 KiB (measured: beyond that
    // only one workgroup per CU runs)
    const int G = 256 / std::max(1, d.FeP / 4);
    double best = 0;
    std::vector<int> tb;
    for (size_t kb = 8; kb <= 62; kb += 2) {
      const int mr = build_tiles(std::max<size_t>(1, kb * 1024 / row_bytes), tb);
      int rounds = 1, max_in = 0, max_nodes = 0;
      for (size_t t = 0; t + 1 < tb.size(); ++t) {
        const int din = h->in_ptr[tb[t + 1]] - h->in_ptr[tb[t]];
        rounds = std::max(rounds, (din + G - 1) / G);
        max_in = std::max(max_in, din);
        max_nodes = std::max(max_nodes, tb[t + 1] - tb[t]);
      }
#if RN_EXPERIMENTS
      const size_t fused_lds_need = want_edge2 ? edge2_lds_bytes(mr, max_in, max_nodes) : edge_fused_lds_bytes(mr, max_in, max_nodes);
#else
      const size_t fused_lds_need = edge_fused_lds_bytes(mr, max_in, max_nodes);
#endif
      const size_t lds = fused_mode ? fused_lds_need
                                    : (size_t)mr * (row_bytes + 4) + (size_t)max_nodes * row_bytes +
                                          (size_t)12 * d.FeP * 4 + (size_t)mr * 4 + (size_t)max_in * 24 + 96;
      // unfused: two aggregation workgroups + one projection workgroup (34 KiB) share a CU
      const size_t lds_cap = fused_mode ? kFusedLdsBudget : (size_t)63 * 1024;
      if (lds > lds_cap && !h->tile_begin.empty()) break;
      const double cost = (double)rounds * (double)(tb.size() - 1);
      if (h->tile_begin.empty() || cost < best * 0.995) {
        best = cost;
        h->tile_begin = tb;
        max_rows = mr;
      }
    }
  }
  // node tiles of the twelve-wave EdgeBlock (edge_block3_kernel): one 768-thread workgroup per CU with the CU's LDS,
  // 48 destinations per round -> as few rounds as possible in total; among equals the larger tiles (fewer per-tile phases)
  int et_max_rows = 0, et_max_in = 0, et_max_nodes = 0;
#if RN_EXPERIMENTS
  const bool want_edge3 = getenv("RN_POTGNN_EDGE3") ? atoi(getenv("RN_POTGNN_EDGE3")) != 0 : false;
  if (fused_mode && want_edge3) {
    double best = 0;
    std::vector<int> tb;
    const int forced = getenv("RN_POTGNN_EDGE3_TILE_ROWS") ? atoi(getenv("RN_POTGNN_EDGE3_TILE_ROWS")) : 0;  // experiment knob
    for (size_t budget = forced > 0 ? forced : 16; budget <= (size_t)(forced > 0 ? forced : 300); budget += 2) {
      const int mr = build_tiles(budget, tb);
      int max_in = 0, max_nodes = 0;
      double cost = 0;
      for (size_t t = 0; t + 1 < tb.size(); ++t) {
        const int din = h->in_ptr[tb[t + 1]] - h->in_ptr[tb[t]];
        cost += (double)((din + edge3_dests_per_round() - 1) / edge3_dests_per_round());
        max_in = std::max(max_in, din);
        max_nodes = std::max(max_nodes, tb[t + 1] - tb[t]);
      }
      if (edge3_lds_bytes(mr, max_in, max_nodes) > kEdge3LdsBudget) {
        if (!h->et_begin.empty()) break;
        continue;
      }
      if (h->et_begin.empty() || cost <= best) {
        best = cost;
        h->et_begin = tb;
        et_max_rows = mr;
        et_max_in = max_in;
        et_max_nodes = max_nodes;
      }
    }
  }
#else
  const bool want_edge3 = false;
#endif
  // node tiles of the EdgeBlock reverse kernel: the largest whose float32 LDS footprint leaves room for two workgroups
  // per CU (RN_POTGNN_BWD_TILES=0: the forward kernel's tiles, one 512-thread workgroup per CU)
  int bt_max_rows = 0, bt_max_in = 0, bt_max_nodes = 0;
  if (!(getenv("RN_POTGNN_BWD_TILES") && atoi(getenv("RN_POTGNN_BWD_TILES")) == 0)) {
    std::vector<int> tb;
    for (size_t budget = 1; budget <= 512; ++budget) {
      const int mr = build_tiles(budget, tb);
      int max_in = 0, max_nodes = 0;
      for (size_t t = 0; t + 1 < tb.size(); ++t) {
        max_in = std::max(max_in, h->in_ptr[tb[t + 1]] - h->in_ptr[tb[t]]);
        max_nodes = std::max(max_nodes, tb[t + 1] - tb[t]);
      }
      if (edge_bwd_tile2_lds_bytes(mr, max_in, max_nodes, d.FeP, sizeof(float)) > (size_t)78 * 1024) {
        if (!h->bt_begin.empty()) break;
        continue;
      }
      if (h->bt_begin.empty() || mr > bt_max_rows) {
        h->bt_begin = tb;
        bt_max_rows = mr;
        bt_max_in = max_in;
        bt_max_nodes = max_nodes;
      }
    }
  }
  // node tiles of the fused NodeBlock kernel: consecutive atoms by IN-edges; four workgroups per CU
  // (40 KiB of LDS each), as few rounds x tiles as possible
  int nt_max_in = 0, nt_max_nodes = 0;
  bool nt_narrow = false;
  if (narrow_mode) {
    // node_tiled_kernel (kernels_narrow.hip): ONE WAVE per tile -- at most 64 atoms (one lane each in its last pass)
    // whose in-edge rows are streamed 64 at a time: the partition that fills those chunks best, within 12 KiB of LDS
    // (thirteen waves per CU and more); among equals the larger tiles (fewer frame starts per row)
    double best = -1;
    const int max_budget = getenv("RN_POTGNN_NODE_TILE_ROWS") ? std::max(1, atoi(getenv("RN_POTGNN_NODE_TILE_ROWS"))) : 256;
    const int lds_kb = getenv("RN_POTGNN_NODE_TILE_KB") ? std::max(1, atoi(getenv("RN_POTGNN_NODE_TILE_KB"))) : 12;
    for (int budget = 32; budget <= std::max(max_budget, 32); budget += 2) {
      std::vector<int> tb(1, 0);
      int rows_in = 0, max_in = 0, max_nodes = 0, first = 0;
      double work = 0;  // chunk slots the partition pays for
      auto close = [&](int n) {
        max_nodes = std::max(max_nodes, n - first);
        work += (double)std::max((rows_in + 63) / 64, 1) * 64.0;
      };
      for (int n = 0; n < N; ++n) {
        const int deg = h->in_ptr[n + 1] - h->in_ptr[n];
        if (n > first && (rows_in + deg > budget || n - first >= 64)) {
          close(n);
          tb.push_back(n);
          first = n;
          rows_in = 0;
        }
        rows_in += deg;
        max_in = std::max(max_in, rows_in);
      }
      close(N);
      tb.push_back(N);
      // (a single atom with more in-edges than the budget still gets its tile: the LDS check below decides)
      if (node_tiled_lds_bytes(d.Fn, d.Fe, max_in, max_nodes) > (size_t)(h->nt_begin.empty() ? 64 : lds_kb) * 1024) {
        if (!h->nt_begin.empty()) break;
        continue;
      }
      const double fill = (double)E / std::max(work, 1.0);
      if (fill >= best * 0.999) {
        best = std::max(best, fill);
        h->nt_begin = tb;
        nt_max_in = max_in;
        nt_max_nodes = max_nodes;
      }
    }
    nt_narrow = !h->nt_begin.empty();
  }
  if (!nt_narrow) {
    double best = 0;
#if RN_EXPERIMENTS
    const bool node_wave = node_fused_wave_tiles() && d.FnP == 64 && d.FeP == 64;
#else
    const bool node_wave = false;
#endif
    const int forced = getenv("RN_POTGNN_NODE_TILE_ROWS") ? atoi(getenv("RN_POTGNN_NODE_TILE_ROWS")) : 0;  // experiment knob
    for (int budget = forced > 0 ? forced : 16; budget <= (forced > 0 ? forced : 256); budget += 8) {
      std::vector<int> tb(1, 0);
      int rows_in = 0, max_in = 0, max_nodes = 0, first = 0;
      for (int n = 0; n < N; ++n) {
        const int deg = h->in_ptr[n + 1] - h->in_ptr[n];
        if (n > first && rows_in + deg > budget) {
          max_nodes = std::max(max_nodes, n - first);
          tb.push_back(n);
          first = n;
          rows_in = 0;
        }
        rows_in += deg;
        max_in = std::max(max_in, rows_in);
      }
      max_nodes = std::max(max_nodes, N - first);
      tb.push_back(N);
      double cost;
      if (node_wave) {
        // wave-autonomous kernel: a workgroup's four waves take the tile's 16-row pieces four at a time;
        // two workgroups per CU (register-bound), so up to 72 KiB of LDS each
#if RN_EXPERIMENTS
        if (node_wave_lds_bytes(max_in, max_nodes) > (size_t)72 * 1024 && !h->nt_begin.empty()) break;
#endif
        cost = 0;
        for (size_t t = 0; t + 1 < tb.size(); ++t) {
          const int rows_t = h->in_ptr[tb[t + 1]] - h->in_ptr[tb[t]];
          cost += (double)(((rows_t + 15) / 16 + 3) / 4);
        }
      } else {
        if (node_fused_lds_bytes(max_in, max_nodes) > (size_t)40 * 1024 && !h->nt_begin.empty()) break;
        cost = (double)((max_in + 15) / 16) * (double)(tb.size() - 1);
      }
      if (h->nt_begin.empty() || cost < best * 0.995) {
        best = cost;
        h->nt_begin = tb;
        nt_max_in = max_in;
        nt_max_nodes = max_nodes;
      }
    }
  }
  h->trip_off.assign(E + 1, 0);
  for (int e = 0; e < E; ++e) {
    const int bd = edge_b[e], ad = edge_a[e];
    int cnt = 0;
    for (int o = h->out_ptr[bd]; o < h->out_ptr[bd + 1]; ++o) cnt += edge_b[o] != ad;
    h->trip_off[e + 1] = h->trip_off[e] + cnt;
  }

  h->in_pos.assign(E, 0);
  for (int i = 0; i < E; ++i) h->in_pos[h->in_edge[i]] = i;  // position of edge e in the (b, a) order
  h->rev_edge.assign(E, -1);
  for (int e = 0; e < E; ++e) {  // reverse edge (b -> a): binary search in b's sorted out-list
    const int bd = edge_b[e], ad = edge_a[e];
    const int *lo = edge_b + h->out_ptr[bd], *hi = edge_b + h->out_ptr[bd + 1];
    const int *it = std::lower_bound(lo, hi, ad);
    if (it != hi && *it == ad) h->rev_edge[e] = (int)(it - edge_b);
  }

  rn_potgnn *hp = h.get();
  int rc = guarded(nullptr, [&]() {
    HIP_TRY(hipSetDevice(cfg->device));
    HIP_TRY(hipEventCreateWithFlags(&hp->ev_start, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&hp->ev_g[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&hp->ev_g[1], hipEventDisableTiming));
    // node tiles of the role-specialised EdgeBlock (kernels_edge_ps.hip): ONE twelve-wave workgroup per CU, 16 destinations
    // per round.  A launch runs floor(CUs / tiles) frame groups side by side, so the cost of a partition is (rounds of its
    // slowest tile) / (frame groups); a partition is admissible when every tile passes the producers' schedule check and
    // the kernel's LDS footprint fits the CU.
    int pt_max_rows = 0, pt_max_in = 0, pt_back = 2, pt_gram = 0;
    const bool want_ps = getenv("RN_POTGNN_EDGE_PS") ? atoi(getenv("RN_POTGNN_EDGE_PS")) != 0 : true;
    if (fused_mode && want_ps) {
      int cus = 256;
      {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
      }
      double best = 0;
      std::vector<int> tb, rb, re;
      const int forced = getenv("RN_POTGNN_PS_TILE_ROWS") ? atoi(getenv("RN_POTGNN_PS_TILE_ROWS")) : 0;  // experiment knob
      const bool want_back3 = !(getenv("RN_POTGNN_PS_BACK") && atoi(getenv("RN_POTGNN_PS_BACK")) == 2);
      // GRAM is opt-in (RN_POTGNN_PS_GRAM=1): parity-green, but the producers' Gram phase costs them more than the consumers' loop
      // gains while the producers are the slower role with it (profiles/r05/edge_ps_experiments.txt: 7.38 against 7.01 ms per launch)
      const bool want_gram = getenv("RN_POTGNN_PS_GRAM") && atoi(getenv("RN_POTGNN_PS_GRAM")) != 0;
      // One variant of the kernel for a partition: {gram, back}.  GRAM (the LayerNorm cross terms on the matrix pipe) has a ring
      // of 7 tiles and needs every round's window to span <= 3 of them, every destination <= 32 source rows, and its tables
      // inside the CU's LDS; back = 3 (the producers three rounds ahead of the slower consumer set) needs the room in the ring.
      // (back = 4: with eight destinations per consumer wave four rounds are in flight at a time -- the eight-lane form of
      //  the kernel --, and a step that may only rewrite the ring behind round g - 4 would stall the producers)
      struct Variant { bool gram; int back; double factor; };
      const int max_back = getenv("RN_POTGNN_PS_BACK") ? atoi(getenv("RN_POTGNN_PS_BACK")) : 5;
      const Variant variants[6] = {{true, 3, 0.88}, {true, 2, 0.94}, {false, 5, 0.94}, {false, 4, 0.96}, {false, 3, 1.0}, {false, 2, 1.07}};
      for (size_t budget = forced > 0 ? forced : 8; budget <= (size_t)(forced > 0 ? forced : 1024); budget += 2) {
        const int mr = build_tiles(budget, tb);
        const int ntiles = (int)tb.size() - 1;
        for (const Variant &v : variants) {
          if ((v.gram && !want_gram) || (v.back == 3 && !want_back3) || v.back > max_back) continue;
          int max_in = 0, max_rounds = 1;
          bool ok = true;
          for (int t = 0; t < ntiles && ok; ++t) {
            const int eo0 = hp->out_ptr[tb[t]];
            rb.clear();
            re.clear();
            int longest = 0;
            for (int i = hp->in_ptr[tb[t]]; i < hp->in_ptr[tb[t + 1]]; ++i) {
              const int bd = edge_b[hp->in_edge[i]];
              rb.push_back(hp->out_ptr[bd] - eo0);
              re.push_back(hp->out_ptr[bd + 1] - eo0);
              longest = std::max(longest, re.back() - rb.back());
            }
            const int din = (int)rb.size();
            max_in = std::max(max_in, din);
            max_rounds = std::max(max_rounds, (din + 15) / 16);
            int window = 0;
            ok = edge_ps_tile_ok(rb.data(), re.data(), din, v.back, edge_ps_ring_tiles(v.gram), &window);
            if (v.gram) ok = ok && window <= edge_ps_gram_window() && longest <= 32;
          }
          if (!ok || edge_ps_lds_bytes(mr, max_in, v.gram) > (size_t)160 * 1024) continue;
          const double groups = ntiles <= cus ? (double)(cus / ntiles) : 1.0 / (double)((ntiles + cus - 1) / cus);
          const double cost = (double)max_rounds / groups * v.factor;
          if (hp->pt_begin.empty() || cost < best * 0.999) {
            best = cost;
            hp->pt_begin = tb;
            pt_max_rows = mr;
            pt_max_in = max_in;
            pt_back = v.back;
            pt_gram = v.gram ? 1 : 0;
          }
          break;  // (the variants are ordered by their factor: the first admissible one is this partition's)
        }
        if (mr >= E) break;  // one tile holds everything: larger budgets change nothing
      }
    }
    // upload all int arrays in one allocation
    std::vector<int> ints;
    auto push = [&](const std::vector<int> &v) {
      size_t o = ints.size();
      ints.insert(ints.end(), v.begin(), v.end());
      while (ints.size() % 4) ints.push_back(0);
      return o;
    };
    const size_t o_a = push(hp->edge_a), o_b = push(hp->edge_b), o_op = push(hp->out_ptr),
                 o_ip = push(hp->in_ptr), o_ie = push(hp->in_edge), o_at = push(hp->atom_type),
                 o_tb = push(hp->tile_begin), o_to = push(hp->trip_off), o_rv = push(hp->rev_edge),
                 o_nt = push(hp->nt_begin), o_et = push(hp->et_begin), o_bt = push(hp->bt_begin), o_pt = push(hp->pt_begin), o_ipos = push(hp->in_pos);
    hp->g_ints.ensure(ints.size() * sizeof(int));
    HIP_TRY(hipMemcpy(hp->g_ints.p, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice));
    const int *base = hp->g_ints.as<int>();
    Graph &g = hp->g;
    g.N = N;
    g.E = E;
    g.edge_a = base + o_a;
    g.edge_b = base + o_b;
    g.out_ptr = base + o_op;
    g.in_ptr = base + o_ip;
    g.in_edge = base + o_ie;
    g.in_pos = base + o_ipos;
    g.atom_type = base + o_at;
    g.rev_edge = base + o_rv;
    g.num_tiles = (int)hp->tile_begin.size() - 1;
    g.tile_begin = base + o_tb;
    g.max_tile_out_rows = max_rows;
    g.max_tile_in_rows = 0;
    g.max_tile_nodes = 0;
    for (size_t t = 0; t + 1 < hp->tile_begin.size(); ++t) {
      g.max_tile_in_rows = std::max(g.max_tile_in_rows,
                                    hp->in_ptr[hp->tile_begin[t + 1]] - hp->in_ptr[hp->tile_begin[t]]);
      g.max_tile_nodes = std::max(g.max_tile_nodes, hp->tile_begin[t + 1] - hp->tile_begin[t]);
    }
    g.trip_off = base + o_to;
    g.T = hp->trip_off[E];
    g.nt_num = (int)hp->nt_begin.size() - 1;
    g.nt_begin = base + o_nt;
    g.nt_max_in_rows = nt_max_in;
    g.nt_max_nodes = nt_max_nodes;
    g.nt_narrow = nt_narrow ? 1 : 0;
    {
      // the atom-owning NodeBlock pays max-in-degree rounds per 16-atom tile; the row-ordered one ceil(rows / 16) per tile of
      // its own partition.  A round of the former is ~25 % cheaper (one barrier, no LDS pass per row): take it unless the
      // in-degrees are so uneven that it runs > 1.2x the rounds.  RN_POTGNN_NODE_ATOM=0 / 1 forces.
      int max_deg = 0;
      long rounds_atom = 0, rounds_row = 0;
      for (int n0 = 0; n0 < N; n0 += 16) {
        int m = 0;
        for (int n = n0; n < std::min(N, n0 + 16); ++n) m = std::max(m, hp->in_ptr[n + 1] - hp->in_ptr[n]);
        rounds_atom += m;
        max_deg = std::max(max_deg, m);
      }
      for (size_t t = 0; t + 1 < hp->nt_begin.size(); ++t)
        rounds_row += (hp->in_ptr[hp->nt_begin[t + 1]] - hp->in_ptr[hp->nt_begin[t]] + 15) / 16;
      bool ok = !nt_narrow && hp->d.FnP == 64 && hp->d.FeP == 64 && node_atom_lds_bytes(max_deg) <= (size_t)40 * 1024 &&
                (double)rounds_atom <= 1.2 * (double)rounds_row;
      if (const char *e = getenv("RN_POTGNN_NODE_ATOM"))
        ok = atoi(e) != 0 && hp->d.FnP == 64 && hp->d.FeP == 64 && node_atom_lds_bytes(max_deg) <= (size_t)64 * 1024;
      g.na_num = ok ? (N + 15) / 16 : 0;
      g.na_max_deg = max_deg;
    }
    g.et_num = hp->et_begin.empty() ? 0 : (int)hp->et_begin.size() - 1;
    g.et_begin = base + o_et;
    g.et_max_out_rows = et_max_rows;
    g.et_max_in_rows = et_max_in;
    g.et_max_nodes = et_max_nodes;
    g.pt_num = hp->pt_begin.empty() ? 0 : (int)hp->pt_begin.size() - 1;
    g.pt_begin = base + o_pt;
    g.pt_max_out_rows = pt_max_rows;
    g.pt_max_in_rows = pt_max_in;
    g.pt_back = pt_back;
    g.pt_gram = pt_gram;
    g.bt_num = hp->bt_begin.empty() ? 0 : (int)hp->bt_begin.size() - 1;
    g.bt_begin = base + o_bt;
    g.bt_max_out_rows = bt_max_rows;
    g.bt_max_in_rows = bt_max_in;
    g.bt_max_nodes = bt_max_nodes;
    double ms[18];
    std::memcpy(ms, hp->mean, sizeof(hp->mean));
    std::memcpy(ms + 9, hp->stdv, sizeof(hp->stdv));
    hp->d_mean_std.ensure(sizeof(ms));
    HIP_TRY(hipMemcpy(hp->d_mean_std.p, ms, sizeof(ms), hipMemcpyHostToDevice));

    pack_weights(hp, weights);
    refresh_mfma_mode(hp);
    // The fused kernels (kernels_fused.hip) are the default where they apply (float32, Fn and
    // Fe padded to 64); RN_POTGNN_FUSED=0 selects projections + edge_agg_kernel.
#if RN_EXPERIMENTS
    hp->use_edge2 = want_fused && want_edge2 && edge2_supported(hp->g, hp->d);
#endif
    hp->use_fused = hp->use_edge2 || (want_fused && edge_fused_supported(hp->g, hp->d));
#if RN_EXPERIMENTS
    hp->use_edge3 = hp->use_fused && !hp->use_edge2 && want_edge3 && edge3_supported(hp->g, hp->d);
#else
    (void)want_edge3;
#endif
    hp->use_ps = hp->use_fused && !hp->use_edge2 && !hp->use_edge3 && hp->g.pt_num > 0;
    hp->ps_fail.ensure(2048);  // [0] the failure word; timing builds (RN_PS_TIMING) keep their cycle counters from byte 64 on
    HIP_TRY(hipMemset(hp->ps_fail.p, 0, 2048));
    hp->split_projections = getenv("RN_POTGNN_SPLIT_PROJ") ? atoi(getenv("RN_POTGNN_SPLIT_PROJ")) != 0 : true;
    hp->use_narrow = narrow_mode && edge_narrow_lds_bytes(hp->d.Fn, hp->d.Fe, hp->g.max_tile_out_rows,
                                                          hp->g.max_tile_in_rows) <= (size_t)64 * 1024;
    // Chunk size and lanes.  Throughput rises monotonically with the frames per launch
    // (profiles/r01_chunk_sweep.txt, profiles/r01/overlap_experiments.txt section 7) and the
    // Infinity Cache does not reward small chunks.  The fused kernels take a whole CU each, so a
    // second lane has nothing to overlap with: ONE lane with an 8 GiB workspace (of 288 GB HBM)
    // beats two lanes of 1.5 GiB (63.7 k vs 62.1 k structures/s at 4000 frames).  The unfused
    // pipeline keeps two alternating lanes of 1.5 GiB.  Round 4: the role-specialised EdgeBlock is bound by SIMD issue and
    // leaves HBM idle and ~10 KiB of LDS per CU free, so the small HBM-bound kernels of a second lane (geometry, per-atom
    // projections, readout reduction) do run under it: RN_POTGNN_LANES=2 gives 48.9 -> 49.2 k structures/s on config 3.
    // Not the default: kernels of two lanes wait for each other's LDS, and HIP events around a launch then time the wait
    // too (the NodeBlock's HBM figure of the bench line reads 6 % instead of 46 %).
    const bool want_node = getenv("RN_POTGNN_NODE_FUSED") ? atoi(getenv("RN_POTGNN_NODE_FUSED")) != 0 : true;
    hp->use_node_fused = hp->use_fused && want_node && node_fused_lds_bytes(hp->g) <= 64 * 1024;
    hp->want_pair_rows = !(getenv("RN_POTGNN_PAIR_ROWS") && atoi(getenv("RN_POTGNN_PAIR_ROWS")) == 0);
    const bool want_ro = getenv("RN_POTGNN_READOUT_FUSED") ? atoi(getenv("RN_POTGNN_READOUT_FUSED")) != 0 : true;
    hp->use_readout_fused = hp->use_fused && want_ro;
    if (!getenv("RN_POTGNN_LANES")) hp->num_lanes = (hp->use_fused || hp->use_narrow) ? 1 : 2;
    int chunk = cfg->max_chunk_structures;
    if (const char *e = getenv("RN_POTGNN_CHUNK")) chunk = atoi(e);
    if (chunk <= 0) {
      const size_t per = per_structure_elems(hp, lean_workspace(hp)) * sizeof(float);
      size_t budget = (hp->use_fused || hp->use_narrow) ? ((size_t)8192 << 20) : ((size_t)1536 << 20);
      size_t free_b = 0, total_b = 0;  // on a shared GPU: at most 1/8 of what is free right now
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0)
        budget = std::min(budget, std::max<size_t>(free_b / 8 / (size_t)hp->num_lanes, (size_t)64 << 20));
      chunk = (int)std::max<size_t>(1, budget / std::max<size_t>(per, 1));
      chunk = std::min(chunk, 4096);
      // the float64 lanes (created on first use) get their own allowance against the same "an eighth of what is free"
      // rule: up to 24 GiB of the full-width float64 layout (1270 frames at 256 atoms, Fn = Fe = 64), so that the phonon
      // finite differences of config 4 -- 1536 cells -- still run in two launches (chunk_frames<double>)
      hp->f64_budget = (size_t)24 << 30;
      if (free_b > 0) hp->f64_budget = std::min(hp->f64_budget, std::max<size_t>(free_b / 8 / (size_t)hp->num_lanes, (size_t)64 << 20));
    }
    hp->chunk = chunk;
    ensure_precision<float>(hp);
  });
  if (rc != RN_OK) return rc;
  *out = h.release();
  return RN_OK;
}

void rn_potgnn_destroy(rn_potgnn *h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  resolve_timers(h);
  for (int l = 0; l < 2; ++l) {
    if (h->f32.lanes[l].stream) (void)hipStreamDestroy(h->f32.lanes[l].stream);
    if (h->f32.lanes[l].done) (void)hipEventDestroy(h->f32.lanes[l].done);
    if (h->f64.lanes[l].stream) (void)hipStreamDestroy(h->f64.lanes[l].stream);
    if (h->f64.lanes[l].done) (void)hipEventDestroy(h->f64.lanes[l].done);
  }
  if (h->f32.side) (void)hipStreamDestroy(h->f32.side);
  if (h->f64.side) (void)hipStreamDestroy(h->f64.side);
  for (int i = 0; i < 3; ++i) {
    if (h->f32.ev_side[i]) (void)hipEventDestroy(h->f32.ev_side[i]);
    if (h->f64.ev_side[i]) (void)hipEventDestroy(h->f64.ev_side[i]);
  }
  for (auto &sl : h->slots) {
    if (sl.copied) (void)hipEventDestroy(sl.copied);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->exec_stream) (void)hipStreamDestroy(h->exec_stream);
  if (h->ev_start) (void)hipEventDestroy(h->ev_start);
  for (int i = 0; i < 2; ++i)
    if (h->ev_g[i]) (void)hipEventDestroy(h->ev_g[i]);
  if (h->step_host) (void)hipHostFree(h->step_host);
  for (int b = 0; b < 2; ++b) {
    if (h->hstage.pin[b]) (void)hipHostFree(h->hstage.pin[b]);
    if (h->hstage.copied[b]) (void)hipEventDestroy(h->hstage.copied[b]);
  }
  if (h->hstage.done) (void)hipEventDestroy(h->hstage.done);
  if (h->hstage.out_pin) (void)hipHostFree(h->hstage.out_pin);
  delete h;
}

// float64 -> float32, round to nearest even: what the device's (float)double does.  Large batches on a few threads
// (the cast reads 8 and writes 4 bytes per coordinate: at the narrow models' rates one core is the bottleneck).
static void cast_to_float(const double *src, float *dst, size_t n) {
  auto run = [](const double *s, float *d, size_t m) {
    for (size_t i = 0; i < m; ++i) d[i] = (float)s[i];
  };
  const size_t per_thread = (size_t)1 << 17;  // (1 MiB of float64 per thread and more: a 1250-frame block of a trajectory that is
                                              //  not in the CPU's caches casts at DRAM rate, ~1 ms on one core, 0.3 on four)
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t nt = std::min<size_t>(std::min<size_t>(4, hw), n / per_thread);
  if (nt < 2) return run(src, dst, n);
  std::vector<std::thread> pool;
  const size_t share = (n + nt - 1) / nt;
  for (size_t t = 1; t < nt; ++t) {
    const size_t lo = t * share, hi = std::min(n, lo + share);
    if (lo < hi) pool.emplace_back(run, src + lo, dst + lo, hi - lo);
  }
  run(src, dst, std::min(n, share));
  for (auto &t : pool) t.join();
}

// Float32 evaluation of caller-owned (usually pageable) float64 host positions into a device array of polarizabilities.
// A float32 evaluation casts the positions to float32 before any arithmetic (_gnn.py:709), so they are cast HERE, into
// page-locked staging, and cross PCIe as float32: half the bytes, no pageable copy, and results bit-identical to a float64
// upload.  The batch goes through in pieces (below: whole work chunks): this thread casts piece k + 1 while piece k crosses
// PCIe on the copy stream and the kernels of piece k - 1 run on the handle's streams behind h->exec_stream.
static void staged_forward(rn_potgnn *h, const double *positions, int64_t S, double *d_alpha, bool sync) {
  const size_t per_frame = (size_t)h->cfg.num_atoms * 3;
  const int64_t chunk = std::max<int64_t>(1, h->chunk);
  if (!h->exec_stream) HIP_TRY(hipStreamCreateWithFlags(&h->exec_stream, hipStreamNonBlocking));
  if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  // Pieces are whole work chunks: a batch that fits one chunk goes through in ONE piece.  Measured on config 3's 1250-frame
  // share of an 8-GPU run (profiles/r06/host_boundary.txt): with the positions crossing as float32 from page-locked memory the
  // cast + copy of the whole block is 0.3 ms of 20, and every extra launch of the persistent kernels costs more in tails than
  // overlapping it saves -- one piece 0.994 of the resident rate, two 0.97, three 0.95, four equal ones 0.93 (round 5, a pageable
  // float64 copy in front of the kernels: 0.76).  Longer batches overlap naturally: piece k + 1 is cast and copied under the
  // kernels of piece k.  RN_POTGNN_HOST_PIECE = frames per piece (the bit-identity test forces small ones).
  int64_t fixed_piece = (S + ((S + chunk - 1) / chunk) - 1) / ((S + chunk - 1) / chunk);  // (equal pieces: forward_device's rule)
  if (const char *e = getenv("RN_POTGNN_HOST_PIECE")) fixed_piece = std::max<int64_t>(1, std::min<int64_t>(chunk, atoll(e)));
  std::vector<int64_t> pieces;
  for (int64_t left = S; left > 0; left -= pieces.back()) pieces.push_back(std::min<int64_t>(fixed_piece, left));
  const int64_t piece = *std::max_element(pieces.begin(), pieces.end());
  auto &hs = h->hstage;
  // A previous call through the to-device entry returned with its copies and kernels still enqueued: its staging buffers
  // must have left the host before they are overwritten, and its kernels must have read their float32 positions before
  // this call's copies land in the same device buffer.
  for (int b = 0; b < 2; ++b)
    if (hs.copied[b]) HIP_TRY(hipEventSynchronize(hs.copied[b]));
  if (hs.done) HIP_TRY(hipStreamWaitEvent(h->copy_stream, hs.done, 0));
  if (hs.elems < (size_t)piece * per_frame) {
    for (int b = 0; b < 2; ++b) {
      if (hs.pin[b]) (void)hipHostFree(hs.pin[b]);
      hs.pin[b] = nullptr;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&hs.pin[b]), (size_t)piece * per_frame * sizeof(float), hipHostMallocDefault));
      if (!hs.copied[b]) HIP_TRY(hipEventCreateWithFlags(&hs.copied[b], hipEventDisableTiming));
    }
    hs.elems = (size_t)piece * per_frame;
  }
  h->io_pos.ensure((size_t)S * per_frame * sizeof(float));
  float *d_pos32 = h->io_pos.as<float>();
  int b = 0;
  int64_t first = 0;
  static const bool timing = getenv("RN_POTGNN_HOST_TIMING") && atoi(getenv("RN_POTGNN_HOST_TIMING")) != 0;
  auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  double t_cast = 0, t_wait = 0;
  for (size_t k = 0; k < pieces.size(); first += pieces[k], ++k, b ^= 1) {
    const int64_t n = pieces[k];
    const double t0 = now();
    if (k >= 2) HIP_TRY(hipEventSynchronize(hs.copied[b]));  // this buffer's previous copy has left it
    const double t1 = now();
    cast_to_float(positions + first * per_frame, hs.pin[b], (size_t)n * per_frame);
    t_wait += t1 - t0;
    t_cast += now() - t1;
    HIP_TRY(hipMemcpyAsync(d_pos32 + first * per_frame, hs.pin[b], (size_t)n * per_frame * sizeof(float), hipMemcpyHostToDevice,
                           h->copy_stream));
    HIP_TRY(hipEventRecord(hs.copied[b], h->copy_stream));
    HIP_TRY(hipStreamWaitEvent(h->exec_stream, hs.copied[b], 0));
    forward_device<float>(h, nullptr, n, d_alpha + first * 9, nullptr, nullptr, h->exec_stream, sync && first + n >= S,
                          nullptr, nullptr, d_pos32 + first * per_frame);
  }
  if (!hs.done) HIP_TRY(hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(hs.done, h->exec_stream));  // (everything this call enqueued, on every lane, precedes it)
  if (timing)
    fprintf(stderr, "[host timing] S=%lld pieces=%zu total %.0f us: cast %.0f, buffer waits %.0f, rest (enqueue%s) %.0f\n", (long long)S,
            pieces.size(), now() - t_begin, t_cast, t_wait, sync ? " + final sync" : "", now() - t_begin - t_cast - t_wait);
}

int rn_potgnn_forward_device(rn_potgnn *h, const double *d_positions, int64_t S, double *d_alpha,
                             float *d_vec6, void *stream, int synchronize) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && !d_positions)) {
    set_error(h, "invalid positions / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    forward_device<float>(h, d_positions, S, d_alpha, d_vec6, nullptr, (hipStream_t)stream,
                          synchronize != 0);
  });
}

int rn_potgnn_forward_device_f64(rn_potgnn *h, const double *d_positions, int64_t S, double *d_alpha,
                                 void *stream, int synchronize) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!d_positions || !d_alpha))) {
    set_error(h, "invalid positions / alpha / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    sync_host(h);  // the float64 copy of the weights is made from the host master copy
    forward_device<double>(h, d_positions, S, d_alpha, nullptr, nullptr, (hipStream_t)stream, synchronize != 0);
  });
}

int rn_potgnn_calc_polarizabilities(rn_potgnn *h, const double *positions, int64_t S,
                                    double *alpha) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!positions || !alpha))) {
    set_error(h, "invalid positions / alpha / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    const size_t per_frame = (size_t)h->cfg.num_atoms * 3;
    h->io_alpha.ensure((size_t)S * 9 * sizeof(double));
    if (!h->exec_stream) HIP_TRY(hipStreamCreateWithFlags(&h->exec_stream, hipStreamNonBlocking));
    const int64_t chunk = std::max<int64_t>(1, h->chunk);
    static const bool stage_f32 = !(getenv("RN_POTGNN_HOST_F32") && atoi(getenv("RN_POTGNN_HOST_F32")) == 0);
    if (!stage_f32) {
      // (the round-5 form, kept for A/B runs: float64 positions, one blocking copy per work chunk)
      h->io_pos.ensure((size_t)S * per_frame * sizeof(double));
      for (int64_t first = 0; first < S; first += chunk) {
        const int64_t n = std::min<int64_t>(chunk, S - first);
        double *d_pos = h->io_pos.as<double>() + first * per_frame;
        HIP_TRY(hipMemcpy(d_pos, positions + first * per_frame, (size_t)n * per_frame * sizeof(double), hipMemcpyHostToDevice));
        forward_device<float>(h, d_pos, n, h->io_alpha.as<double>() + first * 9, nullptr, nullptr, h->exec_stream, first + n >= S);
      }
      HIP_TRY(hipMemcpy(alpha, h->io_alpha.p, (size_t)S * 9 * sizeof(double), hipMemcpyDeviceToHost));
      return;
    }
    // ONE synchronisation per call: the result and the role-specialised EdgeBlock's time-out word come down into page-locked
    // memory behind the kernels (a blocking hipMemcpy each, after a stream synchronisation of its own, was 30 of the ~450 us
    // of a one-structure call: the reference's unchanged Phonons loop makes 2 M of those, dynamics/_phonon.py:93-106)
    staged_forward(h, positions, S, h->io_alpha.as<double>(), false);
    auto &hs = h->hstage;
    const size_t out_bytes = (size_t)S * 9 * sizeof(double);
    if (hs.out_bytes < out_bytes + 16) {
      if (hs.out_pin) (void)hipHostFree(hs.out_pin);
      hs.out_pin = nullptr;
      hs.out_bytes = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&hs.out_pin), out_bytes + 16, hipHostMallocDefault));
      hs.out_bytes = out_bytes + 16;
    }
    int *fail_pin = reinterpret_cast<int *>(hs.out_pin + out_bytes);
    *fail_pin = 0;
    HIP_TRY(hipMemcpyAsync(hs.out_pin, h->io_alpha.p, out_bytes, hipMemcpyDeviceToHost, h->exec_stream));
    const bool watch = h->use_ps && h->ps_fail.p;
    if (watch) HIP_TRY(hipMemcpyAsync(fail_pin, h->ps_fail.p, sizeof(int), hipMemcpyDeviceToHost, h->exec_stream));
    HIP_TRY(hipStreamSynchronize(h->exec_stream));
    resolve_timers(h);
    if (watch && *fail_pin != 0) check_ps_fail(h);  // (reads the word again, clears it and throws)
    std::memcpy(alpha, hs.out_pin, out_bytes);
  });
}

int rn_potgnn_calc_polarizabilities_to_device(rn_potgnn *h, const double *positions, int64_t S, double *d_alpha, void *stream) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!positions || !d_alpha))) {
    set_error(h, "invalid positions / d_alpha / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    staged_forward(h, positions, S, d_alpha, false);
    // the caller's stream continues behind the evaluation (e.g. the RCCL all-gather of ramannoodle_amd.parallel)
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->hstage.done, 0));
  });
}

int rn_potgnn_calc_polarizabilities_f64(rn_potgnn *h, const double *positions, int64_t S, double *alpha) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!positions || !alpha))) {
    set_error(h, "invalid positions / alpha / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    sync_host(h);  // the float64 copy of the weights is made from the host master copy
    const size_t pb = (size_t)S * h->cfg.num_atoms * 3 * sizeof(double);
    h->io_pos.ensure(pb);
    h->io_alpha.ensure((size_t)S * 9 * sizeof(double));
    HIP_TRY(hipMemcpy(h->io_pos.p, positions, pb, hipMemcpyHostToDevice));
    forward_device<double>(h, h->io_pos.as<double>(), S, h->io_alpha.as<double>(), nullptr, nullptr, nullptr, true);
    HIP_TRY(hipMemcpy(alpha, h->io_alpha.p, (size_t)S * 9 * sizeof(double), hipMemcpyDeviceToHost));
  });
}

int rn_potgnn_calc_polarizabilities_async(rn_potgnn *h, const double *positions, int64_t S, double *alpha) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!positions || !alpha))) {
    set_error(h, "invalid positions / alpha / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->exec_stream) HIP_TRY(hipStreamCreateWithFlags(&h->exec_stream, hipStreamNonBlocking));
    for (auto &sl : h->slots) {  // (the streams may exist already: the synchronous host entry uses them too)
      if (!sl.copied) HIP_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
      if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    auto &sl = h->slots[h->next_slot];
    h->next_slot ^= 1;
    if (sl.busy) HIP_TRY(hipEventSynchronize(sl.done));  // its staging buffers are about to be reused
    const size_t pb = (size_t)S * h->cfg.num_atoms * 3 * sizeof(double);
    sl.pos.ensure(pb);
    sl.alpha.ensure((size_t)S * 9 * sizeof(double));
    HIP_TRY(hipMemcpyAsync(sl.pos.p, positions, pb, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipEventRecord(sl.copied, h->copy_stream));
    HIP_TRY(hipStreamWaitEvent(h->exec_stream, sl.copied, 0));
    forward_device<float>(h, sl.pos.as<double>(), S, sl.alpha.as<double>(), nullptr, nullptr, h->exec_stream,
                          false);
    HIP_TRY(hipMemcpyAsync(alpha, sl.alpha.p, (size_t)S * 9 * sizeof(double), hipMemcpyDeviceToHost,
                           h->exec_stream));
    HIP_TRY(hipEventRecord(sl.done, h->exec_stream));
    sl.busy = true;
  });
}

int rn_potgnn_wait(rn_potgnn *h) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  return guarded(h, [&]() {
    for (auto &sl : h->slots)
      if (sl.busy) {
        HIP_TRY(hipEventSynchronize(sl.done));
        sl.busy = false;
      }
    resolve_timers(h);
    check_ps_fail(h);
  });
}

int rn_host_buffer_alloc(size_t bytes, int device, void **out) {
  if (!out || bytes == 0) return RN_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error(nullptr, "no usable HIP device (count=%d, requested=%d)", ndev, device);
    return RN_ERR_NO_DEVICE;
  }
  return guarded(nullptr, [&]() {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
  });
}

void rn_host_buffer_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int rn_potgnn_forward(rn_potgnn *h, const double *positions, int64_t S, float *vec6) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!positions || !vec6))) {
    set_error(h, "invalid positions / vec6 / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    const size_t pb = (size_t)S * h->cfg.num_atoms * 3 * sizeof(double);
    h->io_pos.ensure(pb);
    h->io_vec6.ensure((size_t)S * 6 * sizeof(float));
    HIP_TRY(hipMemcpy(h->io_pos.p, positions, pb, hipMemcpyHostToDevice));
    forward_device<float>(h, h->io_pos.as<double>(), S, nullptr, h->io_vec6.as<float>(), nullptr,
                          nullptr, true);
    HIP_TRY(hipMemcpy(vec6, h->io_vec6.p, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rn_potgnn_forward_lattices(rn_potgnn *h, const double *lattices, const double *positions, int64_t S,
                               float *vec6) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!lattices || !positions || !vec6))) {
    set_error(h, "invalid lattices / positions / vec6 / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    const size_t pb = (size_t)S * h->cfg.num_atoms * 3 * sizeof(double);
    std::vector<float> lat32((size_t)S * 9);  // the reference's forward computes in float32
    for (size_t i = 0; i < lat32.size(); ++i) lat32[i] = (float)lattices[i];
    h->io_pos.ensure(pb);
    h->io_lat.ensure(lat32.size() * sizeof(float));
    h->io_vec6.ensure((size_t)S * 6 * sizeof(float));
    HIP_TRY(hipMemcpy(h->io_pos.p, positions, pb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->io_lat.p, lat32.data(), lat32.size() * sizeof(float), hipMemcpyHostToDevice));
    forward_device<float>(h, h->io_pos.as<double>(), S, nullptr, h->io_vec6.as<float>(), nullptr,
                          nullptr, true, h->io_lat.as<float>());
    HIP_TRY(hipMemcpy(vec6, h->io_vec6.p, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rn_potgnn_forward_samples(rn_potgnn *h, const double *lattices, const int32_t *atom_types,
                              const double *positions, int64_t S, float *vec6) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!positions || !vec6))) {
    set_error(h, "invalid positions / vec6 / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  const size_t SN = (size_t)S * h->cfg.num_atoms;
  if (atom_types)
    for (size_t i = 0; i < SN; ++i)
      if (atom_types[i] < 0 || atom_types[i] >= h->cfg.num_atom_types) {
        set_error(h, "atom type %d of sample %zu, atom %zu is outside [0,%d)", atom_types[i],
                  i / h->cfg.num_atoms, i % h->cfg.num_atoms, h->cfg.num_atom_types);
        return RN_ERR_INVALID_ARGUMENT;
      }
  return guarded(h, [&]() {
    h->io_pos.ensure(SN * 3 * sizeof(double));
    h->io_vec6.ensure((size_t)S * 6 * sizeof(float));
    HIP_TRY(hipMemcpy(h->io_pos.p, positions, SN * 3 * sizeof(double), hipMemcpyHostToDevice));
    if (lattices) {
      std::vector<float> lat32((size_t)S * 9);  // the reference's forward computes in float32
      for (size_t i = 0; i < lat32.size(); ++i) lat32[i] = (float)lattices[i];
      h->io_lat.ensure(lat32.size() * sizeof(float));
      HIP_TRY(hipMemcpy(h->io_lat.p, lat32.data(), lat32.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (atom_types) {
      h->io_types.ensure(SN * sizeof(int32_t));
      HIP_TRY(hipMemcpy(h->io_types.p, atom_types, SN * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    forward_device<float>(h, h->io_pos.as<double>(), S, nullptr, h->io_vec6.as<float>(), nullptr, nullptr, true,
                          lattices ? h->io_lat.as<float>() : nullptr, atom_types ? h->io_types.as<int>() : nullptr);
    HIP_TRY(hipMemcpy(vec6, h->io_vec6.p, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToHost));
  });
}

int rn_potgnn_forward_samples_f64(rn_potgnn *h, const double *lattices, const int32_t *atom_types,
                                  const double *positions, int64_t S, double *vec6) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!positions || !vec6))) {
    set_error(h, "invalid positions / vec6 / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  const size_t SN = (size_t)S * h->cfg.num_atoms;
  if (atom_types)
    for (size_t i = 0; i < SN; ++i)
      if (atom_types[i] < 0 || atom_types[i] >= h->cfg.num_atom_types) {
        set_error(h, "atom type %d of sample %zu, atom %zu is outside [0,%d)", atom_types[i],
                  i / h->cfg.num_atoms, i % h->cfg.num_atoms, h->cfg.num_atom_types);
        return RN_ERR_INVALID_ARGUMENT;
      }
  return guarded(h, [&]() {
    sync_host(h);  // the float64 copy of the weights is made from the host master copy
    h->io_pos.ensure(SN * 3 * sizeof(double));
    h->io_alpha.ensure((size_t)S * 9 * sizeof(double));
    HIP_TRY(hipMemcpy(h->io_pos.p, positions, SN * 3 * sizeof(double), hipMemcpyHostToDevice));
    if (lattices) {
      h->io_lat.ensure((size_t)S * 9 * sizeof(double));
      HIP_TRY(hipMemcpy(h->io_lat.p, lattices, (size_t)S * 9 * sizeof(double), hipMemcpyHostToDevice));
    }
    if (atom_types) {
      h->io_types.ensure(SN * sizeof(int32_t));
      HIP_TRY(hipMemcpy(h->io_types.p, atom_types, SN * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    // the standardised tensor (what PotGNN.forward returns as a 6-vector) is the reduction's value before alpha * sigma + mu
    forward_device<double>(h, h->io_pos.as<double>(), S, nullptr, nullptr, h->io_alpha.as<double>(), nullptr, true,
                           lattices ? h->io_lat.as<double>() : nullptr, atom_types ? h->io_types.as<int>() : nullptr);
    std::vector<double> raw((size_t)S * 9);
    HIP_TRY(hipMemcpy(raw.data(), h->io_alpha.p, raw.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int pick[6] = {0, 4, 8, 1, 2, 5};  // (xx, yy, zz, xy, xz, yz): dataset/torch/utils.py:44-60
    for (int64_t s = 0; s < S; ++s)
      for (int k = 0; k < 6; ++k) vec6[s * 6 + k] = raw[(size_t)s * 9 + pick[k]];
  });
}

int rn_potgnn_raman_tensors(rn_potgnn *h, const double *ref_positions, const double *displacements,
                            int64_t M, double delta, double *raman) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (M < 0 || delta == 0.0 || !ref_positions || (M > 0 && (!displacements || !raman))) {
    set_error(h, "invalid arguments to raman_tensors");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (M == 0) return RN_OK;
  return guarded(h, [&]() {
    const size_t n3 = (size_t)h->cfg.num_atoms * 3;
    // frames 2m / 2m+1 = ref +/- delta * d_m   (_phonon.py:95-101)
    std::vector<double> pos((size_t)2 * M * n3);
    for (int64_t m = 0; m < M; ++m)
      for (size_t i = 0; i < n3; ++i) {
        const double eps = displacements[m * n3 + i] * delta;
        pos[(2 * m) * n3 + i] = ref_positions[i] + eps;
        pos[(2 * m + 1) * n3 + i] = ref_positions[i] - eps;
      }
    sync_host(h);  // (device-resident training may have moved the weights ahead of the float64 copy)
    h->io_pos.ensure(pos.size() * sizeof(double));
    h->io_alpha.ensure((size_t)2 * M * 9 * sizeof(double));
    HIP_TRY(hipMemcpy(h->io_pos.p, pos.data(), pos.size() * sizeof(double), hipMemcpyHostToDevice));
    forward_device<double>(h, h->io_pos.as<double>(), 2 * M, h->io_alpha.as<double>(), nullptr,
                           nullptr, nullptr, true);
    std::vector<double> a((size_t)2 * M * 9);
    HIP_TRY(hipMemcpy(a.data(), h->io_alpha.p, a.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t m = 0; m < M; ++m)
      for (int i = 0; i < 9; ++i)
        raman[m * 9 + i] = (a[(2 * m) * 9 + i] - a[(2 * m + 1) * 9 + i]) / delta;  // _phonon.py:106
  });
}

int rn_potgnn_alpha_jacobian(rn_potgnn *h, const double *positions, int use_float64, double *jac) {
  if (!h || !positions || !jac) {
    set_error(h, "invalid arguments to alpha_jacobian");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() {
    if (use_float64) {
      sync_host(h);
      jacobian<double>(h, positions, jac);
    } else {
      jacobian<float>(h, positions, jac);
    }
  });
}

int rn_potgnn_raman_tensors_analytic(rn_potgnn *h, const double *ref_positions,
                                     const double *displacements, int64_t M, double *raman) {
  if (!h || M < 0 || !ref_positions || (M > 0 && (!displacements || !raman))) {
    set_error(h, "invalid arguments to raman_tensors_analytic");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (M == 0) return RN_OK;
  return guarded(h, [&]() {
    const size_t n3 = (size_t)h->cfg.num_atoms * 3;
    std::vector<double> jac(6 * n3);
    sync_host(h);
    jacobian<double>(h, ref_positions, jac.data());
    // R_m = 2 * sigma (.) (J d_m): the reference divides its +-delta difference by delta, not
    // 2 delta (dynamics/_phonon.py:106), i.e. it returns twice the directional derivative
    const int map[9] = {0, 3, 4, 3, 1, 5, 4, 5, 2};
    for (int64_t m = 0; m < M; ++m) {
      double v[6] = {0, 0, 0, 0, 0, 0};
      const double *dm = displacements + m * n3;
      for (int k = 0; k < 6; ++k) {
        const double *j = jac.data() + k * n3;
        double acc = 0;
        for (size_t i = 0; i < n3; ++i) acc += j[i] * dm[i];
        v[k] = acc;
      }
      for (int i = 0; i < 9; ++i) raman[m * 9 + i] = 2.0 * h->stdv[i] * v[map[i]];
    }
  });
}

int rn_potgnn_set_weights(rn_potgnn *h, const float *weights, size_t num_weights) {
  if (!h || !weights || num_weights != rn_potgnn_weight_count(&h->cfg)) {
    set_error(h, "invalid arguments to set_weights");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() {
    HIP_TRY(hipDeviceSynchronize());
    pack_weights(h, weights);
    refresh_mfma_mode(h);
    h->host_stale = false;
    h->grads_on_device = false;
    if (h->f32.ready) upload_weights<float>(h);
    if (h->f64.ready) upload_weights<double>(h);
  });
}

int rn_potgnn_train_forward(rn_potgnn *h, const double *positions, int64_t S, float *vec6,
                            float *batch_mean, float *batch_var) {
  if (!h || S <= 0 || !positions || !vec6 || !batch_mean || !batch_var) {
    set_error(h, "invalid arguments to train_forward");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (S > h->chunk) {
    set_error(h, "training batch of %lld frames exceeds max_chunk_structures = %d", (long long)S,
              h->chunk);
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() { train_forward<float>(h, positions, (int)S, vec6, batch_mean, batch_var); });
}

static int check_types(rn_potgnn *h, const int32_t *atom_types, int64_t S) {
  if (!atom_types) return RN_OK;
  const size_t SN = (size_t)S * h->cfg.num_atoms;
  for (size_t i = 0; i < SN; ++i)
    if (atom_types[i] < 0 || atom_types[i] >= h->cfg.num_atom_types) {
      set_error(h, "atom type %d of sample %zu, atom %zu is outside [0,%d)", atom_types[i], i / h->cfg.num_atoms,
                i % h->cfg.num_atoms, h->cfg.num_atom_types);
      return RN_ERR_INVALID_ARGUMENT;
    }
  return RN_OK;
}

int rn_potgnn_train_forward_samples(rn_potgnn *h, const double *lattices, const int32_t *atom_types, const double *positions,
                                    int64_t S, float *vec6, float *batch_mean, float *batch_var) {
  if (!h || S <= 0 || !positions || !vec6 || !batch_mean || !batch_var) {
    set_error(h, "invalid arguments to train_forward_samples");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (S > h->chunk) {
    set_error(h, "training batch of %lld frames exceeds max_chunk_structures = %d", (long long)S, h->chunk);
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (const int rc = check_types(h, atom_types, S); rc != RN_OK) return rc;
  return guarded(h, [&]() { train_forward<float>(h, positions, (int)S, vec6, batch_mean, batch_var, lattices, atom_types); });
}

int rn_potgnn_train_forward_samples_f64(rn_potgnn *h, const double *lattices, const int32_t *atom_types,
                                        const double *positions, int64_t S, double *vec6, double *batch_mean,
                                        double *batch_var) {
  if (!h || S <= 0 || !positions || !vec6 || !batch_mean || !batch_var) {
    set_error(h, "invalid arguments to train_forward_samples_f64");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (S > chunk_frames<double>(h)) {
    set_error(h, "training batch of %lld frames exceeds the float64 chunk of %d frames", (long long)S, chunk_frames<double>(h));
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (const int rc = check_types(h, atom_types, S); rc != RN_OK) return rc;
  return guarded(h, [&]() {
    sync_host(h);
    train_forward<double>(h, positions, (int)S, vec6, batch_mean, batch_var, lattices, atom_types);
  });
}

int rn_potgnn_train_forward_samples_device(rn_potgnn *h, const float *d_lattices, const int32_t *d_atom_types,
                                           const double *d_positions, int64_t S, float *d_vec6, void *stream) {
  if (!h || S <= 0 || !d_positions || !d_vec6) {
    set_error(h, "invalid arguments to train_forward_samples_device");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (!h->device_training) {
    set_error(h, "train_forward_samples_device needs device-resident training (rn_potgnn_set_device_training)");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S > h->chunk) {
    set_error(h, "training batch of %lld frames exceeds max_chunk_structures = %d", (long long)S, h->chunk);
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() { train_forward_device(h, d_positions, (int)S, d_lattices, d_atom_types, d_vec6, (hipStream_t)stream); });
}

int rn_potgnn_train_backward_samples_device(rn_potgnn *h, const float *d_dvec6, void *stream) {
  if (!h || !d_dvec6) {
    set_error(h, "invalid arguments to train_backward_samples_device");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (!h->device_training || h->train_S <= 0 || h->train_prec != 4) {
    set_error(h, "train_backward_samples_device needs device-resident training and a preceding float32 train_forward");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() { train_backward_device(h, d_dvec6, (hipStream_t)stream); });
}

int rn_potgnn_forward_samples_device(rn_potgnn *h, const float *d_lattices, const int32_t *d_atom_types,
                                     const double *d_positions, int64_t S, float *d_vec6, void *stream, int synchronize) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  if (S < 0 || (S > 0 && (!d_positions || !d_vec6))) {
    set_error(h, "invalid positions / vec6 / S");
    return RN_ERR_INVALID_ARGUMENT;
  }
  if (S == 0) return RN_OK;
  return guarded(h, [&]() {
    forward_device<float>(h, d_positions, S, nullptr, d_vec6, nullptr, (hipStream_t)stream, synchronize != 0, d_lattices,
                          d_atom_types);
  });
}

int rn_potgnn_train_forward_f64(rn_potgnn *h, const double *positions, int64_t S, double *vec6,
                                double *batch_mean, double *batch_var) {
  if (!h || S <= 0 || !positions || !vec6 || !batch_mean || !batch_var) {
    set_error(h, "invalid arguments to train_forward_f64");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (S > chunk_frames<double>(h)) {
    set_error(h, "training batch of %lld frames exceeds the float64 chunk of %d frames", (long long)S,
              chunk_frames<double>(h));
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() {
    sync_host(h);
    train_forward<double>(h, positions, (int)S, vec6, batch_mean, batch_var);
  });
}

int rn_potgnn_train_backward_f64(rn_potgnn *h, const double *dvec6, double *grads) {
  if (!h || !dvec6 || !grads) {
    set_error(h, "invalid arguments to train_backward_f64");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (h->train_S <= 0 || h->train_prec != 8) {
    set_error(h, "train_backward_f64 needs a preceding train_forward_f64");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() { train_backward<double>(h, dvec6, grads); });
}

int rn_potgnn_set_device_training(rn_potgnn *h, int enabled) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  h->device_training = enabled != 0;
  return RN_OK;
}

int rn_potgnn_train_backward_device(rn_potgnn *h, const float *dvec6) {
  if (!h || !dvec6) {
    set_error(h, "invalid arguments to train_backward_device");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (h->train_S <= 0 || h->train_prec != 4) {
    set_error(h, "train_backward_device needs a preceding train_forward (an evaluation or Jacobian call "
                 "in between discards its tape)");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() { train_backward<float>(h, dvec6, nullptr); });
}

int rn_potgnn_gradient_buffer(rn_potgnn *h, void **device_ptr, size_t *count) {
  if (!h || !device_ptr || !count || !h->grads_on_device) {
    set_error(h, "no gradients on the device (call rn_potgnn_train_backward_device first)");
    return RN_ERR_INVALID_ARGUMENT;
  }
  *device_ptr = h->f32.grad.p;
  *count = h->lay.total;
  return RN_OK;
}

int rn_potgnn_adam_step(rn_potgnn *h, double lr, double beta1, double beta2, double eps, double weight_decay,
                        int64_t step) {
  if (!h || step < 1 || !(lr >= 0) || !(eps >= 0)) {
    set_error(h, "invalid arguments to adam_step");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (!h->grads_on_device) {
    set_error(h, "adam_step needs the gradients of a preceding train_backward_device");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() {
    Precision<float> &P = h->f32;
    const PackedLayout &L = h->lay;
    const size_t n = L.total;
    hipStream_t st = P.lanes[0].stream;
    if (!h->trainable_mask.p) {  // first step: moments, mask and the table of derived ranges
      h->adam_m.ensure(n * sizeof(float));
      h->adam_v.ensure(n * sizeof(float));
      HIP_TRY(hipMemset(h->adam_m.p, 0, n * sizeof(float)));
      HIP_TRY(hipMemset(h->adam_v.p, 0, n * sizeof(float)));
      const std::vector<unsigned char> mask = trainable_mask(h);
      h->trainable_mask.ensure(mask.size());
      HIP_TRY(hipMemcpy(h->trainable_mask.p, mask.data(), mask.size(), hipMemcpyHostToDevice));
      const std::vector<DerivedOp> ops = derived_ops(h, &h->derived_first_stage);
      h->derived_ops.ensure(ops.size() * sizeof(DerivedOp));
      HIP_TRY(hipMemcpy(h->derived_ops.p, ops.data(), ops.size() * sizeof(DerivedOp), hipMemcpyHostToDevice));
      h->num_derived_ops = (int)ops.size();
    }
    float *w = P.weights.as<float>();
    launch_adam(w, P.grad.as<float>(), h->adam_m.as<float>(), h->adam_v.as<float>(),
                h->trainable_mask.as<unsigned char>(), n, lr, beta1, beta2, eps, weight_decay, step, st,
                (h->use_ps && h->ps_fail.p) ? h->ps_fail.as<int>() : nullptr);
    launch_refresh_derived(w, h->derived_ops.as<DerivedOp>(), h->derived_first_stage, st);
    launch_refresh_derived(w, h->derived_ops.as<DerivedOp>() + h->derived_first_stage, h->num_derived_ops - h->derived_first_stage, st);
    launch_setup<float>(w + L.emb, w + L.W2, w + L.b2, w + L.W4, w + L.b4, h->cfg.num_atom_types, h->d,
                        w + L.node_table, w + L.b0, w + L.bn_w, w + L.bn_b, w + L.bn_rm, w + L.bn_rv,
                        w + L.scale0, w + L.shift0, st);
    HIP_TRY(hipGetLastError());
    // The host looks at three kinds of entries after a step: c3_norm_1's folded constants (the triplet loop's folded-scale
    // variant is chosen from them), the prescale pairs (which double as the finiteness flags of the weight blocks,
    // mfma_f16_range_ok) and the readout block of the split-f16 range guard.  One gather kernel + ONE download into pinned
    // memory (2 P + 1 separate small downloads cost a round trip each: 0.2 ms of a 0.3 ms step).
    {
      std::vector<long long> seg;
      long long total = 0;
      auto add = [&](size_t src, size_t n) {
        seg.push_back((long long)src);
        seg.push_back(total);
        seg.push_back((long long)n);
        total += (long long)n;
      };
      for (const auto &q : L.pass) {
        add(q.c3n1_g, 4 * (size_t)h->d.FeP);  // scale then shift, adjacent in the packed layout
        add(q.mfma_scale, 16);                // + mfma_scale_c
      }
      add(L.W0T, L.b5 + 32 - L.W0T);          // W0T .. b5, contiguous in the layout
      if (h->step_seg.bytes < seg.size() * sizeof(long long)) {
        h->step_seg.ensure(seg.size() * sizeof(long long));
        HIP_TRY(hipMemcpy(h->step_seg.p, seg.data(), seg.size() * sizeof(long long), hipMemcpyHostToDevice));
        h->step_stage.ensure((size_t)total * sizeof(float));
        if (h->step_host) (void)hipHostFree(h->step_host);
        h->step_host = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->step_host), (size_t)total * sizeof(float), hipHostMallocDefault));
      }
      launch_gather_segments(w, h->step_seg.as<long long>(), (int)(seg.size() / 3), h->step_stage.as<float>(), st);
      HIP_TRY(hipMemcpyAsync(h->step_host, h->step_stage.p, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      for (size_t i = 0; i < seg.size(); i += 3)
        std::memcpy(h->packed.data() + seg[i], h->step_host + seg[i + 1], (size_t)seg[i + 2] * sizeof(float));
    }
    HIP_TRY(hipStreamSynchronize(st));
    // The taped forward of this step ran the role-specialised EdgeBlock and nothing has looked at its time-out word since.  If it
    // is set the Adam kernel has applied nothing (it reads the same word), so weights, moments and the host mirror are those of
    // before the step; the gradients are void either way and the handle says so before the error leaves.
    h->grads_on_device = false;
    h->train_S = 0;
    h->host_stale = true;
    refresh_pass_flags<float>(h);
    refresh_mfma_mode(h);
    check_ps_fail(h);
  });
}

int rn_potgnn_get_weights(rn_potgnn *h, float *weights, size_t num_weights) {
  if (!h || !weights || num_weights != rn_potgnn_weight_count(&h->cfg)) {
    set_error(h, "invalid arguments to get_weights");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() {
    sync_host(h);
    unpack_grads<float>(h, h->packed.data(), weights, true);
  });
}

int rn_potgnn_set_stat_reducer(rn_potgnn *h, rn_potgnn_reduce_fn fn, void *ctx) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> hold(h->lock);
  h->reducer = fn;
  h->reducer_ctx = ctx;
  return RN_OK;
}

double rn_potgnn_train_row_count(const rn_potgnn *h) {
  if (!h) return 0.0;
  std::lock_guard<std::recursive_mutex> hold(h->lock);
  return h->bn_count;
}

int rn_potgnn_train_backward(rn_potgnn *h, const float *dvec6, float *grads) {
  if (!h || !dvec6 || !grads) {
    set_error(h, "invalid arguments to train_backward");
    return RN_ERR_INVALID_ARGUMENT;
  }
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (h->train_S <= 0 || h->train_prec != 4) {
    set_error(h, "train_backward needs a preceding train_forward (an evaluation or Jacobian call in "
                 "between discards its tape)");
    return RN_ERR_INVALID_ARGUMENT;
  }
  return guarded(h, [&]() { train_backward<float>(h, dvec6, grads); });
}

int rn_potgnn_radius_graph(const double *lattice, const double *positions, int32_t num_atoms,
                           double cutoff, int device, uint8_t *adjacency) {
  if (!lattice || !positions || !adjacency || num_atoms <= 0 || !(cutoff > 0)) {
    set_error(nullptr, "invalid arguments to radius_graph");
    return RN_ERR_INVALID_ARGUMENT;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error(nullptr, "no usable HIP device (count=%d, requested=%d)", ndev, device);
    return RN_ERR_NO_DEVICE;
  }
  return guarded(nullptr, [&]() {
    HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)num_atoms;
    DeviceBuf lat, pos, adj;
    lat.ensure(9 * sizeof(double));
    pos.ensure(n * 3 * sizeof(double));
    adj.ensure(n * n);
    HIP_TRY(hipMemcpy(lat.p, lattice, 9 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pos.p, positions, n * 3 * sizeof(double), hipMemcpyHostToDevice));
    launch_radius_graph(lat.as<double>(), pos.as<double>(), num_atoms, (float)cutoff,
                        adj.as<unsigned char>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(adjacency, adj.p, n * n, hipMemcpyDeviceToHost));
  });
}

int rn_potgnn_config_flags(const rn_potgnn *h) {
  if (!h) return -1;
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the pass flags are rewritten when weights change)
  int flags = (h->use_fused ? 1 : 0) | ((h->use_fused && h->mfma_f16) ? 4 : 0) | (h->use_narrow ? 8 : 0) |
              ((h->use_fused && h->mfma_range_fallback) ? 16 : 0) | (h->use_edge2 ? 32 : 0) | (h->use_edge3 ? 64 : 0) | (RN_EXPERIMENTS ? 128 : 0);
  {  // bit 8: every pass of a float32 evaluation takes the role-specialised EdgeBlock (kernels_edge_ps.hip)
    bool ps = h->use_fused && h->use_ps && !h->f32.pass.empty();  // (split-f16 or exact-f32 products: the same kernel)
    for (const auto &p : h->f32.pass) ps = ps && (p.c3_fast & 1);
    flags |= ps ? 256 : 0;
  }
  {  // bit 9: float32 evaluations take the atom-owning fused NodeBlock (kernels_node_atom.hip)
    static const bool centred = !(getenv("RN_POTGNN_NODE_CENTRED") && atoi(getenv("RN_POTGNN_NODE_CENTRED")) == 0);
    const bool atom = h->use_fused && h->use_node_fused && h->mfma_f16 && centred && h->g.na_num > 0;
    flags |= atom ? 512 : 0;
    // bit 10: float32 evaluations keep their edge rows as split-f16 pairs (bits 8 and 9 and the fused readout)
    flags |= (atom && (flags & 256) && h->use_readout_fused && h->want_pair_rows && !h->keep_stages) ? 1024 : 0;
  }
  bool fast = !h->f32.pass.empty();
  for (const auto &p : h->f32.pass) fast = fast && !h->use_narrow && (p.c3_fast & (h->use_fused ? 1 : 2));
  return flags | (fast ? 2 : 0);
}

int rn_potgnn_debug_ps_schedule(const int32_t *rb, const int32_t *re, int32_t num_destinations, int32_t back, int32_t ring_tiles,
                                int32_t *window) {
  if (!rb || !re || num_destinations < 0 || back < 1 || back > 8 || ring_tiles < 1 || ring_tiles > 64) return RN_ERR_INVALID_ARGUMENT;
  int w = 0;
  const bool ok = edge_ps_tile_ok(rb, re, num_destinations, back, ring_tiles, &w);
  if (window) *window = w;
  return ok ? 1 : 0;
}

int64_t rn_potgnn_num_triplets(const rn_potgnn *h) { return h ? h->g.T : -1; }

int rn_potgnn_debug_triplets(rn_potgnn *h, int32_t *idx_i, int32_t *idx_j, int32_t *idx_k,
                             int32_t *slot5, int32_t *slot6) {
  if (!h || !idx_i || !idx_j || !idx_k || !slot5 || !slot6) return RN_ERR_INVALID_ARGUMENT;
  return guarded(h, [&]() {
    const size_t T = (size_t)h->g.T;
    if (T == 0) return;
    DeviceBuf buf;
    buf.ensure(5 * T * sizeof(int));
    int *p = buf.as<int>();
    launch_enum_triplets(h->g, p, p + T, p + 2 * T, p + 3 * T, p + 4 * T, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    int32_t *outs[5] = {idx_i, idx_j, idx_k, slot5, slot6};
    for (int k = 0; k < 5; ++k)
      HIP_TRY(hipMemcpy(outs[k], p + k * T, T * sizeof(int), hipMemcpyDeviceToHost));
  });
}

int rn_potgnn_debug_stage(rn_potgnn *h, int stage, int index, float *out, size_t out_capacity,
                          int64_t *rows, int64_t *cols) {
  if (!h || !out || !rows || !cols) return RN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> hold(h->lock);  // (the checks below read handle state)
  if (h->last_was_f64) {
    set_error(h, "debug_stage only reads the float32 path");
    return RN_ERR_UNSUPPORTED;
  }
  const int S = h->last_chunk_structs;
  const int lane_of_last = 0;  // debug use: evaluate <= one chunk so lane 0 holds it
  (void)lane_of_last;
  Precision<float> &P = h->f32;
  return guarded(h, [&]() {
    HIP_TRY(hipDeviceSynchronize());
    const float *src = nullptr;
    int64_t r = 0, c = 0, ld = 0;
    const int64_t ME = (int64_t)S * h->g.E, MN = (int64_t)S * h->g.N;
    if (stage == 0) {
      src = P.lanes[0].unit4.as<float>(); r = ME; c = 4; ld = 4;
    } else if (stage == 1 || stage == 2) {
      if (!h->keep_stages || index < 0 || index > h->cfg.num_message_passes)
        throw HipError{hipErrorInvalidValue, "stage snapshots need RN_POTGNN_KEEP_STAGES=1"};
      if (stage == 1) { src = P.snap_node[index].as<float>(); r = MN; c = h->d.Fn; ld = h->d.FnP; }
      else { src = P.snap_edge[index].as<float>(); r = ME; c = h->d.Fe; ld = h->d.FeP; }
    } else if (stage == 3) {
      src = P.lanes[0].bufA.as<float>(); r = ME; c = 12; ld = 32;
    } else {
      throw HipError{hipErrorInvalidValue, "unknown stage"};
    }
    if ((size_t)(r * c) > out_capacity) throw HipError{hipErrorInvalidValue, "out_capacity too small"};
    HIP_TRY(hipMemcpy2D(out, c * sizeof(float), src, ld * sizeof(float), c * sizeof(float), r,
                        hipMemcpyDeviceToHost));
    if (h->last_in_order && r == ME && stage != 1) {  // per-edge rows of a narrow run: back to edge-id order
      std::vector<float> tmp(out, out + r * c);
      const int E = h->g.E;
      for (int64_t s = 0; s < S; ++s)
        for (int e = 0; e < E; ++e)
          std::memcpy(out + (s * E + e) * c, tmp.data() + (s * E + h->in_pos[e]) * c, (size_t)c * sizeof(float));
    }
    *rows = r;
    *cols = c;
  });
}

int rn_potgnn_set_profiling(rn_potgnn *h, int enabled) {
  if (!h) return RN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> hold(h->lock);
  resolve_timers(h);
  h->profiling = enabled;
  for (int k = 0; k < K_COUNT; ++k) {
    h->k_ms[k] = 0;
    h->k_launches[k] = 0;
  }
  return RN_OK;
}

int rn_potgnn_kernel_times(rn_potgnn *h, const char **names, double *millis, int64_t *launches,
                           int cap) {
  if (!h || !names || !millis || !launches) return RN_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> hold(h->lock);
  (void)hipSetDevice(h->cfg.device);
  resolve_timers(h);
  int n = 0;
  for (int k = 0; k < K_COUNT && n < cap; ++k, ++n) {
    names[n] = kKernelNames[k];
    millis[n] = h->k_ms[k];
    launches[n] = h->k_launches[k];
  }
  return n;
}

}  // extern "C"
