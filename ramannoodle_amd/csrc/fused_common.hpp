// Definitions shared by the fused kernels (kernels_fused.hip, kernels_edge_ps.hip) and the opt-in experiment
// kernels (experiments/kernels_fused_experiments.hip): tile constants, the resident weight fragments of a wave,
// LDS-DMA and split-f16 operand helpers.  gfx950 / wave64 only.
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

namespace rn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef RN_FUSED_PK
#define RN_FUSED_PK 1  // packed-f32 (two columns per instruction) arithmetic in the triplet loop
#endif
#ifndef RN_FUSED_PRIO
#define RN_FUSED_PRIO 3  // s_setprio level of every phase but the triplet loop (0: off); +2 % (profiles/r02/edge_phase_probe.txt)
#endif
#ifndef RN_FUSED_PRIO_LATE
#define RN_FUSED_PRIO_LATE 0  // 1: the epilogue stays at the loop's priority (measured 0.8 % slower)
#endif
#ifndef RN_NODE_PRIO
#define RN_NODE_PRIO 3  // NodeBlock kernel: priority of the MFMA phase over the gate phase (-2.7 % of its time)
#endif
#ifndef RN_FUSED_NK_SAMEROUND
#define RN_FUSED_NK_SAMEROUND 0
#endif
#ifndef RN_E3_PAIRWISE
#define RN_E3_PAIRWISE 1  // twelve-wave kernel: two triplets per loop iteration
#endif
#ifndef RN_FUSED_PAIRWISE
#define RN_FUSED_PAIRWISE 1  // measured +1.3 % isolated, +3 % with two lanes; 240 VGPRs, no spills
#endif

namespace {
constexpr int FP = 64;
constexpr int LG = 16;            // lanes per row in the VALU phase
constexpr int NG = 16;            // lane groups per workgroup = destinations per round
constexpr int KS = 16;            // k values per lane of an MFMA operand (64 / 4 lane quads)
constexpr int LDQ = 2 * FP + 4;   // LDS row stride of the pre-activation rows (floats)
constexpr float kLog2e = 1.4426950408889634f;

struct FusedLds {
  size_t bufQ, bufP, bufC, atile, sq, nj, lnp, ints, total;
};
__host__ __device__ inline FusedLds fused_lds(int maxR, int maxD, int maxN) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  FusedLds L;
  size_t off = 0;
  L.bufQ = off; off += up((size_t)maxR * LDQ * 4);
  L.bufP = off; off += up((size_t)NG * LDQ * 4);
  L.bufC = off; off += up((size_t)NG * LDQ * 4);
  L.atile = off; off += 3 * (size_t)NG * FP * 4;
  L.sq = off; off += up((size_t)maxR * 4);
  L.nj = off; off += up((size_t)maxN * 2 * FP * 4);
  L.lnp = off; off += (size_t)12 * FP * 4;
  L.ints = off; off += up(((size_t)maxR + 6 * (size_t)maxD) * 4);
  L.total = off;
  return L;
}

// Retire this wave's outstanding LDS-DMA writes.  An s_barrier does not wait for vmcnt, so every
// wave runs this before the barrier that publishes the operand tiles to the other waves (the
// compiler happens to drain vmcnt earlier today; the protocol must not rest on that).
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 16-byte LDS-DMA: lane l's 16 bytes at `src` land at lds_wave_base + 16 l.
__device__ __forceinline__ void dma16(const float *src, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// One wave's share of a [64 x N] weight matrix W (row-major, leading dimension ld): the 32
// columns colbase .. colbase+31 as two 16-column MFMA tiles, resident in VGPRs.  `product`
// accumulates the TRANSPOSED tile  acc[t] += (X W[:, tile t])^T  for a 16-row X tile of which this
// lane (row l15, quad) holds the 16 consecutive k = 16 quad .. 16 quad + 15 in `af`: the weights go
// in as the MFMA's A operand, the rows as its B operand (the two fragment layouts are the same), so
// a lane ends up with FOUR CONSECUTIVE COLUMNS 16 t + 4 quad .. + 3 of row l15 -- one 16-byte LDS
// store per tile instead of four 4-byte ones.
template <bool F16>
struct WaveB;
template <>
struct WaveB<false> {  // exact f32: v_mfma_f32_16x16x4_f32, k = 16 quad + step
  float w[2][KS];
  __device__ __forceinline__ void load(const float *W, int ld, int colbase, int l15, int quad, float /*prescale: exact f32 needs none*/) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < KS; ++k) w[t][k] = W[(size_t)(quad * KS + k) * ld + colbase + 16 * t + l15];
  }
  template <bool K32_unused = false>  // (same call syntax as the split-f16 form)
  __device__ __forceinline__ void product(const float (&af)[KS], f32x4 (&acc)[2]) const {
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][k], af[k], acc[t], 0, 0, 0);
  }
};
template <>
struct WaveB<true> {  // split f16 (device_utils.hpp: mfma_split3), K = 32 slice s covers k = 16 quad + 8 s + j
  f16x8 h[2][2], l[2][2];
  // `prescale`: the block's power-of-two mfma_prescale (kernels.hpp) -- the halves then sit in f16's
  // normal range whatever the scale of the weights; the caller multiplies the accumulator by 1/prescale
  __device__ __forceinline__ void load(const float *W, int ld, int colbase, int l15, int quad, float prescale) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = prescale * W[(size_t)(quad * KS + 8 * s + j) * ld + colbase + 16 * t + l15];
        split_f16x8(tmp, h[t][s], l[t][s]);
      }
  }
  template <bool K32 = (RN_MFMA_K32 != 0)>
  __device__ __forceinline__ void product(const float (&af)[KS], f32x4 (&acc)[2]) const {
    f16x8 ah[2], al[2];
    split_f16x8(af, ah[0], al[0]);
    split_f16x8(af + 8, ah[1], al[1]);
    product_split<K32>(ah, al, acc);
  }
  // the same with an A operand that is already split (slice s = k 16 quad + 8 s .. + 7)
  // (K32 = false: the two-instruction form, device_utils.hpp -- what edge_block_fused_kernel must use)
  template <bool K32 = (RN_MFMA_K32 != 0)>
  __device__ __forceinline__ void product_split(const f16x8 (&ah)[2], const f16x8 (&al)[2], f32x4 (&acc)[2]) const {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = mfma_split3_t<K32>(h[t][s], l[t][s], ah[s], al[s], acc[t]);
  }
};

// One DMA slot (4 consecutive floats of an operand row) -> [hi x4 | lo x4] halves in the same 16 bytes.
union SplitSlot {
  float4 f;
  struct {
    f16x4 hi, lo;
  } h;
};
__device__ __forceinline__ float4 split_slot(float4 x) {
  SplitSlot u;
  const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 hi = (_Float16)v[j];
    u.h.hi[j] = hi;
    u.h.lo[j] = (_Float16)(v[j] - (float)hi);
  }
  return u.f;
}
// A fragments of one operand tile whose slots were converted by split_slot: four 16-byte reads.
__device__ __forceinline__ void load_split_a(const float *tile, int l15, int quad, f16x8 (&ah)[2], f16x8 (&al)[2]) {
  SplitSlot u[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    u[j].f = *reinterpret_cast<const float4 *>(tile + l15 * FP + (((4 * quad + j) ^ l15) & 15) * 4);
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    ah[s2] = __builtin_shufflevector(u[2 * s2].h.hi, u[2 * s2 + 1].h.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    al[s2] = __builtin_shufflevector(u[2 * s2].h.lo, u[2 * s2 + 1].h.lo, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
}  // namespace

// _NodeBlock.forward (_gnn.py:122-151) in one launch: for the edges e entering the tile's
// atoms,  c1 = W_e edge_e (MFMA, operand rows by LDS-DMA one round ahead) + (W_n node[b_e] + bias)
// (the small per-atom projection npc1), LayerNorm(2Fn) -> sigmoid*tanh, summed per atom in
// in-edge order, LayerNorm(Fn), residual tanh.  Replaces the c1 edge projection (0.59 MB in,
// 1.18 MB out per structure and pass) plus node_agg_kernel (which read it back).
struct NodeFusedArgs {
  const float *edge;     // [S*E, FP]
  const float *node_in;  // [S*N, FP]
  const float *npc1;     // [S*N, 2FP] = W_n node + bias
  float *node_out;       // [S*N, FP]
  int S;
  Graph g;
  Dims d;
  PassW<float> w;
};

namespace {
constexpr int LDG = FP + 4;  // row stride of the gated rows in LDS
struct NodeFusedLds {
  size_t bufP, atile, gated, nj, lnp, ints, total;
};
__host__ __device__ inline NodeFusedLds node_fused_lds(int maxD, int maxN) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  NodeFusedLds L;
  size_t off = 0;
  L.bufP = off; off += up((size_t)NG * LDQ * 4);
  L.atile = off; off += (size_t)NG * FP * 4;
  L.gated = off; off += up((size_t)maxD * LDG * 4);
  L.nj = off; off += up((size_t)maxN * 2 * FP * 4);
  L.lnp = off; off += (size_t)6 * FP * 4;
  L.ints = off; off += up((size_t)2 * maxD * 4);
  L.total = off;
  return L;
}
}  // namespace

#if RN_EXPERIMENTS
void launch_node_wave(const NodeFusedArgs &a, hipStream_t st);  // experiments/: wave-autonomous NodeBlock
#endif

}  // namespace rn
