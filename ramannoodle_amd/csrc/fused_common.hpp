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
  // the same with the two 16-column tiles at col0 and col1 (not necessarily adjacent)
  __device__ __forceinline__ void load2(const float *W, int ld, int col0, int col1, int l15, int quad, float prescale) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = prescale * W[(size_t)(quad * KS + 8 * s + j) * ld + (t ? col1 : col0) + l15];
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

// ---- LDS access in inline assembly (kernels that keep LDS-DMA requests in flight across their rounds) ----------
// With an LDS-DMA outstanding hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every LDS access it can see and
// cannot prove disjoint from the DMA's destination, which would turn requests issued rounds ahead into synchronous
// loads.  LDS operations of a wave are performed in order; the waits below are on lgkmcnt only.
__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}
// hides a lane constant from loop-invariant code motion: what is derived from the result is recomputed where it is
// used instead of occupying a register for the whole kernel (the producers hold 96 VGPRs of weights)
__device__ __forceinline__ int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// LDS reads of a PRODUCER, in inline assembly for the same reason: with an LDS-DMA outstanding hipcc puts
// s_waitcnt vmcnt(0) in front of every LDS read it can see (inside a loop it cannot prove that the read misses the
// DMA's destination), which would turn the one-step-ahead requests into synchronous loads.  Reads and their
// lgkmcnt wait form ONE statement, so no use of a result can be scheduled in between.
__device__ __forceinline__ void lds_read4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, f32x4 &r0, f32x4 &r1,
                                          f32x4 &r2, f32x4 &r3) {
  asm volatile(
      "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
      : "memory");
}
__device__ __forceinline__ void lds_read2(unsigned a0, unsigned a1, f32x4 &r0, f32x4 &r1) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(r0), "=&v"(r1)
               : "v"(a0), "v"(a1)
               : "memory");
}
__device__ __forceinline__ void lds_write4(unsigned a0, float4 x) {  // (a write into an operand tile: same reason)
  const f32x4 v = {x.x, x.y, x.z, x.w};
  asm volatile("ds_write_b128 %0, %1" ::"v"(a0), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read1x3(unsigned a0, unsigned a1, unsigned a2, int &r0, int &r1, int &r2) {
  asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2)
               : "v"(a0), "v"(a1), "v"(a2)
               : "memory");
}
__device__ __forceinline__ void lds_read3(unsigned a0, unsigned a1, unsigned a2, f32x4 &r0, f32x4 &r1, f32x4 &r2) {
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %5\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2)
               : "v"(a0), "v"(a1), "v"(a2)
               : "memory");
}
__device__ __forceinline__ int lds_read1(unsigned a0) {
  int r;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a0) : "memory");
  return r;
}
// A fragments of one operand tile whose slots were converted by split_slot (fused_common.hpp: load_split_a)
__device__ __forceinline__ void ps_load_split_a(unsigned tile_addr, int l15, int quad, f16x8 (&ah)[2], f16x8 (&al)[2]) {
  union { f32x4 v; struct { f16x4 hi, lo; } h; } u[4];
  const unsigned row = tile_addr + (unsigned)l15 * (FP * 4);
  lds_read4(row + (((4 * quad + 0) ^ l15) & 15) * 16, row + (((4 * quad + 1) ^ l15) & 15) * 16,
            row + (((4 * quad + 2) ^ l15) & 15) * 16, row + (((4 * quad + 3) ^ l15) & 15) * 16, u[0].v, u[1].v, u[2].v, u[3].v);
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    ah[s2] = __builtin_shufflevector(u[2 * s2].h.hi, u[2 * s2 + 1].h.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    al[s2] = __builtin_shufflevector(u[2 * s2].h.lo, u[2 * s2 + 1].h.lo, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
// the same plus two more 16-byte reads (a bias / scale pair) under the one wait
__device__ __forceinline__ void ps_load_split_a2(unsigned tile_addr, int l15, int quad, f16x8 (&ah)[2], f16x8 (&al)[2],
                                                 unsigned x0, unsigned x1, f32x4 &e0, f32x4 &e1) {
  union { f32x4 v; struct { f16x4 hi, lo; } h; } u[4];
  const unsigned row = tile_addr + (unsigned)l15 * (FP * 4);
  asm volatile(
      "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
      "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(u[0].v), "=&v"(u[1].v), "=&v"(u[2].v), "=&v"(u[3].v), "=&v"(e0), "=&v"(e1)
      : "v"(row + (((4 * quad + 0) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 1) ^ l15) & 15) * 16),
        "v"(row + (((4 * quad + 2) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 3) ^ l15) & 15) * 16), "v"(x0), "v"(x1)
      : "memory");
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    ah[s2] = __builtin_shufflevector(u[2 * s2].h.hi, u[2 * s2 + 1].h.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    al[s2] = __builtin_shufflevector(u[2 * s2].h.lo, u[2 * s2 + 1].h.lo, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}
// v[lane] + v[lane ^ 16] / v[lane] + v[lane ^ 32] on the VALU (v_permlane16_swap / v_permlane32_swap, new on gfx950): the
// swap of a register with itself leaves (row 0, row 0, row 2, row 2) and (row 1, row 1, row 3, row 3) -- resp. the two
// halves -- in the result pair.  ds_swizzle / ds_bpermute do the same through the LDS crossbar at ~100 cycles of latency
// per step, which is what the consumers' epilogue (five dependent reductions) and the producers' row norms were waiting on.
__device__ __forceinline__ float sum_xor16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over an aligned run of 32 lanes
__device__ __forceinline__ float lg_sum32(float v) { return sum_xor16(lg_sum<16>(v)); }

// ---- "pair" operand layout: the 16-byte slots 2m, 2m+1 of a 64-float row (k = 8m .. 8m+7) become [hi x8] and [lo x8], so an
// MFMA fragment of a K = 32 slice is ONE 16-byte read per half with no register shuffling afterwards (the [hi x4 | lo x4]
// slots of split_slot cost six v_mov per slice to reassemble).  Slot J of row r sits at physical position J ^ r, as before.
// The lane that fetched physical slot `phys` of row `row` converts it: both lanes of a pair belong to the row's 16 lanes of
// one wave, which read (and wait) before either writes.
// hi / lo halves of four floats -> the lane's 8-byte shares of the pair's two slots
__device__ __forceinline__ void write_pair(unsigned tile_addr, int row, int phys, const f32x4 &u) {
  const int j = (phys ^ row) & 15;
  const unsigned base = tile_addr + (unsigned)row * (FP * 4);
  f16x4 hi, lo;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (_Float16)u[i];
    lo[i] = (_Float16)(u[i] - (float)hi[i]);
  }
  const unsigned half = (unsigned)(j & 1) * 8u;
  asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3" ::"v"(base + (unsigned)(((j & ~1) ^ row) & 15) * 16u + half),
               "v"(base + (unsigned)(((j | 1) ^ row) & 15) * 16u + half), "v"(hi), "v"(lo)
               : "memory");
}
__device__ __forceinline__ void split_own_pair(unsigned tile_addr, int row, int phys) {
  f32x4 u;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(u)
               : "v"(tile_addr + (unsigned)row * (FP * 4) + (unsigned)phys * 16u)
               : "memory");
  write_pair(tile_addr, row, phys, u);
}
__device__ __forceinline__ void load_pair_a(unsigned tile_addr, int l15, int quad, f16x8 (&ah)[2], f16x8 (&al)[2]) {
  const unsigned row = tile_addr + (unsigned)l15 * (FP * 4);
  asm volatile(
      "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(ah[0]), "=&v"(al[0]), "=&v"(ah[1]), "=&v"(al[1])
      : "v"(row + (((4 * quad + 0) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 1) ^ l15) & 15) * 16),
        "v"(row + (((4 * quad + 2) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 3) ^ l15) & 15) * 16)
      : "memory");
}
// the same from four precomputed lane addresses of ring slot 0 plus a compile-time byte offset (the slot): no address
// arithmetic at all in a loop unrolled over the ring
template <int OFF>
__device__ __forceinline__ void load_pair_a_at(const unsigned (&a4)[4], f16x8 (&ah)[2], f16x8 (&al)[2]) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\t"
      "ds_read_b128 %3, %7 offset:%8\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(ah[0]), "=&v"(al[0]), "=&v"(ah[1]), "=&v"(al[1])
      : "v"(a4[0]), "v"(a4[1]), "v"(a4[2]), "v"(a4[3]), "n"(OFF)
      : "memory");
}
// the same plus two more 16-byte reads (a bias / scale pair) under the one wait
__device__ __forceinline__ void load_pair_a2(unsigned tile_addr, int l15, int quad, f16x8 (&ah)[2], f16x8 (&al)[2], unsigned x0,
                                             unsigned x1, f32x4 &e0, f32x4 &e1) {
  const unsigned row = tile_addr + (unsigned)l15 * (FP * 4);
  asm volatile(
      "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
      "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(ah[0]), "=&v"(al[0]), "=&v"(ah[1]), "=&v"(al[1]), "=&v"(e0), "=&v"(e1)
      : "v"(row + (((4 * quad + 0) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 1) ^ l15) & 15) * 16),
        "v"(row + (((4 * quad + 2) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 3) ^ l15) & 15) * 16), "v"(x0), "v"(x1)
      : "memory");
}
// exact-f32 products: this lane's sixteen consecutive k = 16 quad .. + 15 of row l15 of a float32 operand tile as the LDS-DMA
// left it (slot J of row r at physical position J ^ r), plus two more 16-byte reads (a bias / scale pair) under the one wait
__device__ __forceinline__ void load_f32_a2(unsigned tile_addr, int l15, int quad, float (&af)[16], unsigned x0, unsigned x1,
                                            f32x4 &e0, f32x4 &e1) {
  const unsigned row = tile_addr + (unsigned)l15 * (FP * 4);
  f32x4 u[4];
  asm volatile(
      "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
      "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(e0), "=&v"(e1)
      : "v"(row + (((4 * quad + 0) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 1) ^ l15) & 15) * 16),
        "v"(row + (((4 * quad + 2) ^ l15) & 15) * 16), "v"(row + (((4 * quad + 3) ^ l15) & 15) * 16), "v"(x0), "v"(x1)
      : "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) af[4 * j + i] = u[j][i];
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
