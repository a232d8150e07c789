// Dense per-row projections  Y[M, NOUT] = epi(X[M, KP] * WT[KP, NOUT])  for gfx950.
//
// These are the factorised pieces of the reference's concatenated Linear layers
// (_gnn.py:142 c1_linear, :224 c2_linear, :280 c3_linear) and the readout MLP
// (_gnn.py:532-539).  M is huge (frames x edges), K and NOUT are small (<= 128 / <= 768):
// the weights live in registers as MFMA B-fragments for the whole kernel, X tiles are
// staged through LDS, and the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 does
// the arithmetic (result == an fp32 fma chain, so parity with the fp32 reference holds).
#include <cstdlib>

#include "device_utils.hpp"
#include "kernels.hpp"

namespace rn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float *X;
  int ldx;     // row stride of X (>= KP)
  int accum;   // Y += instead of Y =
  int64_t M;
  const float *WT;
  int NOUT;
  float *Y;
  const float *scale, *shift;
  const float *node;  // amode 1
  const int *edge_a, *edge_b;
  int N, E;
};

// KP: padded K (16..128).  A workgroup (4 waves) owns WN*TN*32 output columns and walks
// 128-row tiles; wave (wm, wn) computes 32-row sub-tiles wm, wm+WM, .. for its TN column
// tiles.  k is split between the two lane halves of the 32x32x2 instruction as
// k = h*KP/2 + step so that every lane reads one contiguous run of its X row from LDS.
// EXT = true: X has row stride a.ldx and the result may be accumulated into Y (reverse-pass
// products); EXT = false keeps the inference instantiations free of both.
template <int KP, int WN, int TN, int AMODE, int EPI, bool EXT = false>
__global__ __launch_bounds__(256) void rowgemm_mfma_kernel(GemmArgs a) {
  constexpr int WM = 4 / WN;
  constexpr int BM = 128;
  constexpr int LDA = KP + 4;  // +4 floats: conflict-free ds_read_b128 across rows
  constexpr int KH = KP / 2;
  constexpr int C4 = KP / 4;
  __shared__ __attribute__((aligned(16))) float xs[BM * LDA];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int h = lane >> 5, l31 = lane & 31;
  const int ncol0 = blockIdx.y * (WN * TN * 32) + wn * (TN * 32);

  float bfrag[TN][KH];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int s = 0; s < KH; ++s)
      bfrag[t][s] = a.WT[(int64_t)(h * KH + s) * a.NOUT + ncol0 + t * 32 + l31];

  float sc[TN], sh[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    sc[t] = (EPI == 2) ? a.scale[ncol0 + t * 32 + l31] : 1.0f;
    sh[t] = (EPI >= 1) ? a.shift[ncol0 + t * 32 + l31] : 0.0f;
  }

  // X rows of the NEXT tile are fetched into registers while the current tile is in the
  // MFMA/store phase (issue-early / write-late staging), so global latency is off the
  // critical path.  NPF float4 per thread; for amode 1 both gathered node rows are held.
  constexpr int NPF = (BM * C4) / 256;
#ifndef RN_GEMM_EXT_PREFETCH
#define RN_GEMM_EXT_PREFETCH 0
#endif
  constexpr bool PREFETCH = (KP <= 64) || (RN_GEMM_EXT_PREFETCH && EXT && TN == 1);  // (KP = 128: 64 more registers, only beside one column tile)
  float4 pre[NPF], pre2[AMODE == 1 ? NPF : 1];
  auto fetch = [&](int64_t row0) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int i = tid + j * 256;
      const int r = i / C4, c = (i % C4) * 4;
      const int64_t row = row0 + r;
      pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (AMODE == 1) pre2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < a.M) {
        if (AMODE == 0) {
          pre[j] = *reinterpret_cast<const float4 *>(a.X + row * (EXT ? a.ldx : KP) + c);
        } else {
          const int64_t s = row / a.E;
          const int e = (int)(row % a.E);
          pre[j] = *reinterpret_cast<const float4 *>(a.node + (s * a.N + a.edge_b[e]) * KP + c);
          pre2[j] = *reinterpret_cast<const float4 *>(a.node + (s * a.N + a.edge_a[e]) * KP + c);
        }
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int i = tid + j * 256;
      const int r = i / C4, c = (i % C4) * 4;
      float4 v = pre[j];
      if (AMODE == 1)
        v = make_float4(v.x * pre2[j].x, v.y * pre2[j].y, v.z * pre2[j].z, v.w * pre2[j].w);
      *reinterpret_cast<float4 *>(xs + r * LDA + c) = v;
    }
  };

  const int64_t num_tiles = (a.M + BM - 1) / BM;
  if (PREFETCH && (int64_t)blockIdx.x < num_tiles) fetch((int64_t)blockIdx.x * BM);
  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * BM;
    if (!PREFETCH) fetch(row0);
    commit();
    __syncthreads();
    if (PREFETCH && tile + gridDim.x < num_tiles) fetch((tile + gridDim.x) * BM);
#pragma unroll 1
    for (int mt = wm; mt < BM / 32; mt += WM) {
      float afrag[KH];
      const float *ap = xs + (mt * 32 + l31) * LDA + h * KH;
#pragma unroll
      for (int s = 0; s < KH; s += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(ap + s);
        afrag[s] = v.x;
        afrag[s + 1] = v.y;
        afrag[s + 2] = v.z;
        afrag[s + 3] = v.w;
      }
      f32x16 acc[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      if (EXT && a.accum) {  // Y += : the old values seed the accumulators, their loads overlap the products
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.M) acc[t][r] = a.Y[row * a.NOUT + ncol0 + t * 32 + l31];
          }
      }
#pragma unroll
      for (int s = 0; s < KH; ++s)
#pragma unroll
        for (int t = 0; t < TN; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[s], bfrag[t][s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int col = ncol0 + t * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < a.M) {
            float v = acc[t][r];
            if (EPI == 1) v += sh[t];
            if (EPI == 2) v = ssp_fast(v * sc[t] + sh[t]);
            a.Y[row * a.NOUT + col] = v;
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------- K = 64 streams
// Wave-autonomous variant for the three big edge projections of a pass (c1, c3 = P'|Q', c2;
// K = 64, NOUT a multiple of 128).  The weight slice of a workgroup (128 columns) sits in LDS
// transposed; every wave streams its own 32-row tiles straight from HBM into A fragments
// (each lane one contiguous 128-byte run, next tile's run fetched while the current one
// multiplies) and never meets a barrier after the prologue.  160 VGPRs and 34 KiB of LDS
// per workgroup, so one such workgroup fits on a CU next to two workgroups of
// edge_agg_kernel where the register-resident kernel above (196-276 registers per lane)
// is shut out.  Opt-in (RN_POTGNN_GEMM_STREAM=1): measured on MI355X it is 8-30 % slower
// than the kernel above in isolation and the better co-residency does not make up for it
// (profiles/r01/overlap_experiments.txt).
template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void rowgemm_stream64_kernel(GemmArgs a, int tiles_per_wave) {
  constexpr int KP = 64, NS = 128, LDW = KP + 4;
  __shared__ __attribute__((aligned(16))) float wt[NS * LDW];  // wt[n][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int ncol0 = blockIdx.y * NS;
  for (int i = tid; i < KP * NS; i += 256) {
    const int k = i / NS, n = i % NS;
    wt[n * LDW + k] = a.WT[(int64_t)k * a.NOUT + ncol0 + n];
  }
  __syncthreads();

  const int64_t num_tiles = (a.M + 31) / 32;
  const int64_t first = ((int64_t)blockIdx.x * 4 + wave) * tiles_per_wave;
  // this lane's run of the operand row(s) of tile t: columns 32h .. 32h+31 of row 32t + l31
  float4 nxt[8], nxt2[AMODE == 1 ? 8 : 1];
  auto fetch = [&](int64_t tile) {
    int64_t row = tile * 32 + l31;
    if (row >= a.M) row = a.M - 1;
    if (AMODE == 0) {
      const float *p = a.X + row * KP + 32 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) nxt[j] = *reinterpret_cast<const float4 *>(p + 4 * j);
    } else {
      const int64_t s = row / a.E;
      const int e = (int)(row - s * a.E);
      const float *pb = a.node + (s * a.N + a.edge_b[e]) * KP + 32 * h;
      const float *pa = a.node + (s * a.N + a.edge_a[e]) * KP + 32 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        nxt[j] = *reinterpret_cast<const float4 *>(pb + 4 * j);
        nxt2[j] = *reinterpret_cast<const float4 *>(pa + 4 * j);
      }
    }
  };
  if (first < num_tiles) fetch(first);
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int64_t tile = first + it;
    if (tile >= num_tiles) break;
    float afrag[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float4 v = nxt[j];
      if (AMODE == 1) v = make_float4(v.x * nxt2[j].x, v.y * nxt2[j].y, v.z * nxt2[j].z, v.w * nxt2[j].w);
      afrag[4 * j] = v.x; afrag[4 * j + 1] = v.y; afrag[4 * j + 2] = v.z; afrag[4 * j + 3] = v.w;
    }
    if (it + 1 < tiles_per_wave && tile + 1 < num_tiles) fetch(tile + 1);
    const int64_t row0 = tile * 32;
#pragma unroll 1
    for (int nt = 0; nt < NS / 32; ++nt) {
      const float *wp = wt + (nt * 32 + l31) * LDW + 32 * h;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 b = *reinterpret_cast<const float4 *>(wp + 4 * j);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[4 * j], b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[4 * j + 1], b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[4 * j + 2], b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[4 * j + 3], b.w, acc, 0, 0, 0);
      }
      const int col = ncol0 + nt * 32 + l31;
      const float sc = (EPI == 2) ? a.scale[col] : 1.0f;
      const float sh = (EPI >= 1) ? a.shift[col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < a.M) {
          float v = acc[r];
          if (EPI == 1) v += sh;
          if (EPI == 2) v = ssp_fast(v * sc + sh);
          a.Y[row * a.NOUT + col] = v;
        }
      }
    }
  }
}

static bool launch_stream64(const GemmArgs &a, int amode, int epi, hipStream_t st) {
  static const int enabled = getenv("RN_POTGNN_GEMM_STREAM") ? atoi(getenv("RN_POTGNN_GEMM_STREAM")) : 0;
  if (!enabled || a.NOUT % 128 != 0 || a.M < 4096) return false;
  const int64_t tiles = (a.M + 31) / 32;
  // a few tiles per wave: amortises the weight prologue, keeps the tail short
  static const int tpw_env = getenv("RN_POTGNN_GEMM_TPW") ? atoi(getenv("RN_POTGNN_GEMM_TPW")) : 4;
  const int tpw = tpw_env < 1 ? 1 : tpw_env;
  const dim3 grid((unsigned)((tiles + 4 * tpw - 1) / (4 * tpw)), (unsigned)(a.NOUT / 128));
#define RN_ST(AM, EP) rowgemm_stream64_kernel<AM, EP><<<grid, 256, 0, st>>>(a, tpw)
  if (amode == 0) {
    if (epi == 0) RN_ST(0, 0);
    else if (epi == 1) RN_ST(0, 1);
    else RN_ST(0, 2);
  } else {
    if (epi == 0) RN_ST(1, 0);
    else RN_ST(1, 1);
  }
#undef RN_ST
  return true;
}

template <int KP, int WN, int TN>
static void launch_mfma_cfg(const GemmArgs &a, int amode, int epi, dim3 grid, hipStream_t st) {
#define RN_GEMM(AM, EP) rowgemm_mfma_kernel<KP, WN, TN, AM, EP><<<grid, 256, 0, st>>>(a)
  if (amode == 0) {
    if (epi == 0) RN_GEMM(0, 0);
    else if (epi == 1) RN_GEMM(0, 1);
    else RN_GEMM(0, 2);
  } else {
    if (epi == 0) RN_GEMM(1, 0);
    else RN_GEMM(1, 1);
  }
#undef RN_GEMM
}

template <int KP>
static void launch_mfma_ext(const GemmArgs &a, hipStream_t st) {  // plain product, EXT features
  int slice = 32;
  if (a.NOUT % 64 == 0) slice = 64;
  if (a.NOUT % 128 == 0) slice = 128;
  const int64_t tiles = (a.M + 127) / 128;
  dim3 grid((unsigned)(tiles < 2048 ? tiles : 2048), (unsigned)(a.NOUT / slice));
  switch (slice) {
    case 32: rowgemm_mfma_kernel<KP, 1, 1, 0, 0, true><<<grid, 256, 0, st>>>(a); break;
    case 64: rowgemm_mfma_kernel<KP, 2, 1, 0, 0, true><<<grid, 256, 0, st>>>(a); break;
    case 128: rowgemm_mfma_kernel<KP, 4, 1, 0, 0, true><<<grid, 256, 0, st>>>(a); break;
  }
}

template <int KP>
static void launch_mfma_kp(const GemmArgs &a, int amode, int epi, hipStream_t st) {
  // column slice per workgroup: the largest of 256/128/64/32 that divides NOUT and keeps
  // the B fragments within 64 VGPRs (TN * KP/2 <= 64).
  int slice = 32;
  if (a.NOUT % 64 == 0) slice = 64;
  if (a.NOUT % 128 == 0) slice = 128;
  static const int max_slice = getenv("RN_POTGNN_GEMM_MAXSLICE") ? atoi(getenv("RN_POTGNN_GEMM_MAXSLICE")) : 256;
  if (a.NOUT % 256 == 0 && KP <= 64 && max_slice >= 256) slice = 256;
  const int64_t tiles = (a.M + 127) / 128;
  dim3 grid((unsigned)(tiles < 2048 ? tiles : 2048), (unsigned)(a.NOUT / slice));
  switch (slice) {
    case 32: launch_mfma_cfg<KP, 1, 1>(a, amode, epi, grid, st); break;
    case 64: launch_mfma_cfg<KP, 2, 1>(a, amode, epi, grid, st); break;
    case 128: launch_mfma_cfg<KP, 4, 1>(a, amode, epi, grid, st); break;
    case 256:
      if constexpr (KP <= 64) launch_mfma_cfg<KP, 4, 2>(a, amode, epi, grid, st);
      break;
  }
}

template <>
void launch_rowgemm<float>(const float *X, int64_t M, int KP, const float *WT, int NOUT, float *Y,
                           const float *scale, const float *shift, bool act, int amode,
                           const float *node, const Graph &g, hipStream_t st) {
  if (M == 0) return;
  GemmArgs a{X, KP, 0, M, WT, NOUT, Y, scale, shift, node, g.edge_a, g.edge_b, g.N, g.E};
  const int epi = act ? 2 : (shift ? 1 : 0);
  switch (KP) {
    case 16: launch_mfma_kp<16>(a, amode, epi, st); break;
    case 32: launch_mfma_kp<32>(a, amode, epi, st); break;
    case 64:
      if (!launch_stream64(a, amode, epi, st)) launch_mfma_kp<64>(a, amode, epi, st);
      break;
    case 128: launch_mfma_kp<128>(a, amode, epi, st); break;
  }
}

// Y[M, NOUT] (+)= X[M, 0:N] * Wt[N, NOUT] for any N that is a multiple of 16: the inner
// dimension is walked in blocks of <= 128 columns (row stride ldx), accumulating into Y.
// Used for the activation gradients dX = dY * W^T with the pre-transposed weights.
bool launch_rowgemm_blocks(const float *X, int ldx, int N, int64_t M, const float *Wt, int NOUT,
                           float *Y, bool accumulate, const Graph &g, hipStream_t st) {
  if (M == 0) return true;
  if (NOUT % 32 != 0 || N % 16 != 0) return false;
  int bk = 128;
  while (N % bk != 0) bk /= 2;
  for (int off = 0; off < N; off += bk) {
    GemmArgs a{X + off, ldx, (accumulate || off > 0) ? 1 : 0, M, Wt + (size_t)off * NOUT, NOUT, Y,
               nullptr, nullptr, nullptr, g.edge_a, g.edge_b, g.N, g.E};
    switch (bk) {
      case 16: launch_mfma_ext<16>(a, st); break;
      case 32: launch_mfma_ext<32>(a, st); break;
      case 64: launch_mfma_ext<64>(a, st); break;
      case 128: launch_mfma_ext<128>(a, st); break;
    }
  }
  return true;
}

// ---------------------------------------------------------------------------- split-f16 row products
// Y[M, NOUT] (+)= X[M, 0:64 KB] * Wt[64 KB, NOUT] (+ bias) on the f16 matrix pipe as three products per term
// (x = hi + lo in f16: device_utils.hpp mfma_split3), for the reverse pass and its recomputations, where the
// exact-f32 instruction of rowgemm_mfma_kernel is what bounds the products (2.4 GFLOP at 40 % of the 157 TFLOP/s
// f32 MFMA peak = 35-60 us, against 14-19 us of HBM time).  Wave-autonomous: a wave owns 16 NTW output columns,
// keeps their weights as split fragments in registers (64 KB NTW / 4 VGPRs); the workgroup's four waves share the
// 16-row operand tile: each wave loads four whole rows (next tile requested before the products of the current one),
// scales and splits them ONCE and leaves MFMA-ready fragments in a double-buffered LDS tile (one barrier per tile);
// a lane stores four consecutive columns of a row.
// Ranges: cotangent rows can be 1e-9 small and weights arbitrary, so every ROW of X and every wave's 64 x 16 NTW
// weight block are brought to [2^12, 2^13) by an exact power of two before the split (kernels.hpp:
// mfma_prescale) and the accumulators are scaled back in float32.
// AMODE 1: the operand row of edge (frame s, e) is node[s, b_e] * node[s, a_e] (the c2 operand; node rows K wide).
typedef _Float16 gf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gf16x4 __attribute__((ext_vector_type(4)));
template <int KB, int NTW, int AMODE>
__global__ __launch_bounds__(256, 2) void rowgemm_split_kernel(GemmArgs a) {
  // A tile of 16 rows as MFMA-ready fragments: [buffer][kb][slice s][hi | lo][row * 4 + quad] x 16 bytes
  __shared__ __attribute__((aligned(16))) gf16x8 atile[2][KB][2][2][64];
  __shared__ float s_inv[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int col0 = blockIdx.y * (64 * NTW) + wave * (16 * NTW);

  gf16x8 wh[KB][NTW][2], wl[KB][NTW][2];
  float winv[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float tmp[NTW][2][8];
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          tmp[t][s2][j] = a.WT[(int64_t)(64 * kb + 16 * quad + 8 * s2 + j) * a.NOUT + col0 + 16 * t + l15];
          m = fmaxf(m, fabsf(tmp[t][s2][j]));
        }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) m = fmaxf(m, __shfl_xor(m, sft));
    const float sw = mfma_prescale(m);
    winv[kb] = 1.0f / sw;  // (a power of two: exact)
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[t][s2][j] *= sw;
        split_f16x8(tmp[t][s2], wh[kb][t][s2], wl[kb][t][s2]);
      }
  }

  // Loading: wave w brings rows 4w .. 4w+3 of a tile whole (a row is 16 KB float4 pieces, LPR lanes wide), so a
  // row's largest magnitude is a reduction over neighbouring lanes of ONE wave.
  constexpr int LPR = 16 * KB;  // lanes per row (16, 32 or 64)
  const int64_t ntiles = (a.M + 15) / 16;
  float4 nxt[KB];
  auto fetch = [&](int64_t tile) {
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const int idx = j * 64 + lane;
      int64_t row = tile * 16 + 4 * wave + idx / LPR;
      if (row >= a.M) row = a.M - 1;
      const int p = idx % LPR;
      if (AMODE == 0) {
        nxt[j] = *reinterpret_cast<const float4 *>(a.X + row * a.ldx + 4 * p);
      } else {
        const int64_t s = row / a.E;
        const int e = (int)(row - s * a.E);
        const float4 x = *reinterpret_cast<const float4 *>(a.node + (s * a.N + a.edge_b[e]) * (64 * KB) + 4 * p);
        const float4 y = *reinterpret_cast<const float4 *>(a.node + (s * a.N + a.edge_a[e]) * (64 * KB) + 4 * p);
        nxt[j] = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
      }
    }
  };
  // registers -> scaled split fragments in buffer b
  auto stage = [&](int b) {
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const int idx = j * 64 + lane;
      const int rowl = 4 * wave + idx / LPR, p = idx % LPR;
      float4 v = nxt[j];
      float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
      for (int sft = 1; sft < LPR; sft <<= 1) m = fmaxf(m, __shfl_xor(m, sft));
      const float sr = mfma_prescale(m);
      if (p == 0) s_inv[b][rowl] = 1.0f / sr;
      const float x[4] = {v.x * sr, v.y * sr, v.z * sr, v.w * sr};
      gf16x4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (_Float16)x[k];
        lo[k] = (_Float16)(x[k] - (float)hi[k]);
      }
      const int kb = p >> 4, qd = (p >> 2) & 3, s2 = (p >> 1) & 1, half = p & 1;
      gf16x4 *dh = reinterpret_cast<gf16x4 *>(&atile[b][kb][s2][0][rowl * 4 + qd]) + half;
      gf16x4 *dl = reinterpret_cast<gf16x4 *>(&atile[b][kb][s2][1][rowl * 4 + qd]) + half;
      *dh = hi;
      *dl = lo;
    }
  };
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);
  int b = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, b ^= 1) {
    stage(b);
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
    const int64_t row = tile * 16 + l15;
    const bool live = row < a.M;
    float4 old[NTW];
    if (a.accum && live) {  // Y += : requested before the products
#pragma unroll
      for (int t = 0; t < NTW; ++t) old[t] = *reinterpret_cast<const float4 *>(a.Y + row * a.NOUT + col0 + 16 * t + 4 * quad);
    }
    __syncthreads();  // buffer b complete (the other buffer is free: every wave has passed the previous barrier's products)
    const float inv_sr = s_inv[b][l15];
    f32x4_t tot[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) tot[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const gf16x8 ah0 = atile[b][kb][0][0][l15 * 4 + quad], al0 = atile[b][kb][0][1][l15 * 4 + quad];
      const gf16x8 ah1 = atile[b][kb][1][0][l15 * 4 + quad], al1 = atile[b][kb][1][1][l15 * 4 + quad];
      const float scale = winv[kb] * inv_sr;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma_split3(wh[kb][t][0], wl[kb][t][0], ah0, al0, acc);
        acc = mfma_split3(wh[kb][t][1], wl[kb][t][1], ah1, al1, acc);
        tot[t] += acc * scale;
      }
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int col = col0 + 16 * t + 4 * quad;
        float4 v = make_float4(tot[t][0], tot[t][1], tot[t][2], tot[t][3]);
        if (a.shift) {
          const float4 bb = *reinterpret_cast<const float4 *>(a.shift + col);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (a.accum) {
          v.x += old[t].x; v.y += old[t].y; v.z += old[t].z; v.w += old[t].w;
        }
        *reinterpret_cast<float4 *>(a.Y + row * a.NOUT + col) = v;
      }
    }
  }
}

// true = launched.  K a multiple of 64 up to 256, NOUT a multiple of 64 (amode 1: K = 64).
bool launch_rowgemm_split(const float *X, int ldx, int K, int64_t M, const float *Wt, int NOUT, float *Y,
                          bool accumulate, const float *bias, int amode, const float *node, const Graph &g,
                          hipStream_t st) {
  if (M == 0) return true;
  static const bool off = getenv("RN_POTGNN_BWD_SPLIT_GEMM") && atoi(getenv("RN_POTGNN_BWD_SPLIT_GEMM")) == 0;
  if (off || K % 64 != 0 || K > 256 || K == 192 || NOUT % 64 != 0 || (amode == 1 && K > 128)) return false;
  GemmArgs a{X, ldx, accumulate ? 1 : 0, M, Wt, NOUT, Y, nullptr, bias, node, g.edge_a, g.edge_b, g.N, g.E};
  const int kb = K / 64;
  // columns per workgroup 64 NTW: as many as the weights' registers allow (64 KB NTW / 4 <= 64)
  int ntw = 1;
  if (kb == 1 && NOUT % 256 == 0) ntw = 4;
  else if (kb <= 2 && NOUT % 128 == 0) ntw = 2;
  const int64_t ntiles = (M + 15) / 16;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  // (resident workgroups per CU follow from the registers: 80-130 VGPRs for the small fragment sets, 228 for K = 256)
  static const int wgs_per_cu = getenv("RN_POTGNN_SPLIT_GEMM_WGS") ? std::max(1, atoi(getenv("RN_POTGNN_SPLIT_GEMM_WGS"))) : 4;
  const dim3 grid((unsigned)std::min<int64_t>(ntiles, (int64_t)wgs_per_cu * cus), (unsigned)(NOUT / (64 * ntw)));
#define RN_SPLIT(KBV, NTWV)                                                             \
  do {                                                                                  \
    if (amode == 1) rowgemm_split_kernel<1, NTWV, 1><<<grid, 256, 0, st>>>(a);          \
    else rowgemm_split_kernel<KBV, NTWV, 0><<<grid, 256, 0, st>>>(a);                   \
  } while (0)
  if (kb == 1 && ntw == 4) RN_SPLIT(1, 4);
  else if (kb == 1 && ntw == 2) RN_SPLIT(1, 2);
  else if (kb == 1) RN_SPLIT(1, 1);
  else if (kb == 2 && ntw == 2) {
    if (amode == 1) rowgemm_split_kernel<2, 2, 1><<<grid, 256, 0, st>>>(a);
    else rowgemm_split_kernel<2, 2, 0><<<grid, 256, 0, st>>>(a);
  } else if (kb == 2) {
    if (amode == 1) rowgemm_split_kernel<2, 1, 1><<<grid, 256, 0, st>>>(a);
    else rowgemm_split_kernel<2, 1, 0><<<grid, 256, 0, st>>>(a);
  }
  else { if (amode == 1) return false; rowgemm_split_kernel<4, 1, 0><<<grid, 256, 0, st>>>(a); }
#undef RN_SPLIT
  return true;
}

// ---------------------------------------------------------------------------- float64
// Double-precision variant used by the finite-difference Raman-tensor path
// (dynamics/_phonon.py:93-106), where fp32 cancellation would dominate.  Plain FMA
// kernel: 32-row X tile in LDS, one output column per thread.
__global__ __launch_bounds__(256) void rowgemm_f64_kernel(const double *X, int64_t M, int KP,
                                                          const double *WT, int NOUT, double *Y,
                                                          const double *scale, const double *shift,
                                                          int epi, int amode, const double *node,
                                                          Graph g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double *xs = reinterpret_cast<double *>(smem_raw);  // [32][KP]
  const int64_t num_tiles = (M + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * 32;
    for (int i = threadIdx.x; i < 32 * KP; i += 256) {
      const int r = i / KP, c = i % KP;
      const int64_t row = row0 + r;
      double v = 0;
      if (row < M) {
        if (amode == 0) {
          v = X[row * KP + c];
        } else {
          const int64_t s = row / g.E;
          const int e = (int)(row % g.E);
          v = node[(s * g.N + g.edge_b[e]) * KP + c] * node[(s * g.N + g.edge_a[e]) * KP + c];
        }
      }
      xs[i] = v;
    }
    __syncthreads();
    for (int col = threadIdx.x; col < NOUT; col += 256) {
      double acc[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) acc[r] = 0;
      for (int k = 0; k < KP; ++k) {
        const double wv = WT[(int64_t)k * NOUT + col];
#pragma unroll
        for (int r = 0; r < 32; ++r) acc[r] += xs[r * KP + k] * wv;
      }
      const double sc = (epi == 2) ? scale[col] : 1.0;
      const double sh = (epi >= 1) ? shift[col] : 0.0;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int64_t row = row0 + r;
        if (row < M) {
          double v = acc[r];
          if (epi == 1) v += sh;
          if (epi == 2) v = ssp(v * sc + sh);
          Y[row * NOUT + col] = v;
        }
      }
    }
    __syncthreads();
  }
}

// float64 on the matrix pipe: v_mfma_f64_16x16x4_f64 (78 TFLOP/s against the ~10 the FMA kernel above reaches).
// Wave-autonomous like rowgemm_split_kernel's first form: a wave owns 16 NTW output columns with their weights in
// registers (lane (l15, quad): W[KQ quad + ks][col], ks < KQ = K / 4), streams 16-row tiles from global memory into
// the row operand (one contiguous run of KQ doubles per lane, next tile requested before the products) and ends
// with columns quad, 4 + quad, 8 + quad, 12 + quad of row l15 per lane and tile (weights as the instruction's first operand).
typedef double f64x4_t __attribute__((ext_vector_type(4)));
template <int KQ, int NTW>
__global__ __launch_bounds__(256) void rowgemm_f64_mfma_kernel(const double *__restrict__ X, int64_t M,
                                                               const double *__restrict__ WT, int NOUT,
                                                               double *__restrict__ Y, const double *__restrict__ scale,
                                                               const double *__restrict__ shift, int epi, int amode,
                                                               const double *__restrict__ node, Graph g) {
  constexpr int K = 4 * KQ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int col0 = blockIdx.y * (64 * NTW) + wave * (16 * NTW);
  if (col0 >= NOUT) return;  // (NOUT = 32: the last workgroup's upper waves have no columns; no barriers in this kernel)
  double w[NTW][KQ];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) w[t][ks] = WT[(int64_t)(KQ * quad + ks) * NOUT + col0 + 16 * t + l15];
  const int64_t ntiles = (M + 15) / 16;
  double nxt[KQ];
  auto fetch = [&](int64_t tile) {
    int64_t row = tile * 16 + l15;
    if (row >= M) row = M - 1;
    if (amode == 0) {
      const double *p = X + row * K + KQ * quad;
#pragma unroll
      for (int ks = 0; ks < KQ; ks += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(p + ks);
        nxt[ks] = v.x;
        nxt[ks + 1] = v.y;
      }
    } else {
      const int64_t s = row / g.E;
      const int e = (int)(row - s * g.E);
      const double *pb = node + (s * g.N + g.edge_b[e]) * K + KQ * quad, *pa = node + (s * g.N + g.edge_a[e]) * K + KQ * quad;
#pragma unroll
      for (int ks = 0; ks < KQ; ks += 2) {
        const double2 x = *reinterpret_cast<const double2 *>(pb + ks), y = *reinterpret_cast<const double2 *>(pa + ks);
        nxt[ks] = x.x * y.x;
        nxt[ks + 1] = x.y * y.y;
      }
    }
  };
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    double x[KQ];
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) x[ks] = nxt[ks];
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
    f64x4_t acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f64x4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[t][ks], x[ks], acc[t], 0, 0, 0);
    const int64_t row = tile * 16 + l15;
    if (row < M) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        // (the f64 instruction's result layout differs from the f32 one: register j of lane (l15, quad) is row
        //  4 j + quad of the product, i.e. here output column 4 j + quad -- tools/mfma_f64_layout_probe.hip)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = col0 + 16 * t + 4 * j + quad;
          double v = acc[t][j];
          if (epi == 1) v += shift[col];
          if (epi == 2) v = ssp(v * scale[col] + shift[col]);
          Y[row * NOUT + col] = v;
        }
      }
    }
  }
}

template <int KQ>
static bool launch_f64_mfma(const double *X, int64_t M, const double *WT, int NOUT, double *Y, const double *scale,
                            const double *shift, int epi, int amode, const double *node, const Graph &g, hipStream_t st) {
  // column tiles per wave: as many as 128 VGPRs of weights allow (2 KQ per tile) and NOUT divides into
  int ntw = 0;
  for (int c : {4, 2, 1})
    if (2 * KQ * c <= 128 && NOUT % (64 * c) == 0) {
      ntw = c;
      break;
    }
  if (ntw == 0 && NOUT % 16 == 0) ntw = 1;  // e.g. the readout's last layer (32 columns): two waves of a workgroup idle
  if (ntw == 0) return false;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int64_t ntiles = (M + 15) / 16;
  const dim3 grid((unsigned)std::min<int64_t>(ntiles, (int64_t)4 * cus), (unsigned)((NOUT + 64 * ntw - 1) / (64 * ntw)));
  if (ntw == 4) {
    if constexpr (2 * KQ * 4 <= 128) rowgemm_f64_mfma_kernel<KQ, 4><<<grid, 256, 0, st>>>(X, M, WT, NOUT, Y, scale, shift, epi, amode, node, g);
  } else if (ntw == 2) {
    if constexpr (2 * KQ * 2 <= 128) rowgemm_f64_mfma_kernel<KQ, 2><<<grid, 256, 0, st>>>(X, M, WT, NOUT, Y, scale, shift, epi, amode, node, g);
  } else {
    rowgemm_f64_mfma_kernel<KQ, 1><<<grid, 256, 0, st>>>(X, M, WT, NOUT, Y, scale, shift, epi, amode, node, g);
  }
  return true;
}

template <>
void launch_rowgemm<double>(const double *X, int64_t M, int KP, const double *WT, int NOUT,
                            double *Y, const double *scale, const double *shift, bool act,
                            int amode, const double *node, const Graph &g, hipStream_t st) {
  if (M == 0) return;
  const int epi = act ? 2 : (shift ? 1 : 0);
  static const bool no_mfma = getenv("RN_POTGNN_F64_MFMA") && atoi(getenv("RN_POTGNN_F64_MFMA")) == 0;
  if (!no_mfma) {
    bool done = false;
    switch (KP) {
      case 16: done = launch_f64_mfma<4>(X, M, WT, NOUT, Y, scale, shift, epi, amode, node, g, st); break;
      case 32: done = launch_f64_mfma<8>(X, M, WT, NOUT, Y, scale, shift, epi, amode, node, g, st); break;
      case 64: done = launch_f64_mfma<16>(X, M, WT, NOUT, Y, scale, shift, epi, amode, node, g, st); break;
      case 128: done = launch_f64_mfma<32>(X, M, WT, NOUT, Y, scale, shift, epi, amode, node, g, st); break;
    }
    if (done) return;
  }
  const int64_t tiles = (M + 31) / 32;
  const unsigned grid = (unsigned)(tiles < 8192 ? tiles : 8192);
  rowgemm_f64_kernel<<<grid, 256, (size_t)32 * KP * sizeof(double), st>>>(
      X, M, KP, WT, NOUT, Y, scale, shift, epi, amode, node, g);
}

}  // namespace rn
