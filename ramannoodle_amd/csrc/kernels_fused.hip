// Fused EdgeBlock for gfx950 (Fn and Fe padded to the same width FP = 64).
//
// One persistent workgroup per (frame, atom tile) work item computes, without any
// intermediate leaving the CU:
//   Q'_e = W5 edge_e + Wi node[b_e]            for the edges e leaving the tile's atoms
//   P'_d = W4 edge_d + Wj node[b_d] + Wk node[a_d] + bias   for the edges d entering them
//   c2_d = c2_linear(node[b_d] * node[a_d])
// by exact-fp32 MFMA (v_mfma_f32_16x16x4_f32; the three weight matrices stay in VGPRs as
// B-fragments for the whole kernel), and then the triplet stage of _EdgeBlock
// (_gnn.py:270-291): add -> LayerNorm(2Fe) -> sigmoid*tanh -> sum over e, LayerNorm(Fe),
// plus c2 (_gnn.py:223-228) and the residual tanh (_gnn.py:351).
//
// Compared with the unfused kernels (kernels_gemm.hip + edge_agg_kernel) this removes the
// [S*E, 4Fe] and [S*E, 2Fe] projection arrays from HBM (4.9x the algorithmic bytes) and
// lets the MFMA pipe of one workgroup overlap the VALU pipe of the other on the same CU.
//
// LayerNorm of x = P' + Q' uses pre-centred rows: with p = P' - mean(P'), q = Q' - mean(Q')
//   x - mean(x) = p + q,   sum (x - mean)^2 = |p|^2 + |q|^2 + 2 p.q
// so each triplet needs ONE 16-lane DPP reduction (p.q) instead of two.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace rn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct EdgeFusedArgs {
  const float *edge_in;
  float *edge_out;
  const float *node;  // updated node embedding [S*N, FP]
  const float *np3;   // [S*N, 6FP] = node * (Wi | Wj(+bias) | Wk)
  int S;
  Graph g;
  Dims d;
  PassW<float> w;
};

constexpr float kLog2e = 1.4426950408889634f;

template <int FP>
__global__ __launch_bounds__(256, 2) void edge_block_fused_kernel(EdgeFusedArgs a) {
  static_assert(FP == 64, "column ownership below assumes 2*FP = 4 waves x 32 columns");
  constexpr int LG = FP / 4;       // lanes per row in the VALU phase (16)
  constexpr int KS = FP / 4;       // k per lane quad (16)
  constexpr int LDQ = 2 * FP + 4;  // LDS row stride (floats)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *bufQ = reinterpret_cast<float *>(smem_raw);       // [maxRows][LDQ] centred source rows
  float *bufP = bufQ + (size_t)a.g.max_tile_out_rows * LDQ;  // [16][LDQ]
  float *bufC = bufP + 16 * LDQ;                           // [16][LDQ]
  float *sq = bufC + 16 * LDQ;                             // [maxRows] |q|^2
  int *qb = reinterpret_cast<int *>(sq + a.g.max_tile_out_rows);  // [maxRows] b_e

  const Graph &g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int colbase = wave * 32;  // this wave's 32 of the 128 pre-activation columns

  // ---- B fragments (weights), resident for the whole kernel
  float bW4[2][KS], bW5[2][KS], bWc[2][KS];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = quad * KS + s, col = colbase + 16 * t + l15;
      bW4[t][s] = a.w.c3_WeT[(size_t)k * (4 * FP) + col];
      bW5[t][s] = a.w.c3_WeT[(size_t)k * (4 * FP) + 2 * FP + col];
      bWc[t][s] = a.w.c2_WT[(size_t)k * (2 * FP) + col];
    }
  float c2bias[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) c2bias[t] = a.w.c2_bias[colbase + 16 * t + l15];

  // ---- VALU-phase constants: lane q4 of a 16-lane group owns columns 4q4..4q4+3 (+FP)
  const int grp = tid / LG, q4 = tid % LG;
  const int nvalid = min(max(a.d.Fe - 4 * q4, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fe), invn = 1.0f / (float)a.d.Fe;
  float g3f[4], b3f[4], g3c[4], b3c[4];  // c3_norm_1 with the exp2 scale folded in
  {
    const Vec4<float> gf = load4<float>(a.w.c3_norm_1.g + 4 * q4);
    const Vec4<float> bf = load4<float>(a.w.c3_norm_1.b + 4 * q4);
    const Vec4<float> gc = load4<float>(a.w.c3_norm_1.g + FP + 4 * q4);
    const Vec4<float> bc = load4<float>(a.w.c3_norm_1.b + FP + 4 * q4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g3f[i] = -kLog2e * gf.v[i];
      b3f[i] = -kLog2e * bf.v[i];
      g3c[i] = 2.0f * kLog2e * gc.v[i];
      b3c[i] = 2.0f * kLog2e * bc.v[i];
    }
  }

  const int64_t items = (int64_t)a.S * g.num_tiles;
  for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int tile = (int)(item % g.num_tiles);
    const int s = (int)(item / g.num_tiles);
    const int j0 = g.tile_begin[tile], j1 = g.tile_begin[tile + 1];
    const int eo0 = g.out_ptr[j0], rows = g.out_ptr[j1] - eo0;
    const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;

    // ================= source rows: Q' = W5 edge_e + Wi node[b_e]  -> LDS
    for (int mt = 0; mt * 16 < rows; ++mt) {
      float af[KS];
      {
        const int r = mt * 16 + l15;
        const float *src = a.edge_in + (erow0 + eo0 + min(r, rows - 1)) * FP + quad * KS;
#pragma unroll
        for (int s4 = 0; s4 < KS; s4 += 4) {
          const float4 v = *reinterpret_cast<const float4 *>(src + s4);
          af[s4] = v.x; af[s4 + 1] = v.y; af[s4 + 2] = v.z; af[s4 + 3] = v.w;
        }
      }
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k], bW5[t][k], acc[t], 0, 0, 0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = mt * 16 + 4 * quad + rr;
        if (r < rows) {
          const int be = g.edge_b[eo0 + r];
          const float *nrow = a.np3 + (nrow0 + be) * (6 * FP) + colbase + l15;
#pragma unroll
          for (int t = 0; t < 2; ++t) bufQ[r * LDQ + colbase + 16 * t + l15] = acc[t][rr] + nrow[16 * t];
        }
      }
    }
    for (int r = tid; r < rows; r += 256) qb[r] = g.edge_b[eo0 + r];
    __syncthreads();
    // centre the source rows and record |q|^2 (padded columns forced to 0)
    for (int r = grp; r < rows; r += 256 / LG) {
      float *row = bufQ + r * LDQ;
      Vec4<float> f = load4<float>(row + 4 * q4), c = load4<float>(row + FP + 4 * q4);
      float sum = (f.v[0] + f.v[1]) + (f.v[2] + f.v[3]) + (c.v[0] + c.v[1]) + (c.v[2] + c.v[3]);
      const float mean = lg_sum<LG>(sum) * inv2n;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f.v[i] = (i < nvalid) ? f.v[i] - mean : 0.f;
        c.v[i] = (i < nvalid) ? c.v[i] - mean : 0.f;
        ss += f.v[i] * f.v[i] + c.v[i] * c.v[i];
      }
      ss = lg_sum<LG>(ss);
      store4(row + 4 * q4, f);
      store4(row + FP + 4 * q4, c);
      if (q4 == 0) sq[r] = ss;
    }
    __syncthreads();

    // ================= destination edges, 16 at a time (one per lane group)
    for (int mt = 0; mt * 16 < dcount; ++mt) {
      // ---- MFMA: P' and c2 pre-activations for 16 destination edges -> LDS
      {
        const int ia = min(mt * 16 + l15, dcount - 1);  // row this lane feeds as A operand
        const int dA = g.in_edge[di0 + ia];
        float af[KS];
        const float *src = a.edge_in + (erow0 + dA) * FP + quad * KS;
#pragma unroll
        for (int s4 = 0; s4 < KS; s4 += 4) {
          const float4 v = *reinterpret_cast<const float4 *>(src + s4);
          af[s4] = v.x; af[s4 + 1] = v.y; af[s4 + 2] = v.z; af[s4 + 3] = v.w;
        }
        f32x4 accP[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            accP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k], bW4[t][k], accP[t], 0, 0, 0);
        // c2 operand: node[b_d] * node[a_d]
        const float *nb = a.node + (nrow0 + g.edge_b[dA]) * FP + quad * KS;
        const float *na = a.node + (nrow0 + g.edge_a[dA]) * FP + quad * KS;
#pragma unroll
        for (int s4 = 0; s4 < KS; s4 += 4) {
          const float4 x = *reinterpret_cast<const float4 *>(nb + s4);
          const float4 y = *reinterpret_cast<const float4 *>(na + s4);
          af[s4] = x.x * y.x; af[s4 + 1] = x.y * y.y; af[s4 + 2] = x.z * y.z; af[s4 + 3] = x.w * y.w;
        }
        f32x4 accC[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            accC[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k], bWc[t][k], accC[t], 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int i = 4 * quad + rr;  // row of the 16x16 output tile held in register rr
          if (mt * 16 + i < dcount) {
            const int dd = g.in_edge[di0 + mt * 16 + i];
            const float *nj = a.np3 + (nrow0 + g.edge_b[dd]) * (6 * FP) + 2 * FP + colbase + l15;
            const float *nk = a.np3 + (nrow0 + g.edge_a[dd]) * (6 * FP) + 4 * FP + colbase + l15;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              bufP[i * LDQ + colbase + 16 * t + l15] = accP[t][rr] + nj[16 * t] + nk[16 * t];
              bufC[i * LDQ + colbase + 16 * t + l15] = accC[t][rr] + c2bias[t];
            }
          }
        }
      }
      __syncthreads();
      // ---- VALU: lane group `grp` owns destination edge mt*16 + grp
      if (mt * 16 + grp < dcount) {
        const int dst = g.in_edge[di0 + mt * 16 + grp];
        const int bd = g.edge_b[dst];
        const int64_t drow = erow0 + dst;
        Vec4<float> pf = load4<float>(bufP + grp * LDQ + 4 * q4);
        Vec4<float> pc = load4<float>(bufP + grp * LDQ + FP + 4 * q4);
        float sp;
        {
          float sum = (pf.v[0] + pf.v[1]) + (pf.v[2] + pf.v[3]) + (pc.v[0] + pc.v[1]) + (pc.v[2] + pc.v[3]);
          const float mean = lg_sum<LG>(sum) * inv2n;
          float ss = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            pf.v[i] = (i < nvalid) ? pf.v[i] - mean : 0.f;
            pc.v[i] = (i < nvalid) ? pc.v[i] - mean : 0.f;
            ss += pf.v[i] * pf.v[i] + pc.v[i] * pc.v[i];
          }
          sp = lg_sum<LG>(ss);
        }
        const int rb = g.out_ptr[bd] - eo0, re = g.out_ptr[bd + 1] - eo0;
        const int rev = g.rev_edge[dst];                 // edge (b_d -> a_d) or -1
        const int rskip = rev >= 0 ? rev - eo0 : re;     // triplets with i == k are excluded
        const int cnt = (re - rb) - (rev >= 0 ? 1 : 0);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < cnt; ++t) {
          const int r = rb + t + ((rb + t >= rskip) ? 1 : 0);
          const float *qr = bufQ + r * LDQ + 4 * q4;
          const Vec4<float> qf = load4<float>(qr);
          const Vec4<float> qc = load4<float>(qr + FP);
          float dot = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) dot += pf.v[i] * qf.v[i] + pc.v[i] * qc.v[i];
          dot = lg_sum<LG>(dot);
          const float var = fmaxf((sp + sq[r] + 2.0f * dot) * inv2n, 0.0f);
          const float rstd = fast_rsq(var + 1e-5f);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float yf = ((pf.v[i] + qf.v[i]) * rstd) * g3f[i] + b3f[i];   // -log2e * LN(filter)
            float yc = ((pc.v[i] + qc.v[i]) * rstd) * g3c[i] + b3c[i];         // 2 log2e * LN(core)
            yc = fminf(fmaxf(yc, -43.28f), 43.28f);
            const float e1 = fast_exp2(yf), e2 = fast_exp2(yc);
            acc[i] += (e2 - 1.0f) * fast_rcp((1.0f + e1) * (1.0f + e2));
          }
        }
        const LnParams<float> p3n{load4<float>(a.w.c3_norm_2.g + 4 * q4),
                                  load4<float>(a.w.c3_norm_2.b + 4 * q4)};
        const Vec4<float> c3 = ln_row<LG, true>(Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}}, p3n,
                                                 invn, nvalid);
        // c2 (_gnn.py:223-228)
        const LnParams<float> p2f{load4<float>(a.w.c2_norm_1.g + 4 * q4),
                                  load4<float>(a.w.c2_norm_1.b + 4 * q4)};
        const LnParams<float> p2c{load4<float>(a.w.c2_norm_1.g + FP + 4 * q4),
                                  load4<float>(a.w.c2_norm_1.b + FP + 4 * q4)};
        const Vec4<float> c2f = load4<float>(bufC + grp * LDQ + 4 * q4);
        const Vec4<float> c2c = load4<float>(bufC + grp * LDQ + FP + 4 * q4);
        const Vec4<float> g2 = ln_gate<LG, true>(c2f, c2c, p2f, p2c, inv2n, nvalid);
        const LnParams<float> p2n{load4<float>(a.w.c2_norm_2.g + 4 * q4),
                                  load4<float>(a.w.c2_norm_2.b + 4 * q4)};
        const Vec4<float> c2 = ln_row<LG, true>(g2, p2n, invn, nvalid);
        const Vec4<float> old = load4<float>(a.edge_in + drow * FP + 4 * q4);
        Vec4<float> out;
#pragma unroll
        for (int i = 0; i < 4; ++i) out.v[i] = acc_tanh(old.v[i] + c2.v[i] + c3.v[i]);
        store4(a.edge_out + drow * FP + 4 * q4, out);
      }
      __syncthreads();  // bufP / bufC are rewritten by the next 16 destinations
    }
  }
}

size_t edge_fused_lds_bytes(const Graph &g, int FP) {
  const size_t ldq = 2 * (size_t)FP + 4;
  return ((size_t)g.max_tile_out_rows * ldq + 2 * 16 * ldq + g.max_tile_out_rows) * sizeof(float) +
         (size_t)g.max_tile_out_rows * sizeof(int);
}

bool edge_fused_supported(const Graph &g, Dims d) {
  return d.FnP == 64 && d.FeP == 64 && edge_fused_lds_bytes(g, 64) <= 80 * 1024;
}

void launch_edge_fused(const float *edge_in, float *edge_out, const float *node, const float *np3,
                       int S, const Graph &g, Dims d, const PassW<float> &w, hipStream_t st) {
  if (S == 0) return;
  EdgeFusedArgs a{edge_in, edge_out, node, np3, S, g, d, w};
  const size_t lds = edge_fused_lds_bytes(g, 64);
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&edge_block_fused_kernel<64>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int64_t items = (int64_t)S * g.num_tiles;
  const unsigned grid = (unsigned)(items < 512 ? items : 512);  // 2 persistent workgroups per CU
  edge_block_fused_kernel<64><<<grid, 256, lds, st>>>(a);
}

}  // namespace rn
