// Fused EdgeBlock for gfx950 (float32, Fn and Fe padded to the same width FP = 64).
//
// One persistent workgroup per (atom tile, frame group) computes, without any projection
// leaving the CU:
//   Q'_e = W5 edge_e + Wi node[b_e]                        edges e leaving the tile's atoms
//   P'_d = W4 edge_d + Wj node[b_d] + Wk node[a_d] + bias  edges d entering them
//   c2_d = c2_linear(node[b_d] * node[a_d])
// by MFMA (W4 and the c2 weight stay in VGPRs as B-fragments for the whole kernel, W5 is re-read
// from L2 once per frame) -- by default as three split-f16 products on v_mfma_f32_16x16x16_f16
// (device_utils.hpp: f32-grade accuracy at 3/16 of the f32 matrix-pipe time, and the f16 pipe
// does not block the VALU as the f32-input MFMA does), or exact-fp32 v_mfma_f32_16x16x4_f32
// (RN_POTGNN_MFMA=f32) -- then the
// triplet stage of _EdgeBlock (_gnn.py:270-291): add -> LayerNorm(2Fe) -> sigmoid*tanh ->
// sum over e -> LayerNorm(Fe), plus c2 (_gnn.py:223-228) and the residual tanh
// (_gnn.py:351).  The node terms (Wi|Wj|Wk) node come from the small per-atom projection
// `np3` as in the unfused path.
//
// Against projections + edge_agg_kernel this removes the [S*E, 4Fe] and [S*E, 2Fe] arrays
// from HBM (3.5 MB written and read again per structure and pass) and lets the matrix
// pipe of one workgroup run under the VALU-bound triplet loop of the other on the same CU.
//
// Structure of one frame (S1/S2 = workgroup barriers):
//   stage:  Q' tile by MFMA -> LDS; centre rows, |q|^2                      (2 barriers)
//   round r (16 destination edges, one per 16-lane group):
//     MFMA   A fragments of the 16 destinations' operands (edge row; node[b] * node[a]) are read
//            from LDS tiles that the PREVIOUS round filled by LDS-DMA (global_load_lds,
//            XOR-swizzled through the per-lane source address) and that the fetching lanes
//            turned into [hi x4 | lo x4] halves in place; the MFMAs take the weights as their
//            A operand, so a lane ends with four consecutive columns of one row and
//            P' and c2 -> LDS as 16-byte stores                                      S1
//     DMA    issue the next round's (or next frame's first round's) operand rows
//     VALU   triplet loop (two-column packed arithmetic) on LDS operands, epilogue, store;
//            then each lane splits the operand slots it fetched itself               S2
// The tile topology lives in LDS for the whole launch (the graph is the same in every
// frame), so no load inside the frame loop has a dependent address.
// (Round 5: moved here from kernels_fused.hip -- retired from the product build, see the note there.)
#include "../ramannoodle_amd/csrc/fused_common.hpp"

namespace rn {

// Timing-only probe build (RN_BUILD_TAG=probe RN_EXTRA_FLAGS=-DRN_FUSED_PROBE=1; wrong results):
// RN_FUSED_PROBE_MASK in the environment switches phases of the EdgeBlock kernel off at run time --
// 1 triplet loop, 2 epilogue (LayerNorms, c2 gate, tanh), 4 Q' MFMA staging, 8 centring pass,
// 16 per-round MFMA phase, 32 operand LDS-DMA.  The product build compiles none of it.
#ifndef RN_FUSED_PROBE
#define RN_FUSED_PROBE 0
#endif
#if RN_FUSED_PROBE
#define RN_PROBE(bit) ((a.probe & (bit)) != 0)
#else
#define RN_PROBE(bit) false
#endif

struct EdgeFusedArgs {
  const float *edge_in;
  float *edge_out;
  const float *node;  // updated node embedding [S*N, FP]
  const float *np3;   // [S*N, 6FP] = node * (Wi | Wj(+bias) | Wk)
  float *agg_out;     // taped runs: the pre-LayerNorm triplet sums [S*E, FP] (what the reverse pass needs); else null
  int S;
  Graph g;
  Dims d;
  PassW<float> w;
#if RN_FUSED_PROBE
  int probe;
#endif
};

template <bool PAD, bool FASTG, bool F16>
__global__ __launch_bounds__(256, 2) void edge_block_fused_kernel(EdgeFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const FusedLds L = fused_lds(g.max_tile_out_rows, g.max_tile_in_rows, g.max_tile_nodes);
  float *bufQ = reinterpret_cast<float *>(smem_raw + L.bufQ);   // [maxR][LDQ] centred source rows
  float *bufP = reinterpret_cast<float *>(smem_raw + L.bufP);   // [16][LDQ] W4 edge_d
  float *bufC = reinterpret_cast<float *>(smem_raw + L.bufC);   // [16][LDQ] c2 pre-activation
  float *atile = reinterpret_cast<float *>(smem_raw + L.atile); // 3 x [16][64] swizzled operand rows
  float *nj = reinterpret_cast<float *>(smem_raw + L.nj);       // [maxN][2FP] Wj node[j] + bias
  float *lnp = reinterpret_cast<float *>(smem_raw + L.lnp);
  float *s_c3n2g = lnp, *s_c3n2b = lnp + FP, *s_c2n1g = lnp + 2 * FP, *s_c2n1b = lnp + 4 * FP,
        *s_c2n2g = lnp + 6 * FP, *s_c2n2b = lnp + 7 * FP, *s_g3 = lnp + 8 * FP, *s_ig3 = lnp + 10 * FP;
  int *qb = reinterpret_cast<int *>(smem_raw + L.ints);
  const int maxD = g.max_tile_in_rows;
  int *d_edge = qb + g.max_tile_out_rows, *d_a = d_edge + maxD, *d_bl = d_a + maxD, *d_rb = d_bl + maxD,
      *d_cnt = d_rb + maxD, *d_skip = d_cnt + maxD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int colbase = wave * 32;  // this wave's 32 of the 128 pre-activation columns
#if RN_FUSED_PRIO
  __builtin_amdgcn_s_setprio(RN_FUSED_PRIO);
#endif

  // Workgroups of one frame group share node / np3 rows: keep them on one XCD (its L2).
  // Dispatch is round-robin over the 8 XCDs, so consecutive logical ids = same XCD.
  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.num_tiles;
  const int sg = logical / g.num_tiles, nsg = gridDim.x / g.num_tiles;
  const int j0 = g.tile_begin[tile], j1 = g.tile_begin[tile + 1];
  const int eo0 = g.out_ptr[j0], rows = g.out_ptr[j1] - eo0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
  const int nrounds = (dcount + NG - 1) / NG;

  // ---- once per launch: LayerNorm parameters and the tile topology -> LDS
  for (int c = tid; c < 2 * FP; c += 256) {
    s_c2n1g[c] = a.w.c2_norm_1.g[c];
    s_c2n1b[c] = a.w.c2_norm_1.b[c];
    const float gam = a.w.c3_norm_1.g[c] * (c < FP ? -kLog2e : 2.0f * kLog2e);
    s_g3[c] = gam;
    s_ig3[c] = ((c % FP) < a.d.Fe) ? 1.0f / gam : 0.0f;
    if (c < FP) {
      s_c3n2g[c] = a.w.c3_norm_2.g[c];
      s_c3n2b[c] = a.w.c3_norm_2.b[c];
      s_c2n2g[c] = a.w.c2_norm_2.g[c];
      s_c2n2b[c] = a.w.c2_norm_2.b[c];
    }
  }
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

  // ---- B fragments resident for the whole kernel: lane (n = l15, quad) holds
  //      W[k = 16 quad ..+15][colbase + 16 t + n]
  // split-f16 products run on power-of-two prescaled weights (kernels.hpp: mfma_prescale); the inverse
  // scales are folded into the first float32 operation on each accumulator
  const float s4 = F16 ? a.w.mfma_scale[2] : 1.0f, inv4 = F16 ? a.w.mfma_scale[3] : 1.0f;
  const float s5 = F16 ? a.w.mfma_scale[4] : 1.0f, inv5 = F16 ? a.w.mfma_scale[5] : 1.0f;
  const float sc2 = F16 ? a.w.mfma_scale[6] : 1.0f, invc2 = F16 ? a.w.mfma_scale[7] : 1.0f;
  WaveB<F16> bW4, bWc;
  bW4.load(a.w.c3_WeT, 4 * FP, colbase, l15, quad, s4);
  bWc.load(a.w.c2_WT, 2 * FP, colbase, l15, quad, sc2);
  const float *c2bias_p = a.w.c2_bias + colbase + 4 * quad;  // (re-read per round: 8 VGPRs the kernel does not have)

  // ---- VALU-phase constants: lane q4 of group grp owns columns 4q4..4q4+3 (+FP)
  const int grp = tid / LG, q4 = tid % LG, c0 = 4 * q4;
  const int nvalid = min(max(a.d.Fe - c0, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fe), invn = 1.0f / (float)a.d.Fe;
  float b3f[4], b3c[4];  // c3_norm_1's shift with the exp2 scale of the gate folded in (its scale: s_g3 in LDS)
  {
    const Vec4<float> bf = load4<float>(a.w.c3_norm_1.b + c0), bc = load4<float>(a.w.c3_norm_1.b + FP + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      b3f[i] = -kLog2e * bf.v[i];
      b3c[i] = 2.0f * kLog2e * bc.v[i];
    }
  }
  __syncthreads();

  // LDS-DMA of the operand rows of round `r` of frame `s`: wave w brings rows 4w..4w+3 of
  // each of the three tiles; slot (row, piece p) receives global piece p ^ row.
  auto prefetch_round = [&](int s, int r) {
    if (RN_PROBE(32)) return;
    const int row = 4 * wave + quad;
    const int i = min(r * NG + row, dcount - 1);
    const int piece = (l15 ^ row) & 15;
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
    float *dst = atile + wave * 256;
    dma16(a.edge_in + (erow0 + d_edge[i]) * FP + 4 * piece, dst);
    dma16(a.node + (nrow0 + j0 + d_bl[i]) * FP + 4 * piece, dst + NG * FP);
    dma16(a.node + (nrow0 + d_a[i]) * FP + 4 * piece, dst + 2 * NG * FP);
  };
  // Split-f16 path: once its own DMA has landed (dma_wait), every lane turns the slots IT fetched
  // into MFMA-ready halves in place -- the edge row, and node[b] * node[a] (the c2 operand) in the
  // node[b] tile -- so the four waves read fragments instead of each splitting the whole tile.
  auto split_landed_tiles = [&]() {
    if constexpr (F16) {
      float *slot = atile + wave * 256 + lane * 4;
      const float4 e = *reinterpret_cast<const float4 *>(slot);
      const float4 x = *reinterpret_cast<const float4 *>(slot + NG * FP);
      const float4 y = *reinterpret_cast<const float4 *>(slot + 2 * NG * FP);
      *reinterpret_cast<float4 *>(slot) = split_slot(e);
      *reinterpret_cast<float4 *>(slot + NG * FP) = split_slot(float4{x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w});
    }
  };
  if (sg < a.S && dcount > 0) prefetch_round(sg, 0);

  for (int s = sg; s < a.S; s += nsg) {
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
    // ---- per-atom part of P': Wj node[j] + bias for the tile's atoms
    for (int i = tid; i < (j1 - j0) * (2 * FP / 4); i += 256) {
      const int n = i / (2 * FP / 4), c = (i % (2 * FP / 4)) * 4;
      store4(nj + (size_t)n * 2 * FP + c, load4<float>(a.np3 + (nrow0 + j0 + n) * (6 * FP) + 2 * FP + c));
    }
    // ================= source rows: W5 edge_e by MFMA -> bufQ (raw)
    // (requesting the rows of the next 16-row tile before this tile's products, or the Wi node[b_e]
    //  terms of the centring pass one row ahead, measured 3 % and 6 % SLOWER in round 2 although
    //  nothing spills at 253 VGPRs: the kernel is issue-bound, not waiting on these loads)
    {
      WaveB<F16> bW5;
      bW5.load(a.w.c3_WeT + 2 * FP, 4 * FP, colbase, l15, quad, s5);
      for (int mt = 0; mt * 16 < (RN_PROBE(4) ? 0 : rows); ++mt) {
        float af[KS];
        const float *src = a.edge_in + (erow0 + eo0 + min(mt * 16 + l15, rows - 1)) * FP + quad * KS;
#pragma unroll
        for (int s4 = 0; s4 < KS; s4 += 4) {
          const float4 v = *reinterpret_cast<const float4 *>(src + s4);
          af[s4] = v.x; af[s4 + 1] = v.y; af[s4 + 2] = v.z; af[s4 + 3] = v.w;
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bW5.template product<false>(af, acc);  // K = 16 MFMAs in this kernel: device_utils.hpp
        if (const int r = mt * 16 + l15; r < rows) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
            *reinterpret_cast<f32x4 *>(bufQ + r * LDQ + colbase + 16 * t + 4 * quad) = acc[t];
        }
      }
    }
    __syncthreads();
    // ---- add Wi node[b_e], centre, record |q|^2 (padded columns forced to 0)
    for (int r = grp; r < (RN_PROBE(8) ? 0 : rows); r += NG) {
      float *row = bufQ + r * LDQ;
      const float *np = a.np3 + (nrow0 + qb[r]) * (6 * FP) + c0;
      Vec4<float> f = load4<float>(row + c0), c = load4<float>(row + FP + c0);
      const Vec4<float> nf = load4<float>(np), nc = load4<float>(np + FP);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f.v[i] = fmaf(f.v[i], inv5, nf.v[i]);
        c.v[i] = fmaf(c.v[i], inv5, nc.v[i]);
        sum += f.v[i] + c.v[i];
      }
      const float mean = lg_sum<LG>(sum) * inv2n;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f.v[i] = (!PAD || i < nvalid) ? f.v[i] - mean : 0.f;
        c.v[i] = (!PAD || i < nvalid) ? c.v[i] - mean : 0.f;
        ss += f.v[i] * f.v[i] + c.v[i] * c.v[i];
      }
      ss = lg_sum<LG>(ss);
      if (FASTG) {
        const Vec4<float> g3f = load4<float>(s_g3 + c0), g3c = load4<float>(s_g3 + FP + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f.v[i] *= g3f.v[i];
          c.v[i] *= g3c.v[i];
        }
        ss *= inv2n;
      }
      store4(row + c0, f);
      store4(row + FP + c0, c);
      if (q4 == 0) row[2 * FP] = ss;  // |q|^2 rides in the row's pad (column 2FP): one pointer serves the whole triplet
    }
    dma_wait();  // round 0's operand rows (issued before the frame loop / in the last round)
    if (s == sg && !RN_PROBE(32)) split_landed_tiles();  // (later frames: split at the end of the previous frame)
    __syncthreads();

    // ================= destination edges, 16 per round (one per lane group)
    // Destination of this lane group in round `round` (-1: none).  A round with at most 8
    // destinations is split: groups g and g+8 share destination g, half of its triplets each.
    auto dest_index = [&](int round) {
      const int rm = dcount - round * NG;
      const int sl = (rm <= NG / 2) ? (grp & (NG / 2 - 1)) : grp;
      return sl < rm ? round * NG + sl : -1;
    };
    Vec4<float> nkf, nkc;  // Wk node[a_d] of this group's destination, fetched one round ahead
    if (const int i0 = dest_index(0); i0 >= 0) {
      const float *nk = a.np3 + (nrow0 + d_a[i0]) * (6 * FP) + 4 * FP + c0;
      nkf = load4<float>(nk);
      nkc = load4<float>(nk + FP);
    }
    for (int r = 0; r < nrounds; ++r) {
      // ---- MFMA: P' and c2 pre-activations of 16 destinations from the DMA'd operand rows
      if (!RN_PROBE(16)) {
        f32x4 accP[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 accC[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if constexpr (F16) {
          f16x8 ah[2], al[2];
          load_split_a(atile, l15, quad, ah, al);
          bW4.template product_split<false>(ah, al, accP);
          load_split_a(atile + NG * FP, l15, quad, ah, al);
          bWc.template product_split<false>(ah, al, accC);
        } else {
          float af[KS];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(atile + l15 * FP + (((4 * quad + j) ^ l15) & 15) * 4);
            af[4 * j] = v.x; af[4 * j + 1] = v.y; af[4 * j + 2] = v.z; af[4 * j + 3] = v.w;
          }
          bW4.product(af, accP);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int off = l15 * FP + (((4 * quad + j) ^ l15) & 15) * 4;
            const float4 x = *reinterpret_cast<const float4 *>(atile + NG * FP + off);
            const float4 y = *reinterpret_cast<const float4 *>(atile + 2 * NG * FP + off);
            af[4 * j] = x.x * y.x; af[4 * j + 1] = x.y * y.y; af[4 * j + 2] = x.z * y.z; af[4 * j + 3] = x.w * y.w;
          }
          bWc.product(af, accC);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {  // row l15, columns colbase + 16 t + 4 quad .. + 3
          *reinterpret_cast<f32x4 *>(bufP + l15 * LDQ + colbase + 16 * t + 4 * quad) = accP[t];
          *reinterpret_cast<f32x4 *>(bufC + l15 * LDQ + colbase + 16 * t + 4 * quad) =
              accC[t] * invc2 + *reinterpret_cast<const f32x4 *>(c2bias_p + 16 * t);
        }
      }
      __syncthreads();  // S1: bufP / bufC complete, operand tiles free
      if (r + 1 < nrounds) prefetch_round(s, r + 1);
      else if (s + nsg < a.S) prefetch_round(s + nsg, 0);

      // ---- VALU: lane group `grp` owns destination r*16 + slot
      const int rem = dcount - r * NG;
      const bool split = rem <= NG / 2;                        // uniform over the workgroup
      const int slot = split ? (grp & (NG / 2 - 1)) : grp;
      const int part = split ? (grp >> 3) : 0;                 // which half of the triplets
      const bool active = slot < rem;
      const int i = r * NG + slot;
      const int64_t drow = active ? erow0 + d_edge[i] : 0;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      Vec4<float> old;
      if (active) {
        if (part == 0) old = load4<float>(a.edge_in + drow * FP + c0);
        float pf[4], pc[4];
        {
#if RN_FUSED_NK_SAMEROUND
          {
            const float *nk = a.np3 + (nrow0 + d_a[i]) * (6 * FP) + 4 * FP + c0;
            nkf = load4<float>(nk);
            nkc = load4<float>(nk + FP);
          }
#endif
          const Vec4<float> xf = load4<float>(bufP + slot * LDQ + c0), xc = load4<float>(bufP + slot * LDQ + FP + c0);
          const Vec4<float> jf = load4<float>(nj + (size_t)d_bl[i] * 2 * FP + c0);
          const Vec4<float> jc = load4<float>(nj + (size_t)d_bl[i] * 2 * FP + FP + c0);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pf[k] = fmaf(xf.v[k], inv4, nkf.v[k]) + jf.v[k];
            pc[k] = fmaf(xc.v[k], inv4, nkc.v[k]) + jc.v[k];
          }
        }
#if RN_FUSED_NK_SAMEROUND  // experiment (profiles/r03/determinism.txt): nothing per-destination crosses the MFMA phase in registers
        if (const int inext = -1; inext >= 0) {
#else
        if (const int inext = (r + 1 < nrounds) ? dest_index(r + 1) : -1; inext >= 0) {  // next round's Wk node[a_d]
#endif
          const float *nk = a.np3 + (nrow0 + d_a[inext]) * (6 * FP) + 4 * FP + c0;
          nkf = load4<float>(nk);
          nkc = load4<float>(nk + FP);
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) sum += pf[k] + pc[k];
        const float mean = lg_sum<LG>(sum) * inv2n;
        float sp = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pf[k] = (!PAD || k < nvalid) ? pf[k] - mean : 0.f;
          pc[k] = (!PAD || k < nvalid) ? pc[k] - mean : 0.f;
          sp += pf[k] * pf[k] + pc[k] * pc[k];
        }
        sp = lg_sum<LG>(sp);

        const int rb = d_rb[i], cnt = RN_PROBE(1) ? 0 : d_cnt[i], rskip = d_skip[i];
#if RN_FUSED_PRIO
        __builtin_amdgcn_s_setprio(0);  // the triplet loop is always ready to issue: let the other wave's sparse phases go first
#endif
        const int half = split ? (cnt + 1) / 2 : cnt;
        const int t0 = part ? half : 0, t1 = part ? cnt : half;  // this group's triplets
        if constexpr (FASTG) {
          // pd = p/gamma * (2/2Fe), pg = p*gamma; var + eps = pd.qg + (|p|^2/2Fe + eps) + |q|^2/2Fe
          float pdf[4], pdc[4];
          {
            const Vec4<float> igf = load4<float>(s_ig3 + c0), igc = load4<float>(s_ig3 + FP + c0);
            const Vec4<float> g3f = load4<float>(s_g3 + c0), g3c = load4<float>(s_g3 + FP + c0);
            const float two_inv = 2.0f * inv2n;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              pdf[k] = pf[k] * igf.v[k] * two_inv;
              pdc[k] = pc[k] * igc.v[k] * two_inv;
              pf[k] *= g3f.v[k];
              pc[k] *= g3c.v[k];
            }
          }
          const float spe = sp * inv2n + 1e-5f;
#if RN_FUSED_PK
          // Two columns per instruction: v_pk_add_f32 / v_pk_fma_f32 carry the non-transcendental
          // half of the loop at twice the columns per issue slot (the exp2 / rcp stay per column).
          f32x2 pf2[2], pc2[2], pdf2[2], pdc2[2], bf2[2], bc2[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            pf2[hh] = f32x2{pf[2 * hh], pf[2 * hh + 1]};
            pc2[hh] = f32x2{pc[2 * hh], pc[2 * hh + 1]};
            pdf2[hh] = f32x2{pdf[2 * hh], pdf[2 * hh + 1]};
            pdc2[hh] = f32x2{pdc[2 * hh], pdc[2 * hh + 1]};
            bf2[hh] = f32x2{b3f[2 * hh], b3f[2 * hh + 1]};
            bc2[hh] = f32x2{b3c[2 * hh], b3c[2 * hh + 1]};
          }
          struct QRow {
            float4 f, c;  // columns c0..c0+3 of the filter and core halves of one source row
            float s;      // its |q|^2 term
          };
          const int sdelta = 2 * FP - c0;  // from this lane's filter columns of a row to the row's |q|^2
          auto load_q = [&](const float *qr) {
            return QRow{*reinterpret_cast<const float4 *>(qr), *reinterpret_cast<const float4 *>(qr + FP), qr[sdelta]};
          };
          auto triplet_q = [&](const QRow &q, float (&sumk)[4]) {
            const float4 qfv = q.f, qcv = q.c;
            const f32x2 qf2[2] = {{qfv.x, qfv.y}, {qfv.z, qfv.w}}, qc2[2] = {{qcv.x, qcv.y}, {qcv.z, qcv.w}};
            f32x2 d2 = pdf2[0] * qf2[0];
            f32x2 d3 = pdc2[0] * qc2[0];
            d2 = __builtin_elementwise_fma(pdf2[1], qf2[1], d2);
            d3 = __builtin_elementwise_fma(pdc2[1], qc2[1], d3);
            d2 += d3;
            const float dot = lg_sum<LG>(d2.x + d2.y);
            float ve = dot + (spe + q.s);
            ve = ve > 1e-5f ? ve : 1e-5f;
            const float rstd = fast_rsq(ve);
            const f32x2 rstd2 = {rstd, rstd}, one2 = {1.0f, 1.0f};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const f32x2 xf = __builtin_elementwise_fma(pf2[hh] + qf2[hh], rstd2, bf2[hh]);
              const f32x2 xc = __builtin_elementwise_fma(pc2[hh] + qc2[hh], rstd2, bc2[hh]);
              const f32x2 e1 = {fast_exp2(xf.x), fast_exp2(xf.y)}, e2 = {fast_exp2(xc.x), fast_exp2(xc.y)};
              const f32x2 t2 = e2 + one2;  // (1 + e1)(1 + e2) = t2 + e1 t2: one fma
              const f32x2 den = __builtin_elementwise_fma(e1, t2, t2);
              const f32x2 rd = {fast_rcp(den.x), fast_rcp(den.y)};
              f32x2 sk = {sumk[2 * hh], sumk[2 * hh + 1]};
              sk = __builtin_elementwise_fma(e2 - one2, rd, sk);
              sumk[2 * hh] = sk.x;
              sumk[2 * hh + 1] = sk.y;
            }
          };
          auto triplet = [&](const float *qr, float (&sumk)[4]) { triplet_q(load_q(qr), sumk); };
#else
          auto triplet = [&](int rq, float (&sumk)[4]) {
            const float *qr = bufQ + rq * LDQ + c0;
            const Vec4<float> qf = load4<float>(qr), qc = load4<float>(qr + FP);
            float dotf = pdf[0] * qf.v[0], dotc = pdc[0] * qc.v[0];  // two short chains, not one of eight
#pragma unroll
            for (int k = 1; k < 4; ++k) {
              dotf = fmaf(pdf[k], qf.v[k], dotf);
              dotc = fmaf(pdc[k], qc.v[k], dotc);
            }
            float dot = lg_sum<LG>(dotf + dotc);
            float ve = dot + (spe + bufQ[rq * LDQ + 2 * FP]);
            ve = ve > 1e-5f ? ve : 1e-5f;
            const float rstd = fast_rsq(ve);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float e1 = fast_exp2((pf[k] + qf.v[k]) * rstd + b3f[k]);
              const float e2 = fast_exp2((pc[k] + qc.v[k]) * rstd + b3c[k]);
              const float t2 = 1.0f + e2;  // (1 + e1)(1 + e2) = t2 + e1 t2: one fma
              sumk[k] = fmaf(e2 - 1.0f, fast_rcp(fmaf(e1, t2, t2)), sumk[k]);
            }
          };
#endif
#if RN_FUSED_PK && RN_FUSED_PAIRWISE
          // two independent triplets per iteration: at two waves per SIMD the second chain
          // fills the dependency stalls of the first (summation order: even/odd partial sums).
          // One LDS pointer per lane steps from row to row (two rows where the numbering jumps over the
          // reverse edge) instead of a row index and an address per triplet.
          // (requesting the next pair's first row early -- a rotated loop, with or without scheduling
          //  barriers -- measured 1-3 % slower: the other wave of the SIMD already covers this latency)
          float acc2[4] = {0.f, 0.f, 0.f, 0.f};
          const int tskip = rskip - rb;
          auto step = [&](const float *p, int tnext) { return p + (tnext == tskip ? 2 * LDQ : LDQ); };
          const float *qr = bufQ + (rb + t0 + (t0 >= tskip ? 1 : 0)) * LDQ + c0;
          int t = t0;
          for (; t + 1 < t1; t += 2) {
            const float *qn = step(qr, t + 1);
            triplet(qr, acc);
            triplet(qn, acc2);
            qr = step(qn, t + 2);
          }
          if (t < t1) triplet(qr, acc);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += acc2[k];
#elif RN_FUSED_PK
          for (int t = t0; t < t1; ++t) triplet(bufQ + (rb + t + ((rb + t >= rskip) ? 1 : 0)) * LDQ + c0, acc);
#else
          for (int t = t0; t < t1; ++t) triplet(rb + t + ((rb + t >= rskip) ? 1 : 0), acc);
#endif
        } else {
          const Vec4<float> g3fv = load4<float>(s_g3 + c0), g3cv = load4<float>(s_g3 + FP + c0);
          const float g3f[4] = {g3fv.v[0], g3fv.v[1], g3fv.v[2], g3fv.v[3]}, g3c[4] = {g3cv.v[0], g3cv.v[1], g3cv.v[2], g3cv.v[3]};
          for (int t = t0; t < t1; ++t) {
            const int rq = rb + t + ((rb + t >= rskip) ? 1 : 0);
            const float *qr = bufQ + rq * LDQ + c0;
            const Vec4<float> qf = load4<float>(qr), qc = load4<float>(qr + FP);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) dot += pf[k] * qf.v[k] + pc[k] * qc.v[k];
            dot = lg_sum<LG>(dot);
            const float var = fmaxf((sp + bufQ[rq * LDQ + 2 * FP] + 2.0f * dot) * inv2n, 0.0f);
            const float rstd = fast_rsq(var + 1e-5f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float yf = ((pf[k] + qf.v[k]) * rstd) * g3f[k] + b3f[k];
              float yc = ((pc[k] + qc.v[k]) * rstd) * g3c[k] + b3c[k];
              yc = fminf(fmaxf(yc, -43.28f), 43.28f);
              const float e1 = fast_exp2(yf), e2 = fast_exp2(yc);
              acc[k] += (e2 - 1.0f) * fast_rcp((1.0f + e1) * (1.0f + e2));
            }
          }
        }
      }
#if RN_FUSED_PRIO && !RN_FUSED_PRIO_LATE
      __builtin_amdgcn_s_setprio(RN_FUSED_PRIO);
#endif
      if (split) {  // second halves reach their partner through the unused rows 8..15 of bufP
        float *xch = bufP + (NG / 2 + slot) * LDQ + c0;
        if (active && part == 1) store4(xch, Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}});
        __syncthreads();
        if (active && part == 0) {
          const Vec4<float> other = load4<float>(xch);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += other.v[k];
        }
      }
      if (RN_PROBE(2)) {
        if (active && part == 0)
          store4(a.edge_out + drow * FP + c0, Vec4<float>{{acc[0] + old.v[0], acc[1] + old.v[1], acc[2], acc[3]}});
      } else if (active && part == 0) {
        if (a.agg_out) store4(a.agg_out + drow * FP + c0, Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}});
        const LnParams<float> p3n{load4<float>(s_c3n2g + c0), load4<float>(s_c3n2b + c0)};
        const Vec4<float> c3 = ln_row<LG, PAD>(Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}}, p3n, invn, nvalid);
        // c2: gate(LayerNorm(c2_linear(node[b]*node[a]))) -> LayerNorm   (_gnn.py:223-228)
        const LnParams<float> p2f{load4<float>(s_c2n1g + c0), load4<float>(s_c2n1b + c0)};
        const LnParams<float> p2c{load4<float>(s_c2n1g + FP + c0), load4<float>(s_c2n1b + FP + c0)};
        const Vec4<float> c2f = load4<float>(bufC + slot * LDQ + c0), c2c = load4<float>(bufC + slot * LDQ + FP + c0);
        const Vec4<float> g2 = ln_gate<LG, PAD>(c2f, c2c, p2f, p2c, inv2n, nvalid);
        const LnParams<float> p2n{load4<float>(s_c2n2g + c0), load4<float>(s_c2n2b + c0)};
        const Vec4<float> c2 = ln_row<LG, PAD>(g2, p2n, invn, nvalid);
        Vec4<float> out;
#pragma unroll
        for (int k = 0; k < 4; ++k) out.v[k] = fast_tanh(old.v[k] + c2.v[k] + c3.v[k]);
        store4(a.edge_out + drow * FP + c0, out);
      }
#if RN_FUSED_PRIO && RN_FUSED_PRIO_LATE
      __builtin_amdgcn_s_setprio(RN_FUSED_PRIO);
#endif
      dma_wait();
      if ((r + 1 < nrounds || s + nsg < a.S) && !RN_PROBE(32)) split_landed_tiles();
      __syncthreads();  // S2: bufP / bufC (and, after the last round, bufQ / nj) may be rewritten; next operand rows landed
    }
  }
}

template <bool PAD, bool FASTG, bool F16>
static void launch_cfg(const EdgeFusedArgs &a, size_t lds, hipStream_t st) {
  auto kern = &edge_block_fused_kernel<PAD, FASTG, F16>;
  // experiment knob (profiles/r03/determinism.txt): a dynamic-LDS request of at least this many KiB,
  // e.g. 96 leaves room for ONE workgroup per CU, so no SIMD hosts waves of two workgroups
  static const size_t min_lds = getenv("RN_POTGNN_FUSED_MIN_LDS_KB") ? (size_t)atoi(getenv("RN_POTGNN_FUSED_MIN_LDS_KB")) * 1024 : 0;
  if (lds < min_lds) lds = min_lds;
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
  per_cu = std::min(per_cu, 2);
  int nsg = per_cu * cus / a.g.num_tiles;
  nsg = nsg < 1 ? 1 : (nsg > a.S ? a.S : nsg);
  kern<<<(unsigned)nsg * (unsigned)a.g.num_tiles, 256, lds, st>>>(a);
}

void launch_edge_fused(const float *edge_in, float *edge_out, const float *node, const float *np3,
                       float *agg_out, int S, const Graph &g, Dims d, const PassW<float> &w, bool f16,
                       hipStream_t st) {
  if (S == 0 || g.E == 0) return;
  EdgeFusedArgs a{edge_in, edge_out, node, np3, agg_out, S, g, d, w};
#if RN_FUSED_PROBE
  a.probe = getenv("RN_FUSED_PROBE_MASK") ? atoi(getenv("RN_FUSED_PROBE_MASK")) : 0;
#endif
  const size_t lds = edge_fused_lds_bytes(g);
  const bool pad = d.Fe != d.FeP;
  const bool fast = (w.c3_fast & 1) != 0;
  if (f16) {
    if (pad) fast ? launch_cfg<true, true, true>(a, lds, st) : launch_cfg<true, false, true>(a, lds, st);
    else fast ? launch_cfg<false, true, true>(a, lds, st) : launch_cfg<false, false, true>(a, lds, st);
  } else {
    if (pad) fast ? launch_cfg<true, true, false>(a, lds, st) : launch_cfg<true, false, false>(a, lds, st);
    else fast ? launch_cfg<false, true, false>(a, lds, st) : launch_cfg<false, false, false>(a, lds, st);
  }
}

}  // namespace rn
