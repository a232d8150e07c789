// Opt-in experiment kernels of round 3 that measured level with or slower than the product kernels
// (profiles/r03/edge2_experiment.txt, edge3_experiment.txt, node_tile_sweep.txt).  Compiled only with
// -DRN_EXPERIMENTS=1 (RN_EXTRA_FLAGS=-DRN_EXPERIMENTS=1 bash build.sh); the product build carries none of them.
#include "../ramannoodle_amd/csrc/fused_common.hpp"

namespace rn {

// ============================================================================ EdgeBlock, twelve waves
// The same frame structure as edge_block_fused_kernel with THREE waves per SIMD instead of two: its VALU idles
// 39 % of the time at two waves per SIMD (profiles/r03/edge2_experiment.txt), and a third wave needs <= 168
// VGPRs.  Two things make room: the c2 branch runs in edge_c2_kernel (its weight fragments, 32 VGPRs, its operand
// tiles and its pre-activation rows leave this kernel; the finished c2 row of a destination is one 16-byte load
// per lane in the epilogue), and ONE 768-thread workgroup owns the CU's whole LDS, so a tile holds 8-10 atoms and
// a round serves 48 destinations: three 16-row MFMA tiles side by side (wave w: tile w / 4, columns 32 (w % 4)),
// fewer, fuller rounds, and a third of the barriers per destination.
struct Edge3Args {
  const float *edge_in;
  float *edge_out;
  const float *np3;  // [S*N, 6FP] = node * (Wi | Wj(+bias) | Wk)
  const float *c2;   // [S*E, FP]  the finished c2 embedding of every edge (edge_c2_kernel)
  float *agg_out;    // taped runs: the pre-LayerNorm triplet sums [S*E, FP]; else null
  int S;
  Graph g;
  Dims d;
  PassW<float> w;
};

namespace {
#ifndef RN_E3_WAVES
#define RN_E3_WAVES 12  // 12: one workgroup per CU (158 KiB of LDS); 6: two workgroups per CU (80 KiB each)
#endif
constexpr int NW3 = RN_E3_WAVES;     // waves per workgroup
constexpr int NT3 = 64 * NW3;        // threads
constexpr int NG3 = NT3 / LG;        // lane groups = destinations per round
constexpr int MT3 = (NG3 + 15) / 16; // 16-row MFMA tiles per round
constexpr int G43 = NW3 / 4;         // complete groups of four waves (one wave per 32 of the 128 columns): they run the MFMAs
constexpr int WGS3 = NW3 <= 4 ? 3 : (NW3 <= 6 ? 2 : 1);  // workgroups per CU (three waves per SIMD in every form)
constexpr bool NJ_LDS3 = NW3 > 4;    // four waves: the Wj node[j] rows come from L2 per destination (53 KiB of LDS per workgroup)
struct Fused3Lds {
  size_t bufQ, bufP, atile, nj, lnp, ints, total;
};
__host__ __device__ inline Fused3Lds fused3_lds(int maxR, int maxD, int maxN) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  Fused3Lds L;
  size_t off = 0;
  L.bufQ = off; off += up((size_t)maxR * LDQ * 4);
  L.bufP = off; off += up((size_t)MT3 * 16 * LDQ * 4);
  L.atile = off; off += (size_t)MT3 * 16 * FP * 4;
  L.nj = off; off += NJ_LDS3 ? up((size_t)maxN * 2 * FP * 4) : 0;
  L.lnp = off; off += (size_t)6 * FP * 4;
  L.ints = off; off += up(((size_t)maxR + 6 * (size_t)maxD) * 4);
  L.total = off;
  return L;
}
}  // namespace

template <bool PAD>
__global__ __launch_bounds__(NT3, WGS3) void edge_block3_kernel(Edge3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const Fused3Lds L = fused3_lds(g.et_max_out_rows, g.et_max_in_rows, g.et_max_nodes);
  float *bufQ = reinterpret_cast<float *>(smem_raw + L.bufQ);   // [maxR][LDQ] centred source rows (|q|^2 in the pad)
  float *bufP = reinterpret_cast<float *>(smem_raw + L.bufP);   // [48][LDQ] W4 edge_d of a round
  float *atile = reinterpret_cast<float *>(smem_raw + L.atile); // [48][64] swizzled operand rows
  float *nj = reinterpret_cast<float *>(smem_raw + L.nj);       // [maxN][2FP] Wj node[j] + bias
  float *lnp = reinterpret_cast<float *>(smem_raw + L.lnp);
  float *s_c3n2g = lnp, *s_c3n2b = lnp + FP, *s_g3 = lnp + 2 * FP, *s_ig3 = lnp + 4 * FP;
  int *qb = reinterpret_cast<int *>(smem_raw + L.ints);
  const int maxD = g.et_max_in_rows;
  int *d_edge = qb + g.et_max_out_rows, *d_a = d_edge + maxD, *d_bl = d_a + maxD, *d_rb = d_bl + maxD,
      *d_cnt = d_rb + maxD, *d_skip = d_cnt + maxD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int mtile = wave >> 2;           // first 16-row tile of a step this wave multiplies (then every G43-th)
  const int colbase = (wave & 3) * 32;   // and which 32 of the 128 pre-activation columns
  const bool mma = wave < 4 * G43;       // (waves beyond the last complete group of four sit the MFMA phases out)
  __builtin_amdgcn_s_setprio(RN_FUSED_PRIO);

  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.et_num;
  const int sg = logical / g.et_num, nsg = gridDim.x / g.et_num;
  const int j0 = g.et_begin[tile], j1 = g.et_begin[tile + 1];
  const int eo0 = g.out_ptr[j0], rows = g.out_ptr[j1] - eo0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
  const int nrounds = (dcount + NG3 - 1) / NG3;

  for (int c = tid; c < 2 * FP; c += NT3) {
    const float gam = a.w.c3_norm_1.g[c] * (c < FP ? -kLog2e : 2.0f * kLog2e);
    s_g3[c] = gam;
    s_ig3[c] = ((c % FP) < a.d.Fe) ? 1.0f / gam : 0.0f;
    if (c < FP) {
      s_c3n2g[c] = a.w.c3_norm_2.g[c];
      s_c3n2b[c] = a.w.c3_norm_2.b[c];
    }
  }
  for (int r = tid; r < rows; r += NT3) qb[r] = g.edge_b[eo0 + r];
  for (int i = tid; i < dcount; i += NT3) {
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

  const float s4 = a.w.mfma_scale[2], inv4 = a.w.mfma_scale[3];
  const float s5 = a.w.mfma_scale[4], inv5 = a.w.mfma_scale[5];
  WaveB<true> bW4;  // resident; W5's fragments are rebuilt every frame (the registers are the loop's in between)
  bW4.load(a.w.c3_WeT, 4 * FP, colbase, l15, quad, s4);

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

  // LDS-DMA of the 48 destination rows of round `r` of frame `s`: wave w brings rows 4w..4w+3;
  // slot (row, piece p) receives global piece p ^ row
  auto prefetch_round = [&](int s, int r) {
    const int row = 4 * wave + quad;
    const int i = min(r * NG3 + row, dcount - 1);
    const int piece = (l15 ^ row) & 15;
    dma16(a.edge_in + ((int64_t)s * g.E + d_edge[i]) * FP + 4 * piece, atile + wave * 256);
  };
  auto split_landed_tile = [&]() {  // every lane turns the slot IT fetched into [hi x4 | lo x4] halves
    float *slot = atile + wave * 256 + lane * 4;
    *reinterpret_cast<float4 *>(slot) = split_slot(*reinterpret_cast<const float4 *>(slot));
  };
  if (sg < a.S && dcount > 0) prefetch_round(sg, 0);

  for (int s = sg; s < a.S; s += nsg) {
    const int64_t erow0 = (int64_t)s * g.E, nrow0 = (int64_t)s * g.N;
    if constexpr (NJ_LDS3) {
      for (int i = tid; i < (j1 - j0) * (2 * FP / 4); i += NT3) {
        const int n = i / (2 * FP / 4), c = (i % (2 * FP / 4)) * 4;
        store4(nj + (size_t)n * 2 * FP + c, load4<float>(a.np3 + (nrow0 + j0 + n) * (6 * FP) + 2 * FP + c));
      }
    }
    // ================= source rows: W5 edge_e by MFMA -> bufQ (raw), 16 G43 rows per step
    {
      WaveB<true> bW5;
      bW5.load(a.w.c3_WeT + 2 * FP, 4 * FP, colbase, l15, quad, s5);
      for (int mt = mtile; mma && mt * 16 < rows; mt += G43) {
        float af[KS];
        const float *src = a.edge_in + (erow0 + eo0 + min(mt * 16 + l15, rows - 1)) * FP + quad * KS;
#pragma unroll
        for (int k4 = 0; k4 < KS; k4 += 4) {
          const float4 v = *reinterpret_cast<const float4 *>(src + k4);
          af[k4] = v.x; af[k4 + 1] = v.y; af[k4 + 2] = v.z; af[k4 + 3] = v.w;
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bW5.product(af, acc);
        if (const int r = mt * 16 + l15; r < rows) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
            *reinterpret_cast<f32x4 *>(bufQ + r * LDQ + colbase + 16 * t + 4 * quad) = acc[t];
        }
      }
    }
    __syncthreads();
    // ---- add Wi node[b_e], centre, fold the gate scale, record |q|^2 (padded columns forced to 0)
    for (int r = grp; r < rows; r += NG3) {
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
      const Vec4<float> g3f = load4<float>(s_g3 + c0), g3c = load4<float>(s_g3 + FP + c0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f.v[i] *= g3f.v[i];
        c.v[i] *= g3c.v[i];
      }
      store4(row + c0, f);
      store4(row + FP + c0, c);
      if (q4 == 0) row[2 * FP] = ss * inv2n;
    }
    dma_wait();  // round 0's operand rows (issued before the frame loop / in the last round)
    if (s == sg) split_landed_tile();  // (later frames: split at the end of the previous frame)
    __syncthreads();

    // Destination of this lane group in round `round` (-1: none).  A round with at most 24 destinations is
    // split: groups g and g+24 share destination g, half of its triplets each.
    auto dest_index = [&](int round) {
      const int rm = dcount - round * NG3;
      const int sl = (rm <= NG3 / 2) ? (grp >= NG3 / 2 ? grp - NG3 / 2 : grp) : grp;
      return sl < rm ? round * NG3 + sl : -1;
    };
    Vec4<float> nkf, nkc;  // Wk node[a_d] of this group's destination, fetched one round ahead
    if (const int i0 = dest_index(0); i0 >= 0) {
      const float *nk = a.np3 + (nrow0 + d_a[i0]) * (6 * FP) + 4 * FP + c0;
      nkf = load4<float>(nk);
      nkc = load4<float>(nk + FP);
    }
    for (int r = 0; r < nrounds; ++r) {
      // ---- MFMA: P' of the round's destinations from the DMA'd, pre-split operand rows
      for (int mt = mtile; mma && mt < MT3; mt += G43) {
        f32x4 accP[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f16x8 ah[2], al[2];
        load_split_a(atile + mt * 16 * FP, l15, quad, ah, al);
        bW4.product_split(ah, al, accP);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<f32x4 *>(bufP + (mt * 16 + l15) * LDQ + colbase + 16 * t + 4 * quad) = accP[t];
      }
      __syncthreads();  // S1: bufP complete, operand tile free
      if (r + 1 < nrounds) prefetch_round(s, r + 1);
      else if (s + nsg < a.S) prefetch_round(s + nsg, 0);

      const int rem = dcount - r * NG3;
      const bool split = rem <= NG3 / 2;                        // uniform over the workgroup
      const int part = (split && grp >= NG3 / 2) ? 1 : 0;       // which half of the triplets
      const int slot = part ? grp - NG3 / 2 : grp;
      const bool active = slot < rem;
      const int i = r * NG3 + slot;
      const int64_t drow = active ? erow0 + d_edge[i] : 0;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      Vec4<float> old, c2v;
      if (active) {
        if (part == 0) {
          old = load4<float>(a.edge_in + drow * FP + c0);
          c2v = load4<float>(a.c2 + drow * FP + c0);
        }
        float pf[4], pc[4];
        {
          const Vec4<float> xf = load4<float>(bufP + slot * LDQ + c0), xc = load4<float>(bufP + slot * LDQ + FP + c0);
          const float *njr = NJ_LDS3 ? nj + (size_t)d_bl[i] * 2 * FP + c0
                                     : a.np3 + (nrow0 + j0 + d_bl[i]) * (6 * FP) + 2 * FP + c0;
          const Vec4<float> jf = load4<float>(njr), jc = load4<float>(njr + FP);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pf[k] = fmaf(xf.v[k], inv4, nkf.v[k]) + jf.v[k];
            pc[k] = fmaf(xc.v[k], inv4, nkc.v[k]) + jc.v[k];
          }
        }
        if (const int inext = (r + 1 < nrounds) ? dest_index(r + 1) : -1; inext >= 0) {  // next round's Wk node[a_d]
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

        const int rb = d_rb[i], cnt = d_cnt[i], rskip = d_skip[i];
        __builtin_amdgcn_s_setprio(0);  // the triplet loop is always ready to issue: let the other waves' sparse phases go first
        const int half = split ? (cnt + 1) / 2 : cnt;
        const int t0 = part ? half : 0, t1 = part ? cnt : half;  // this group's triplets
        // pd = p/gamma * (2/2Fe), pg = p*gamma; var + eps = pd.qg + (|p|^2/2Fe + eps) + |q|^2/2Fe
        f32x2 pf2[2], pc2[2], pdf2[2], pdc2[2], bf2[2], bc2[2];
        {
          const Vec4<float> igf = load4<float>(s_ig3 + c0), igc = load4<float>(s_ig3 + FP + c0);
          const Vec4<float> g3f = load4<float>(s_g3 + c0), g3c = load4<float>(s_g3 + FP + c0);
          const float two_inv = 2.0f * inv2n;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            pdf2[hh] = f32x2{pf[2 * hh] * igf.v[2 * hh] * two_inv, pf[2 * hh + 1] * igf.v[2 * hh + 1] * two_inv};
            pdc2[hh] = f32x2{pc[2 * hh] * igc.v[2 * hh] * two_inv, pc[2 * hh + 1] * igc.v[2 * hh + 1] * two_inv};
            pf2[hh] = f32x2{pf[2 * hh] * g3f.v[2 * hh], pf[2 * hh + 1] * g3f.v[2 * hh + 1]};
            pc2[hh] = f32x2{pc[2 * hh] * g3c.v[2 * hh], pc[2 * hh + 1] * g3c.v[2 * hh + 1]};
            bf2[hh] = f32x2{b3f[2 * hh], b3f[2 * hh + 1]};
            bc2[hh] = f32x2{b3c[2 * hh], b3c[2 * hh + 1]};
          }
        }
        const float spe = sp * inv2n + 1e-5f;
        const int sdelta = 2 * FP - c0;  // from this lane's filter columns of a row to the row's |q|^2
        auto triplet = [&](const float *qr, float (&sumk)[4]) {
          const float4 qfv = *reinterpret_cast<const float4 *>(qr), qcv = *reinterpret_cast<const float4 *>(qr + FP);
          const float qs = qr[sdelta];
          const f32x2 qf2[2] = {{qfv.x, qfv.y}, {qfv.z, qfv.w}}, qc2[2] = {{qcv.x, qcv.y}, {qcv.z, qcv.w}};
          f32x2 d2 = pdf2[0] * qf2[0];
          f32x2 d3 = pdc2[0] * qc2[0];
          d2 = __builtin_elementwise_fma(pdf2[1], qf2[1], d2);
          d3 = __builtin_elementwise_fma(pdc2[1], qc2[1], d3);
          d2 += d3;
          const float dot = lg_sum<LG>(d2.x + d2.y);
          float ve = dot + (spe + qs);
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
        const int tskip = rskip - rb;
        auto step = [&](const float *p, int tnext) { return p + (tnext == tskip ? 2 * LDQ : LDQ); };
        const float *qr = bufQ + (rb + t0 + (t0 >= tskip ? 1 : 0)) * LDQ + c0;
#if RN_E3_PAIRWISE
        float acc2[4] = {0.f, 0.f, 0.f, 0.f};
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
#else
        for (int t = t0; t < t1; ++t) {
          triplet(qr, acc);
          qr = step(qr, t + 1);
        }
#endif
      }
      __builtin_amdgcn_s_setprio(RN_FUSED_PRIO);
      if (split) {  // second halves reach their partner through the unused rows 24..47 of bufP
        float *xch = bufP + (NG3 / 2 + slot) * LDQ + c0;
        if (active && part == 1) store4(xch, Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}});
        __syncthreads();
        if (active && part == 0) {
          const Vec4<float> other = load4<float>(xch);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += other.v[k];
        }
      }
      if (active && part == 0) {
        if (a.agg_out) store4(a.agg_out + drow * FP + c0, Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}});
        const LnParams<float> p3n{load4<float>(s_c3n2g + c0), load4<float>(s_c3n2b + c0)};
        const Vec4<float> c3 = ln_row<LG, PAD>(Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}}, p3n, invn, nvalid);
        Vec4<float> out;
#pragma unroll
        for (int k = 0; k < 4; ++k) out.v[k] = fast_tanh(old.v[k] + c2v.v[k] + c3.v[k]);
        store4(a.edge_out + drow * FP + c0, out);
      }
      dma_wait();
      if (r + 1 < nrounds || s + nsg < a.S) split_landed_tile();
      __syncthreads();  // S2: bufP (and, after the last round, bufQ / nj) may be rewritten; next operand rows landed
    }
  }
}

size_t edge3_lds_bytes(int rows, int in_rows, int nodes) { return fused3_lds(rows, in_rows, nodes).total; }
bool edge3_supported(const Graph &g, Dims d) {
  return d.FnP == 64 && d.FeP == 64 && g.E > 0 && g.et_num > 0 &&
         fused3_lds(g.et_max_out_rows, g.et_max_in_rows, g.et_max_nodes).total <= kEdge3LdsBudget;
}
// the twelve-wave kernel exists for the split-f16 products with the gate scale folded into the rows
bool edge3_applicable(const PassW<float> &w, bool f16) { return f16 && (w.c3_fast & 1) != 0; }

void launch_edge3(const float *edge_in, float *edge_out, const float *np3, const float *c2, float *agg_out, int S,
                  const Graph &g, Dims d, const PassW<float> &w, hipStream_t st) {
  if (S == 0 || g.E == 0) return;
  Edge3Args a{edge_in, edge_out, np3, c2, agg_out, S, g, d, w};
  const size_t lds = fused3_lds(g.et_max_out_rows, g.et_max_in_rows, g.et_max_nodes).total;
  const bool pad = d.Fe != d.FeP;
  auto kern = pad ? &edge_block3_kernel<true> : &edge_block3_kernel<false>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int nsg = WGS3 * cus / g.et_num;  // three waves per SIMD: one twelve-wave, two six-wave or three four-wave workgroups per CU
  nsg = nsg < 1 ? 1 : (nsg > S ? S : nsg);
  kern<<<(unsigned)nsg * (unsigned)g.et_num, NT3, lds, st>>>(a);
}

// ---------------------------------------------------------------------------- wave-autonomous NodeBlock
// The round structure above spends its time between barriers (phase probe, profiles/r03/node_tile_sweep.txt:
// MFMA stage, gate and per-atom phase of a workgroup are serialised by two barriers per 16-row round).  Here a
// WAVE owns a 16-row tile of in-edge rows for ALL 128 output columns: the edge part of c1_linear stays in its
// registers as split-f16 B fragments (128 VGPRs), the operand rows go from global memory straight into A
// fragments (one 64-byte run per lane, fetched one tile ahead), and LayerNorm(2Fn) + gate run on the
// accumulators -- a row is spread over the four lanes l15 + 16 quad, so its statistics take two cross-row
// exchanges.  Only the per-atom sums need the workgroup: the gated rows of the whole atom tile meet in LDS, two
// barriers per (frame, atom tile) instead of two per 16 rows.  Two workgroups per CU (register-bound).
namespace {
struct NodeWaveLds {
  size_t gated, nj, lnp, ints, total;
};
__host__ __device__ inline NodeWaveLds node_wave_lds(int maxD, int maxN) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  NodeWaveLds L;
  size_t off = 0;
  L.gated = off; off += up((size_t)maxD * LDG * 4);
  L.nj = off; off += up((size_t)maxN * 2 * FP * 4);
  L.lnp = off; off += (size_t)6 * FP * 4;
  L.ints = off; off += up((size_t)2 * maxD * 4);
  L.total = off;
  return L;
}
}  // namespace

template <bool PAD>
__global__ __launch_bounds__(256, 2) void node_block_wave_kernel(NodeFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const NodeWaveLds L = node_wave_lds(g.nt_max_in_rows, g.nt_max_nodes);
  float *gated = reinterpret_cast<float *>(smem_raw + L.gated);  // [maxD][LDG] gate outputs of the tile
  float *nj = reinterpret_cast<float *>(smem_raw + L.nj);        // [maxN][2FP] W_n node + bias
  float *lnp = reinterpret_cast<float *>(smem_raw + L.lnp);
  float *s_c1g = lnp, *s_c1b = lnp + 2 * FP, *s_fg = lnp + 4 * FP, *s_fb = lnp + 5 * FP;
  int *d_edge = reinterpret_cast<int *>(smem_raw + L.ints), *d_bl = d_edge + g.nt_max_in_rows;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.nt_num;
  const int sg = logical / g.nt_num, nsg = gridDim.x / g.nt_num;
  const int j0 = g.nt_begin[tile], j1 = g.nt_begin[tile + 1];
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
  const int nmt = (dcount + 15) / 16;

  for (int c = tid; c < 2 * FP; c += 256) {
    s_c1g[c] = a.w.c1_norm.g[c];
    s_c1b[c] = a.w.c1_norm.b[c];
    if (c < FP) {
      s_fg[c] = a.w.final_norm.g[c];
      s_fb[c] = a.w.final_norm.b[c];
    }
  }
  for (int i = tid; i < dcount; i += 256) {
    const int e = g.in_edge[di0 + i];
    d_edge[i] = e;
    d_bl[i] = g.edge_b[e] - j0;
  }
  WaveB<true> bW[4];  // all 128 columns of the edge part of c1_linear: column tile T = 2 p + t covers 16 T .. 16 T + 15
  const float s1 = a.w.mfma_scale[0], inv1 = a.w.mfma_scale[1];
#pragma unroll
  for (int p = 0; p < 4; ++p) bW[p].load(a.w.c1_WeT, 2 * FP, 32 * p, l15, quad, s1);

  const int grp = tid / LG, q4 = tid % LG, c0 = 4 * q4;
  const int nvalid = min(max(a.d.Fn - c0, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fn), invn = 1.0f / (float)a.d.Fn;
  __syncthreads();

  // this lane's run of the operand row of tile-row 16 mt + l15: k = 16 quad .. 16 quad + 15
  float4 nxt[4];
  auto fetch = [&](int s, int mt) {
    const int i = min(mt * 16 + l15, dcount - 1);
    const float *p = a.edge + ((int64_t)s * g.E + d_edge[i]) * FP + 16 * quad;
#pragma unroll
    for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const float4 *>(p + 4 * j);
  };
  if (sg < a.S && wave < nmt) fetch(sg, wave);

  for (int s = sg; s < a.S; s += nsg) {
    const int64_t nrow0 = (int64_t)s * g.N;
    for (int i = tid; i < (j1 - j0) * (2 * FP / 4); i += 256) {
      const int n = i / (2 * FP / 4), c = (i % (2 * FP / 4)) * 4;
      store4(nj + (size_t)n * 2 * FP + c, load4<float>(a.npc1 + (nrow0 + j0 + n) * (2 * FP) + c));
    }
    __syncthreads();  // A: nj complete; the previous frame's per-atom phase is done with `gated`
    for (int mt = wave; mt < nmt; mt += 4) {
      float af[KS];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        af[4 * j] = nxt[j].x; af[4 * j + 1] = nxt[j].y; af[4 * j + 2] = nxt[j].z; af[4 * j + 3] = nxt[j].w;
      }
      {  // the wave's next tile: further down this frame, else its first of the next frame
        int nm = mt + 4, ns = s;
        if (nm >= nmt) {
          nm = wave;
          ns = s + nsg;
        }
        if (ns < a.S && nm < nmt) fetch(ns, nm);
      }
      f16x8 ah[2], al[2];
      split_f16x8(af, ah[0], al[0]);
      split_f16x8(af + 8, ah[1], al[1]);
      f32x4 acc[4][2];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        acc[p][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        bW[p].product_split(ah, al, acc[p]);
      }
      const int irow = mt * 16 + l15, i = min(irow, dcount - 1);
      const float *njr = nj + (size_t)d_bl[i] * 2 * FP + 4 * quad;
      // x[T][j]: column 16 T + 4 quad + j of row l15 (T < 4 filter, T >= 4 core)
      float x[8][4];
      float sum = 0.f;
#pragma unroll
      for (int T = 0; T < 8; ++T) {
        const float4 nb = *reinterpret_cast<const float4 *>(njr + 16 * T);
        const f32x4 v = acc[T >> 1][T & 1];
        x[T][0] = fmaf(v[0], inv1, nb.x);
        x[T][1] = fmaf(v[1], inv1, nb.y);
        x[T][2] = fmaf(v[2], inv1, nb.z);
        x[T][3] = fmaf(v[3], inv1, nb.w);
        sum += (x[T][0] + x[T][1]) + (x[T][2] + x[T][3]);
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float mean = sum * inv2n;
      float q = 0.f;
#pragma unroll
      for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[T][j] -= mean;
          if (PAD) {
            if (16 * (T & 3) + 4 * quad + j >= a.d.Fn) x[T][j] = 0.f;
          }
          q += x[T][j] * x[T][j];
        }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      const float rstd = fast_rsq(q * inv2n + 1e-5f);
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const int col = 16 * T + 4 * quad;
        const float4 gf = *reinterpret_cast<const float4 *>(s_c1g + col), bf = *reinterpret_cast<const float4 *>(s_c1b + col);
        const float4 gc = *reinterpret_cast<const float4 *>(s_c1g + FP + col), bc = *reinterpret_cast<const float4 *>(s_c1b + FP + col);
        Vec4<float> out;
        out.v[0] = gate(x[T][0] * rstd * gf.x + bf.x, x[T + 4][0] * rstd * gc.x + bc.x);
        out.v[1] = gate(x[T][1] * rstd * gf.y + bf.y, x[T + 4][1] * rstd * gc.y + bc.y);
        out.v[2] = gate(x[T][2] * rstd * gf.z + bf.z, x[T + 4][2] * rstd * gc.z + bc.z);
        out.v[3] = gate(x[T][3] * rstd * gf.w + bf.w, x[T + 4][3] * rstd * gc.w + bc.w);
        if (irow < dcount) store4(gated + (size_t)irow * LDG + col, out);
      }
    }
    __syncthreads();  // B: the tile's gated rows are complete
    // ---- per atom: sum over its in-edges (ascending, as the reference's scatter), LayerNorm, residual
    for (int n = grp; n < j1 - j0; n += NG) {
      const int i0 = g.in_ptr[j0 + n] - di0, i1 = g.in_ptr[j0 + n + 1] - di0;
      Vec4<float> sacc{{0.f, 0.f, 0.f, 0.f}};
      for (int i = i0; i < i1; ++i) {
        const Vec4<float> v = load4<float>(gated + (size_t)i * LDG + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) sacc.v[k] += v.v[k];
      }
      const LnParams<float> pn{load4<float>(s_fg + c0), load4<float>(s_fb + c0)};
      const Vec4<float> ln = ln_row<LG, PAD>(sacc, pn, invn, nvalid);
      const Vec4<float> old = load4<float>(a.node_in + (nrow0 + j0 + n) * FP + c0);
      Vec4<float> out;
#pragma unroll
      for (int k = 0; k < 4; ++k) out.v[k] = fast_tanh(old.v[k] + ln.v[k]);
      store4(a.node_out + (nrow0 + j0 + n) * FP + c0, out);
    }
  }
}

// RN_POTGNN_NODE_WAVE=1 selects it (with its own atom tiles, csrc/api.hip).  Measured 1.29-1.30 ms per 2000 frames
// against 1.18 ms for the round-structured kernel (profiles/r03/node_tile_sweep.txt): with 128 VGPRs of weights
// only eight waves fit a CU, and the kernel turns out throughput-bound (VALU ~0.58 ms + matrix pipe ~0.36 ms per
// 2000 frames, poorly overlapped at two waves per SIMD) rather than barrier-bound, so it stays opt-in.
static bool node_wave_enabled() {
  static const bool on = getenv("RN_POTGNN_NODE_WAVE") ? atoi(getenv("RN_POTGNN_NODE_WAVE")) != 0 : false;
  return on;
}
bool node_fused_wave_tiles() { return node_wave_enabled(); }
size_t node_wave_lds_bytes(int tile_in_rows, int tile_nodes) { return node_wave_lds(tile_in_rows, tile_nodes).total; }
void launch_node_wave(const NodeFusedArgs &a, hipStream_t st) {
  const Graph &g = a.g;
  const int S = a.S;
  const bool pad = a.d.Fn != a.d.FnP;
  const size_t wlds = node_wave_lds(g.nt_max_in_rows, g.nt_max_nodes).total;
  auto wkern = pad ? &node_block_wave_kernel<true> : &node_block_wave_kernel<false>;
  if (wlds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wkern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds);
  static int wcus = 0;
  if (wcus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) wcus = prop.multiProcessorCount;
    if (wcus <= 0) wcus = 256;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wkern, 256, wlds) != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, 2);
  int nsg = per_cu * wcus / g.nt_num;
  nsg = nsg < 1 ? 1 : (nsg > S ? S : nsg);
  wkern<<<(unsigned)nsg * (unsigned)g.nt_num, 256, wlds, st>>>(a);
}

// ============================================================================ EdgeBlock, frame-pipelined form
// The kernel above spends 60 % of its time outside the triplet loop, most of it waiting: per frame it
// re-loads and re-splits W5, fetches the tile's source rows 16 at a time with the global-memory latency
// exposed each time (as does the centring pass with its Wi node[b] rows), and only then starts its rounds
// (profiles/r02/edge_phase_probe.txt).  This form removes every such wait from the frame:
//   * Q' of the NEXT frame is staged while the current frame's rounds run: 16 source rows per round ride
//     through the round's MFMA phase next to the 16 destinations' P' (W5 resident in VGPRs beside W4),
//     are centred in the round's VALU phase and land in the other half of a double-buffered bufQ, so a
//     frame is nothing but rounds, and every round is the same;
//   * everything a round reads from global memory -- destination edge rows, source edge rows, the
//     Wi node[b] rows of the centring pass, the Wk node[a] rows of P', Wj node[j] -- arrives by LDS-DMA,
//     requested one round ahead; the VALU phase itself issues two loads (old row, c2 row) it needs only
//     after the triplet loop;
//   * the c2 branch (Linear -> LayerNorm -> gate -> LayerNorm of node[b] * node[a], _gnn.py:223-228)
//     moves to edge_c2_kernel below: it is a per-edge quantity with no triplet structure, and its
//     weight fragments are what made room for a resident W5.
// The double buffer halves the tile (2 x rows of Q'), so the last round of a frame is usually a short one:
// destinations are then split over 2, 4, 8 or 16 lane groups (partial sums added in a fixed order).
// Timing-only probe builds of edge_block2_kernel (RN_EXTRA_FLAGS=-DRN_E2_PROBE=mask; wrong results):
// 1 no output store, 2 no final vmcnt(0) wait, 4 no triplet loop, 8 no MFMA phase, 16 no LDS-DMA / split,
// 32 no centring, 64 no epilogue arithmetic
#ifndef RN_E2_PROBE
#define RN_E2_PROBE 0
#endif
struct Edge2Args {
  const float *edge_in;
  float *edge_out;
  const float *np3;  // [S*N, 6FP] = node * (Wi | Wj(+bias) | Wk)
  const float *c2;   // [S*E, FP]  the finished c2 embedding of every edge (edge_c2_kernel)
  float *agg_out;    // taped runs: the pre-LayerNorm triplet sums [S*E, FP]; else null
  int S;
  Graph g;
  Dims d;
  PassW<float> w;
};

namespace {
struct Edge2Lds {
  size_t bufQ, bufP, atile, npI, npK, nj, lnp, ints, total;
};
__host__ __device__ inline Edge2Lds edge2_lds(int maxR, int maxD, int maxN) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  Edge2Lds L;
  size_t off = 0;
  L.bufQ = off; off += 2 * up((size_t)maxR * LDQ * 4);   // [2][maxR][LDQ] centred source rows (|q|^2 in column 2FP): this frame | next frame
  L.bufP = off; off += up((size_t)NG * LDQ * 4);         // [16][LDQ] W4 edge_d of a round
  L.atile = off; off += 2 * (size_t)NG * FP * 4;         // 2 x [16][64] swizzled operand rows: destinations | staged sources
  L.npI = off; off += (size_t)NG * 2 * FP * 4;           // [16][2FP] Wi node[b_e] of the staged rows
  L.npK = off; off += (size_t)NG * 2 * FP * 4;           // [16][2FP] Wk node[a_d] of the destinations
  L.nj = off; off += 2 * (size_t)((maxN + 1) & ~1) * 2 * FP * 4;  // [2][maxN rounded up to even][2FP] Wj node[j] + bias: this frame | next frame (requests bring two rows)
  L.lnp = off; off += (size_t)6 * FP * 4;
  L.ints = off; off += up(((size_t)maxR + 6 * (size_t)maxD) * 4);
  L.total = off;
  return L;
}
}  // namespace

template <bool PAD, bool FASTG, bool F16>
__global__ __launch_bounds__(256, 2) void edge_block2_kernel(Edge2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const int maxR = g.max_tile_out_rows, maxD = g.max_tile_in_rows, maxN = g.max_tile_nodes;
  const Edge2Lds L = edge2_lds(maxR, maxD, maxN);
  float *bufQ0 = reinterpret_cast<float *>(smem_raw + L.bufQ);
  const int bufQ_stride = (int)((((size_t)maxR * LDQ * 4 + 15) & ~size_t(15)) / 4);
  const int nj_stride = ((maxN + 1) & ~1) * 2 * FP;
  float *bufP = reinterpret_cast<float *>(smem_raw + L.bufP);
  float *atile = reinterpret_cast<float *>(smem_raw + L.atile);
  float *npI = reinterpret_cast<float *>(smem_raw + L.npI);
  float *npK = reinterpret_cast<float *>(smem_raw + L.npK);
  float *nj0 = reinterpret_cast<float *>(smem_raw + L.nj);
  float *lnp = reinterpret_cast<float *>(smem_raw + L.lnp);
  float *s_c3n2g = lnp, *s_c3n2b = lnp + FP, *s_g3 = lnp + 2 * FP, *s_ig3 = lnp + 4 * FP;
  int *qb = reinterpret_cast<int *>(smem_raw + L.ints);
  int *d_edge = qb + maxR, *d_a = d_edge + maxD, *d_bl = d_a + maxD, *d_rb = d_bl + maxD, *d_cnt = d_rb + maxD,
      *d_skip = d_cnt + maxD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int colbase = wave * 32;
#if RN_FUSED_PRIO
  __builtin_amdgcn_s_setprio(RN_FUSED_PRIO);
#endif
  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.num_tiles;
  const int sg = logical / g.num_tiles, nsg = gridDim.x / g.num_tiles;
  const int j0 = g.tile_begin[tile], j1 = g.tile_begin[tile + 1];
  const int eo0 = g.out_ptr[j0], rows = g.out_ptr[j1] - eo0;
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
  const int nrD = (dcount + NG - 1) / NG, nrS = (rows + NG - 1) / NG;
  const int nrounds = max(max(nrD, nrS), 1);

  // ---- once per launch: LayerNorm parameters and the tile topology -> LDS
  for (int c = tid; c < 2 * FP; c += 256) {
    const float gam = a.w.c3_norm_1.g[c] * (c < FP ? -kLog2e : 2.0f * kLog2e);
    s_g3[c] = gam;
    s_ig3[c] = ((c % FP) < a.d.Fe) ? 1.0f / gam : 0.0f;
    if (c < FP) {
      s_c3n2g[c] = a.w.c3_norm_2.g[c];
      s_c3n2b[c] = a.w.c3_norm_2.b[c];
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

  // ---- weight fragments resident for the whole launch (prescaled: kernels.hpp mfma_prescale)
  const float s4 = F16 ? a.w.mfma_scale[2] : 1.0f, inv4 = F16 ? a.w.mfma_scale[3] : 1.0f;
  const float s5 = F16 ? a.w.mfma_scale[4] : 1.0f, inv5 = F16 ? a.w.mfma_scale[5] : 1.0f;
  WaveB<F16> bW4, bW5;
  bW4.load(a.w.c3_WeT, 4 * FP, colbase, l15, quad, s4);
  bW5.load(a.w.c3_WeT + 2 * FP, 4 * FP, colbase, l15, quad, s5);

  // ---- VALU-phase constants: lane q4 of group grp owns columns 4q4..4q4+3 (+FP)
  const int grp = tid / LG, q4 = tid % LG, c0 = 4 * q4;
  const int nvalid = min(max(a.d.Fe - c0, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fe), invn = 1.0f / (float)a.d.Fe;
  float b3f[4], b3c[4], g3f[4], g3c[4];  // c3_norm_1 with the exp2 scale of the gate folded in
  {
    const Vec4<float> gf = load4<float>(a.w.c3_norm_1.g + c0), bf = load4<float>(a.w.c3_norm_1.b + c0);
    const Vec4<float> gc = load4<float>(a.w.c3_norm_1.g + FP + c0), bc = load4<float>(a.w.c3_norm_1.b + FP + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g3f[i] = -kLog2e * gf.v[i];
      b3f[i] = -kLog2e * bf.v[i];
      g3c[i] = 2.0f * kLog2e * gc.v[i];
      b3c[i] = 2.0f * kLog2e * bc.v[i];
    }
  }
  __syncthreads();

  // ---- LDS-DMA requests.  A frame index outside [0, S) or a round beyond what the tile has: nothing is
  //      requested (the consumers test the same conditions).  All conditions are workgroup-uniform.
  // The rows a round will ask for are named by five table entries per lane; they are read from LDS in one
  // go (one round trip instead of one per request) right after S1.
  struct RoundIdx {
    int e_dst;       // destination edge of operand row 4 wave + quad
    int r_src;       // staged source row (tile-local) of operand row 4 wave + quad
    int b0, b1;      // b_e of the staged rows 4 wave + (lane >> 5) and + 2: rows of np3's Wi block
    int a0, a1;      // a_d of the destinations 4 wave + (lane >> 5) and + 2: rows of np3's Wk block
  };
  auto lookup = [&](int r) {
    RoundIdx x;
    const int row = r * NG + 4 * wave + quad, rw = r * NG + 4 * wave + (lane >> 5);
    const int dl = max(dcount - 1, 0), rl = max(rows - 1, 0);
    x.e_dst = d_edge[min(row, dl)];
    x.r_src = min(row, rl);
    x.b0 = qb[min(rw, rl)];
    x.b1 = qb[min(rw + 2, rl)];
    x.a0 = d_a[min(rw, dl)];
    x.a1 = d_a[min(rw + 2, dl)];
    return x;
  };
  // operand rows of the MFMA phase of round r: destinations of frame sd, staged source rows of frame ss;
  // wave w brings rows 4w..4w+3 of each tile, slot (row, piece p) receives global piece p ^ row
  auto request_tiles = [&](int sd, int ss, int r, const RoundIdx &x) {
    if (RN_E2_PROBE & 16) return;
    const int row = 4 * wave + quad;
    const int piece = (l15 ^ row) & 15;
    if (sd >= 0 && sd < a.S && r < nrD)
      dma16(a.edge_in + ((int64_t)sd * g.E + x.e_dst) * FP + 4 * piece, atile + wave * 256);
    if (ss < a.S && r < nrS)
      dma16(a.edge_in + ((int64_t)ss * g.E + eo0 + x.r_src) * FP + 4 * piece, atile + NG * FP + wave * 256);
  };
  // the 2FP-wide rows of np3 a round's VALU phase adds: wave w brings rows 4w..4w+3 (two per request), which
  // are the rows its own four lane groups read -- except in rounds whose destinations are split over groups
  auto request_npI = [&](int ss, int r, const RoundIdx &x) {  // Wi node[b_e] of the staged rows (np3 columns 0 .. 2FP)
    if (!(ss < a.S && r < nrS) || (RN_E2_PROBE & 16)) return;
    const float *base = a.np3 + (int64_t)ss * g.N * (6 * FP) + 4 * (lane & 31);
    dma16(base + (int64_t)x.b0 * (6 * FP), npI + (4 * wave) * 2 * FP);
    dma16(base + (int64_t)x.b1 * (6 * FP), npI + (4 * wave + 2) * 2 * FP);
  };
  auto request_npK = [&](int sd, int r, const RoundIdx &x) {  // Wk node[a_d] of the destinations (np3 columns 4FP .. 6FP)
    if (!(sd >= 0 && sd < a.S && r < nrD) || (RN_E2_PROBE & 16)) return;
    const float *base = a.np3 + (int64_t)sd * g.N * (6 * FP) + 4 * FP + 4 * (lane & 31);
    dma16(base + (int64_t)x.a0 * (6 * FP), npK + (4 * wave) * 2 * FP);
    dma16(base + (int64_t)x.a1 * (6 * FP), npK + (4 * wave + 2) * 2 * FP);
  };
  auto request_nj = [&](int sd, float *dst) {  // Wj node[j] + bias of the tile's atoms (np3 columns 2FP .. 4FP)
    if (!(sd >= 0 && sd < a.S) || (RN_E2_PROBE & 16)) return;
    for (int n0 = 2 * wave; n0 < j1 - j0; n0 += 8) {  // two atoms per request
      const int n = min(n0 + (lane >> 5), j1 - j0 - 1);
      dma16(a.np3 + ((int64_t)sd * g.N + j0 + n) * (6 * FP) + 2 * FP + 4 * (lane & 31), dst + n0 * 2 * FP);
    }
  };
  // every lane turns the 16-byte operand slots IT fetched into [hi x4 | lo x4] in place (split-f16 path)
  auto split_landed_tiles = [&](bool dests, bool stage) {
    if (RN_E2_PROBE & 16) return;
    if constexpr (F16) {
      float *slot = atile + wave * 256 + lane * 4;
      if (dests) *reinterpret_cast<float4 *>(slot) = split_slot(*reinterpret_cast<const float4 *>(slot));
      if (stage)
        *reinterpret_cast<float4 *>(slot + NG * FP) = split_slot(*reinterpret_cast<const float4 *>(slot + NG * FP));
    }
  };
  // A fragments of one operand tile -> this wave's 32 columns of the transposed product (see WaveB)
  auto tile_product = [&](const float *tl, const WaveB<F16> &bw, f32x4 (&acc)[2]) {
    if constexpr (F16) {
      f16x8 ah[2], al[2];
      load_split_a(tl, l15, quad, ah, al);
      bw.product_split(ah, al, acc);
    } else {
      float af[KS];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4 *>(tl + l15 * FP + (((4 * quad + j) ^ l15) & 15) * 4);
        af[4 * j] = v.x; af[4 * j + 1] = v.y; af[4 * j + 2] = v.z; af[4 * j + 3] = v.w;
      }
      bw.product(af, acc);
    }
  };

  // The frame loop starts one frame early: frame sg - nsg has no destinations, its rounds only stage Q' of
  // frame sg.  par: which half of bufQ / sq / nj holds the frame whose destinations are being served.
  int par = 0;
  {
    const RoundIdx x0 = lookup(0);
    request_tiles(-1, sg, 0, x0);
    request_npI(sg, 0, x0);
  }
  dma_wait();
  split_landed_tiles(false, true);
  __syncthreads();

  for (int s = sg - nsg; s < a.S; s += nsg, par ^= 1) {
    const bool have_cur = s >= 0, have_nxt = s + nsg < a.S;
    const int64_t erow0 = (int64_t)s * g.E;
    float *bufQ = bufQ0 + par * bufQ_stride, *bufQn = bufQ0 + (par ^ 1) * bufQ_stride;
    const float *nj = nj0 + par * nj_stride;

    for (int r = 0; r < nrounds; ++r) {
      const bool do_dest = have_cur && r < nrD, do_stage = have_nxt && r < nrS;
      // ================= MFMA phase: P' of 16 destinations, Q' of 16 source rows of the next frame
      if (do_dest && !(RN_E2_PROBE & 8)) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        tile_product(atile, bW4, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4 *>(bufP + l15 * LDQ + colbase + 16 * t + 4 * quad) = acc[t];
      }
      if (do_stage && !(RN_E2_PROBE & 8)) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        tile_product(atile + NG * FP, bW5, acc);
        if (const int rr = r * NG + l15; rr < rows) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
            *reinterpret_cast<f32x4 *>(bufQn + rr * LDQ + colbase + 16 * t + 4 * quad) = acc[t];
        }
      }
      __syncthreads();  // S1: bufP and the raw Q' rows complete, operand tiles free; npI / npK of this round landed (S2 below)

      // ---- what the NEXT round needs (next round of this frame pair, or round 0 of the next pair)
      const bool wrap = r + 1 == nrounds;
      const int sd_n = wrap ? s + nsg : s, ss_n = wrap ? s + 2 * nsg : s + nsg, r_n = wrap ? 0 : r + 1;
      const RoundIdx xn = lookup(r_n);
      request_tiles(sd_n, ss_n, r_n, xn);
      if (wrap) request_nj(sd_n, nj0 + (par ^ 1) * nj_stride);

      // ================= VALU phase
      // ---- centring of the staged row of this group: add Wi node[b_e], centre, fold the gate scale, |q|^2
      if (const int rr = r * NG + grp; do_stage && rr < rows && !(RN_E2_PROBE & 32)) {
        float *row = bufQn + rr * LDQ;
        const float *np = npI + grp * 2 * FP + c0;
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
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f.v[i] *= g3f[i];
            c.v[i] *= g3c[i];
          }
          ss *= inv2n;
        }
        store4(row + c0, f);
        store4(row + FP + c0, c);
        if (q4 == 0) row[2 * FP] = ss;
      }
      // this wave's groups have consumed their npI rows: the next round's may land (wave-private rows)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      request_npI(ss_n, r_n, xn);

      // ---- destinations of this round: `parts` lane groups share one when at most 8 / 4 / 2 / 1 are left
      const int rem = do_dest ? dcount - r * NG : 0;
      const int parts = (rem > 8 || rem <= 0) ? 1 : (rem > 4 ? 2 : (rem > 2 ? 4 : (rem > 1 ? 8 : 16)));  // uniform over the workgroup
      const int nslots = NG / parts;
      const int slot = grp & (nslots - 1), part = grp / nslots;
      const bool active = slot < rem;
      const int i = r * NG + slot;
      const int64_t drow = active ? erow0 + d_edge[i] : 0;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      float pf[4], pc[4], sp = 0.f;
      Vec4<float> old, c2v;
      if (active) {
        if (part == 0) {
          old = load4<float>(a.edge_in + drow * FP + c0);
          c2v = load4<float>(a.c2 + drow * FP + c0);
        }
        const Vec4<float> xf = load4<float>(bufP + slot * LDQ + c0), xc = load4<float>(bufP + slot * LDQ + FP + c0);
        const Vec4<float> kf = load4<float>(npK + slot * 2 * FP + c0), kc = load4<float>(npK + slot * 2 * FP + FP + c0);
        const Vec4<float> jf = load4<float>(nj + (size_t)d_bl[i] * 2 * FP + c0);
        const Vec4<float> jc = load4<float>(nj + (size_t)d_bl[i] * 2 * FP + FP + c0);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pf[k] = fmaf(xf.v[k], inv4, kf.v[k]) + jf.v[k];
          pc[k] = fmaf(xc.v[k], inv4, kc.v[k]) + jc.v[k];
          sum += pf[k] + pc[k];
        }
        const float mean = lg_sum<LG>(sum) * inv2n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pf[k] = (!PAD || k < nvalid) ? pf[k] - mean : 0.f;
          pc[k] = (!PAD || k < nvalid) ? pc[k] - mean : 0.f;
          sp += pf[k] * pf[k] + pc[k] * pc[k];
        }
        sp = lg_sum<LG>(sp);
      }
      // npK rows are wave-private unless the round is split over groups (then: after the exchange barrier below)
      if (parts == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        request_npK(sd_n, r_n, xn);
      }
      if (active) {
        const int rb = d_rb[i], cnt = (RN_E2_PROBE & 4) ? 0 : d_cnt[i], rskip = d_skip[i];
#if RN_FUSED_PRIO
        __builtin_amdgcn_s_setprio(0);  // the triplet loop is always ready to issue: let the other wave's sparse phases go first
#endif
        const int t0 = (cnt * part) / parts, t1 = (cnt * (part + 1)) / parts;  // this group's triplets
        auto row_of = [&](int tt) { return rb + tt + ((rb + tt >= rskip) ? 1 : 0); };
        if constexpr (FASTG) {
          // pd = p/gamma * (2/2Fe), pg = p*gamma; var + eps = pd.qg + (|p|^2/2Fe + eps) + |q|^2/2Fe
          float pdf[4], pdc[4];
          {
            const Vec4<float> igf = load4<float>(s_ig3 + c0), igc = load4<float>(s_ig3 + FP + c0);
            const float two_inv = 2.0f * inv2n;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              pdf[k] = pf[k] * igf.v[k] * two_inv;
              pdc[k] = pc[k] * igc.v[k] * two_inv;
              pf[k] *= g3f[k];
              pc[k] *= g3c[k];
            }
          }
          const float spe = sp * inv2n + 1e-5f;
          // two columns per instruction (v_pk_add_f32 / v_pk_fma_f32); the exp2 / rcp stay per column
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
          // `qr`: this lane's four filter columns of a source row; the core columns sit FP floats on, |q|^2
          // (one value per row, read by every lane) in the row's pad at column 2FP
          auto triplet = [&](const float *qr, const float *qsq, float (&sumk)[4]) {
            const float4 qfv = *reinterpret_cast<const float4 *>(qr), qcv = *reinterpret_cast<const float4 *>(qr + FP);
            const float qs = *qsq;
            const f32x2 qf2[2] = {{qfv.x, qfv.y}, {qfv.z, qfv.w}}, qc2[2] = {{qcv.x, qcv.y}, {qcv.z, qcv.w}};
            f32x2 d2 = pdf2[0] * qf2[0];
            f32x2 d3 = pdc2[0] * qc2[0];
            d2 = __builtin_elementwise_fma(pdf2[1], qf2[1], d2);
            d3 = __builtin_elementwise_fma(pdc2[1], qc2[1], d3);
            d2 += d3;
            const float dot = lg_sum<LG>(d2.x + d2.y);
            float ve = dot + (spe + qs);
            ve = ve > 1e-5f ? ve : 1e-5f;
            const float rstd = fast_rsq(ve);
            const f32x2 rstd2 = {rstd, rstd}, one2 = {1.0f, 1.0f};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const f32x2 xf2 = __builtin_elementwise_fma(pf2[hh] + qf2[hh], rstd2, bf2[hh]);
              const f32x2 xc2 = __builtin_elementwise_fma(pc2[hh] + qc2[hh], rstd2, bc2[hh]);
              const f32x2 e1 = {fast_exp2(xf2.x), fast_exp2(xf2.y)}, e2 = {fast_exp2(xc2.x), fast_exp2(xc2.y)};
              const f32x2 t2 = e2 + one2;  // (1 + e1)(1 + e2) = t2 + e1 t2: one fma
              const f32x2 den = __builtin_elementwise_fma(e1, t2, t2);
              const f32x2 rd = {fast_rcp(den.x), fast_rcp(den.y)};
              f32x2 sk = {sumk[2 * hh], sumk[2 * hh + 1]};
              sk = __builtin_elementwise_fma(e2 - one2, rd, sk);
              sumk[2 * hh] = sk.x;
              sumk[2 * hh + 1] = sk.y;
            }
          };
          // The triplets of a destination are the rows rb .. of its atom's out-edges without the reverse
          // edge.  The loop keeps one LDS pointer per lane and steps it row by row (two rows where the
          // numbering jumps over the reverse edge) instead of recomputing a row index and an address per
          // triplet; the trip count stays uniform over the lane groups of a wave.  Two independent
          // triplets per iteration, even / odd partial sums.
          float acc2[4] = {0.f, 0.f, 0.f, 0.f};
          const int tskip = rskip - rb;  // triplet number at which the row numbering jumps over the reverse edge
          const int sdelta = 2 * FP - c0;  // from this lane's filter columns of a row to the row's |q|^2
          auto step = [&](const float *p, int tnext) { return p + (tnext == tskip ? 2 * LDQ : LDQ); };
          const float *qr = bufQ + (rb + t0 + (t0 >= tskip ? 1 : 0)) * LDQ + c0;
          int t = t0;
          for (; t + 1 < t1; t += 2) {
            const float *qn = step(qr, t + 1);
            triplet(qr, qr + sdelta, acc);
            triplet(qn, qn + sdelta, acc2);
            qr = step(qn, t + 2);
          }
          if (t < t1) triplet(qr, qr + sdelta, acc);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += acc2[k];
        } else {
          for (int t = t0; t < t1; ++t) {
            const int rq = row_of(t);
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
#if RN_FUSED_PRIO
      __builtin_amdgcn_s_setprio(RN_FUSED_PRIO);
#endif
      if (parts > 1) {  // the other parts reach part 0 through rows nslots .. 15 of bufP (read by no one in a split round)
        if (active && part > 0)
          store4(bufP + (part * nslots + slot) * LDQ + c0, Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}});
        __syncthreads();
        request_npK(sd_n, r_n, xn);  // every group has read its npK row before the barrier
        if (active && part == 0) {
          for (int p = 1; p < parts; ++p) {
            const Vec4<float> other = load4<float>(bufP + (p * nslots + slot) * LDQ + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += other.v[k];
          }
        }
      }
      if (active && part == 0) {
        if (a.agg_out) store4(a.agg_out + drow * FP + c0, Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}});
        Vec4<float> out;
        if (RN_E2_PROBE & 64) {
#pragma unroll
          for (int k = 0; k < 4; ++k) out.v[k] = old.v[k] + c2v.v[k] + acc[k];
        } else {
          const LnParams<float> p3n{load4<float>(s_c3n2g + c0), load4<float>(s_c3n2b + c0)};
          const Vec4<float> c3 = ln_row<LG, PAD>(Vec4<float>{{acc[0], acc[1], acc[2], acc[3]}}, p3n, invn, nvalid);
#pragma unroll
          for (int k = 0; k < 4; ++k) out.v[k] = fast_tanh(old.v[k] + c2v.v[k] + c3.v[k]);
        }
        if (!(RN_E2_PROBE & 1) || out.v[0] == 12345.678f) store4(a.edge_out + drow * FP + c0, out);
      }
      if (!(RN_E2_PROBE & 2)) dma_wait();
      split_landed_tiles(sd_n >= 0 && sd_n < a.S && r_n < nrD, ss_n < a.S && r_n < nrS);
      __syncthreads();  // S2: bufP may be rewritten; the next round's operand tiles and np rows have landed
    }
  }
}

// ---------------------------------------------------------------------------- c2 of every edge
// _EdgeBlock._get_c2_embedding (_gnn.py:200-228):  LayerNorm_Fe(gate(LayerNorm_2Fe(c2_linear(node[b] * node[a]))))
// for all S*E edges, with the UPDATED node embedding.  One wave per 16-row tile: the operand rows (products
// of two gathered node rows) are split in registers, all 128 output columns of the tile come from one wave
// (W fragments of the whole matrix resident: 128 VGPRs), pass through a wave-private LDS slab (a wave's LDS
// operations complete in order: no barrier) into the row-per-lane-group layout, and leave as finished rows.
struct EdgeC2Args {
  const float *node;  // [S*N, FP] updated node embedding
  float *c2;          // [S*E, FP]
  int64_t M;          // S * E
  Graph g;
  Dims d;
  PassW<float> w;
};

template <bool PAD, bool F16>
__global__ __launch_bounds__(256, 2) void edge_c2_kernel(EdgeC2Args a, int tiles_per_wave) {
  __shared__ __attribute__((aligned(16))) float slab_all[4 * NG * LDQ];
  __shared__ __attribute__((aligned(16))) float s_ln[6 * FP];  // c2_norm_1 g | b (2FP each), c2_norm_2 g | b
  __shared__ __attribute__((aligned(16))) float s_bias[2 * FP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  for (int c = tid; c < 2 * FP; c += 256) {
    s_ln[c] = a.w.c2_norm_1.g[c];
    s_ln[2 * FP + c] = a.w.c2_norm_1.b[c];
    s_bias[c] = a.w.c2_bias[c];
    if (c < FP) {
      s_ln[4 * FP + c] = a.w.c2_norm_2.g[c];
      s_ln[5 * FP + c] = a.w.c2_norm_2.b[c];
    }
  }
  const float sc = F16 ? a.w.mfma_scale[6] : 1.0f, inv = F16 ? a.w.mfma_scale[7] : 1.0f;
  WaveB<F16> bw[4];  // columns 32 j .. 32 j + 31
#pragma unroll
  for (int j = 0; j < 4; ++j) bw[j].load(a.w.c2_WT, 2 * FP, 32 * j, l15, quad, sc);
  __syncthreads();
  float *slab = slab_all + wave * NG * LDQ;
  const int q4 = lane & 15, c0 = 4 * q4;
  const int nvalid = min(max(a.d.Fe - c0, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fe), invn = 1.0f / (float)a.d.Fe;
  const LnParams<float> p1f{load4<float>(s_ln + c0), load4<float>(s_ln + 2 * FP + c0)};
  const LnParams<float> p1c{load4<float>(s_ln + FP + c0), load4<float>(s_ln + 3 * FP + c0)};
  const LnParams<float> p2{load4<float>(s_ln + 4 * FP + c0), load4<float>(s_ln + 5 * FP + c0)};
  const Graph &g = a.g;
  const int64_t num_tiles = (a.M + 15) / 16;
  const int64_t first = ((int64_t)blockIdx.x * 4 + wave) * tiles_per_wave;

  float nb[KS], na[KS];  // next tile's operand rows: this lane's 16 k of node[b] and node[a]
  auto fetch = [&](int64_t tl) {
    int64_t row = tl * 16 + l15;
    if (row >= a.M) row = a.M - 1;
    const int64_t s = row / g.E;
    const int e = (int)(row - s * g.E);
    const float *pb = a.node + (s * g.N + g.edge_b[e]) * FP + quad * KS;
    const float *pa = a.node + (s * g.N + g.edge_a[e]) * FP + quad * KS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 x = *reinterpret_cast<const float4 *>(pb + 4 * j), y = *reinterpret_cast<const float4 *>(pa + 4 * j);
      nb[4 * j] = x.x; nb[4 * j + 1] = x.y; nb[4 * j + 2] = x.z; nb[4 * j + 3] = x.w;
      na[4 * j] = y.x; na[4 * j + 1] = y.y; na[4 * j + 2] = y.z; na[4 * j + 3] = y.w;
    }
  };
  if (first < num_tiles) fetch(first);
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int64_t tl = first + it;
    if (tl >= num_tiles) break;
    float af[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) af[k] = nb[k] * na[k];
    if (it + 1 < tiles_per_wave && tl + 1 < num_tiles) fetch(tl + 1);
    if constexpr (F16) {
      f16x8 ah[2], al[2];
      split_f16x8(af, ah[0], al[0]);
      split_f16x8(af + 8, ah[1], al[1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bw[j].product_split(ah, al, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4 *>(slab + l15 * LDQ + 32 * j + 16 * t + 4 * quad) = acc[t];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bw[j].product(af, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4 *>(slab + l15 * LDQ + 32 * j + 16 * t + 4 * quad) = acc[t];
      }
    }
    // four rows per step: lane group `quad` takes row 4 st + quad of the tile
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int rl = 4 * st + quad;
      const float *rp = slab + rl * LDQ;
      Vec4<float> xf = load4<float>(rp + c0), xc = load4<float>(rp + FP + c0);
      const Vec4<float> bf = load4<float>(s_bias + c0), bc = load4<float>(s_bias + FP + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xf.v[k] = fmaf(xf.v[k], inv, bf.v[k]);
        xc.v[k] = fmaf(xc.v[k], inv, bc.v[k]);
      }
      const Vec4<float> g2 = ln_gate<LG, PAD>(xf, xc, p1f, p1c, inv2n, nvalid);
      const Vec4<float> out = ln_row<LG, PAD>(g2, p2, invn, nvalid);
      const int64_t row = tl * 16 + rl;
      if (row < a.M) store4(a.c2 + row * FP + c0, out);
    }
  }
}

size_t edge2_lds_bytes(int rows, int in_rows, int nodes) { return edge2_lds(rows, in_rows, nodes).total; }
size_t edge2_lds_bytes(const Graph &g) { return edge2_lds_bytes(g.max_tile_out_rows, g.max_tile_in_rows, g.max_tile_nodes); }
bool edge2_supported(const Graph &g, Dims d) {
  return d.FnP == 64 && d.FeP == 64 && g.E > 0 && edge2_lds_bytes(g) <= kFusedLdsBudget;
}

void launch_edge_c2(const float *node, float *c2, int S, const Graph &g, Dims d, const PassW<float> &w, bool f16,
                    hipStream_t st) {
  if (S == 0 || g.E == 0) return;
  EdgeC2Args a{node, c2, (int64_t)S * g.E, g, d, w};
  const int64_t tiles = (a.M + 15) / 16;
  // a resident grid (two 4-wave workgroups per CU at 252 VGPRs): every wave loads and splits its 128 VGPRs
  // of weight fragments once and then walks a contiguous run of tiles
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int64_t waves = std::min<int64_t>((int64_t)2 * cus * 4, tiles);
  const int tpw = (int)((tiles + waves - 1) / waves);
  const unsigned blocks = (unsigned)((tiles + (int64_t)4 * tpw - 1) / ((int64_t)4 * tpw));
  const bool pad = d.Fe != d.FeP;
  if (f16) {
    if (pad) edge_c2_kernel<true, true><<<blocks, 256, 0, st>>>(a, tpw);
    else edge_c2_kernel<false, true><<<blocks, 256, 0, st>>>(a, tpw);
  } else {
    if (pad) edge_c2_kernel<true, false><<<blocks, 256, 0, st>>>(a, tpw);
    else edge_c2_kernel<false, false><<<blocks, 256, 0, st>>>(a, tpw);
  }
}

template <bool PAD, bool FASTG, bool F16>
static void launch_cfg2(const Edge2Args &a, size_t lds, hipStream_t st) {
  auto kern = &edge_block2_kernel<PAD, FASTG, F16>;
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

void launch_edge2(const float *edge_in, float *edge_out, const float *np3, const float *c2, float *agg_out, int S,
                  const Graph &g, Dims d, const PassW<float> &w, bool f16, hipStream_t st) {
  if (S == 0 || g.E == 0) return;
  Edge2Args a{edge_in, edge_out, np3, c2, agg_out, S, g, d, w};
  const size_t lds = edge2_lds_bytes(g);
  const bool pad = d.Fe != d.FeP;
  const bool fast = (w.c3_fast & 1) != 0;
  if (f16) {
    if (pad) fast ? launch_cfg2<true, true, true>(a, lds, st) : launch_cfg2<true, false, true>(a, lds, st);
    else fast ? launch_cfg2<false, true, true>(a, lds, st) : launch_cfg2<false, false, true>(a, lds, st);
  } else {
    if (pad) fast ? launch_cfg2<true, true, false>(a, lds, st) : launch_cfg2<true, false, false>(a, lds, st);
    else fast ? launch_cfg2<false, true, false>(a, lds, st) : launch_cfg2<false, false, false>(a, lds, st);
  }
}

}  // namespace rn
