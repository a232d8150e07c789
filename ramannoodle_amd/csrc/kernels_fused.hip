// Fused kernels of the 64-wide float32 model that are not the EdgeBlock: the row-ordered NodeBlock (node_block_fused_kernel:
// taped runs and very uneven in-degrees; evaluation takes kernels_node_atom.hip) and the readout MLP in one launch
// (readout_fused_kernel).  gfx950 / wave64.
//
// The per-frame fused EdgeBlock (edge_block_fused_kernel) that lived here through round 4 was RETIRED in round 5: its
// split-f16 instantiation was not bit-reproducible on v_mfma_f32_16x16x32_f16 when two of its workgroups shared a CU
// (profiles/r04/mfma_k32.txt) and the cause was never found.  Its callers -- graphs the role-specialised kernel's ring
// refuses, passes without the folded gate scale, the exact-f32 mode, RN_POTGNN_EDGE_PS=0 / RN_POTGNN_TAPE_PS=0 -- now run
// the unfused chain (projections + edge_agg_kernel, api.hip: edge_unfused_in_blocks).  The kernel is kept, unchanged, in
// experiments/kernels_edge_frame.hip (-DRN_EXPERIMENTS=1 builds only, RN_POTGNN_EDGE_FRAME=1 to select it).
#include "fused_common.hpp"

namespace rn {

// ============================================================================ NodeBlock
// (NodeFusedArgs and its LDS layout: fused_common.hpp)

// RN_NODE_REGRING = D > 0: the operand rows of the next D rounds travel through a ring of D float4 registers per
// lane (plain global loads issued D rounds ahead, split and stored to the operand tile one round ahead) instead
// of one LDS-DMA per round: D x 4 KiB in flight per workgroup instead of 4 KiB.  0 = the LDS-DMA form.
#ifndef RN_NODE_REGRING
#define RN_NODE_REGRING 2
#endif
#ifndef RN_NODE_PROBE
#define RN_NODE_PROBE 0  // timing experiments only (results wrong): 1 no MFMA, 2 no gate, 4 no per-atom phase, 8 no npc1 staging
#endif
template <bool PAD, bool F16, bool ZM = false /* centred c1_linear: zero row mean (kernels.hpp) */>
__global__ __launch_bounds__(256, 4) void node_block_fused_kernel(NodeFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const NodeFusedLds L = node_fused_lds(g.nt_max_in_rows, g.nt_max_nodes);
  float *bufP = reinterpret_cast<float *>(smem_raw + L.bufP);    // [16][LDQ] W_e edge_e of a round
  float *atile = reinterpret_cast<float *>(smem_raw + L.atile);  // [16][64] swizzled operand rows
  float *gated = reinterpret_cast<float *>(smem_raw + L.gated);  // [maxD][LDG] gate outputs of the tile
  float *nj = reinterpret_cast<float *>(smem_raw + L.nj);        // [maxN][2FP] W_n node + bias
  float *lnp = reinterpret_cast<float *>(smem_raw + L.lnp);
  float *s_c1g = lnp, *s_c1b = lnp + 2 * FP, *s_fg = lnp + 4 * FP, *s_fb = lnp + 5 * FP;
  int *d_edge = reinterpret_cast<int *>(smem_raw + L.ints), *d_bl = d_edge + g.nt_max_in_rows;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int colbase = wave * 32;
  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.nt_num;
  const int sg = logical / g.nt_num, nsg = gridDim.x / g.nt_num;
  const int j0 = g.nt_begin[tile], j1 = g.nt_begin[tile + 1];
  const int di0 = g.in_ptr[j0], dcount = g.in_ptr[j1] - di0;
  const int nrounds = (dcount + NG - 1) / NG;

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
  WaveB<F16> bW;  // B fragments of the edge part of c1_linear, resident (prescaled: see the EdgeBlock kernel)
  const float s1 = F16 ? (ZM ? a.w.mfma_scale_c[6] : a.w.mfma_scale[0]) : 1.0f, inv1 = F16 ? (ZM ? a.w.mfma_scale_c[7] : a.w.mfma_scale[1]) : 1.0f;
  bW.load(ZM ? a.w.c1_WeT_c : a.w.c1_WeT, 2 * FP, colbase, l15, quad, s1);

  const int grp = tid / LG, q4 = tid % LG, c0 = 4 * q4;
  const int nvalid = min(max(a.d.Fn - c0, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fn), invn = 1.0f / (float)a.d.Fn;
  __syncthreads();
  // this lane's LayerNorm(2Fn) parameters stay in registers (16 VGPRs): re-reading them every round was a
  // quarter of the kernel's LDS traffic
  const LnParams<float> pf{load4<float>(s_c1g + c0), load4<float>(s_c1b + c0)};
  const LnParams<float> pc{load4<float>(s_c1g + FP + c0), load4<float>(s_c1b + FP + c0)};

  [[maybe_unused]] auto prefetch_round = [&](int s, int r) {
    const int row = 4 * wave + quad;
    const int i = min(r * NG + row, dcount - 1);
    const int piece = (l15 ^ row) & 15;
    dma16(a.edge + ((int64_t)s * g.E + d_edge[i]) * FP + 4 * piece, atile + wave * 256);
  };
  // Split-f16 path: as in the EdgeBlock kernel, every lane splits the slot it fetched itself.
  [[maybe_unused]] auto split_landed_tiles = [&]() {
    if constexpr (F16) {
      float *slot = atile + wave * 256 + lane * 4;
      *reinterpret_cast<float4 *>(slot) = split_slot(*reinterpret_cast<const float4 *>(slot));
    }
  };
#if RN_NODE_REGRING
  constexpr int RING = RN_NODE_REGRING;
  float4 ring[RING];
  int fs = sg, fr = 0;  // (frame, round) the next fetch is for
  auto fetch_next = [&]() {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fs < a.S) {
      const int row = 4 * wave + quad;
      const int i = min(fr * NG + row, dcount - 1);
      const int piece = (l15 ^ row) & 15;
      v = *reinterpret_cast<const float4 *>(a.edge + ((int64_t)fs * g.E + d_edge[i]) * FP + 4 * piece);
      if (++fr == nrounds) {
        fr = 0;
        fs += nsg;
      }
    }
    return v;
  };
  // the ring's head (the next round's rows) -> this lane's slot of the operand tile; refill the tail
  auto publish_next = [&]() {
    float4 v = ring[0];
    if constexpr (F16) v = split_slot(v);
    *reinterpret_cast<float4 *>(atile + tid * 4) = v;
#pragma unroll
    for (int j = 0; j + 1 < RING; ++j) ring[j] = ring[j + 1];
    ring[RING - 1] = fetch_next();
  };
  if (dcount > 0) {
#pragma unroll
    for (int j = 0; j < RING; ++j) ring[j] = fetch_next();
    if (sg < a.S) publish_next();
  }
#else
  if (sg < a.S && dcount > 0) prefetch_round(sg, 0);
#endif

  for (int s = sg; s < a.S; s += nsg) {
    const int64_t nrow0 = (int64_t)s * g.N;
    for (int i = tid; i < ((RN_NODE_PROBE & 8) ? 0 : (j1 - j0) * (2 * FP / 4)); i += 256) {
      const int n = i / (2 * FP / 4), c = (i % (2 * FP / 4)) * 4;
      store4(nj + (size_t)n * 2 * FP + c, load4<float>(a.npc1 + (nrow0 + j0 + n) * (2 * FP) + c));
    }
#if !RN_NODE_REGRING
    dma_wait();
    if (s == sg) split_landed_tiles();  // (later frames: split at the end of the previous frame)
#endif
    __syncthreads();  // the operand rows of round 0 (every wave's share) have landed
    for (int r = 0; r < nrounds; ++r) {
      if (!(RN_NODE_PROBE & 1)) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if constexpr (F16) {
          f16x8 ah[2], al[2];
          load_split_a(atile, l15, quad, ah, al);
          bW.product_split(ah, al, acc);
        } else {
          float af[KS];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(atile + l15 * FP + (((4 * quad + j) ^ l15) & 15) * 4);
            af[4 * j] = v.x; af[4 * j + 1] = v.y; af[4 * j + 2] = v.z; af[4 * j + 3] = v.w;
          }
          bW.product(af, acc);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4 *>(bufP + l15 * LDQ + colbase + 16 * t + 4 * quad) = acc[t];
      }
      if (!(RN_NODE_PROBE & 32))
      __syncthreads();  // S1: bufP complete, operand tile free
#if RN_NODE_REGRING
      if (r + 1 < nrounds || s + nsg < a.S) publish_next();
#else
      if (r + 1 < nrounds) prefetch_round(s, r + 1);
      else if (s + nsg < a.S) prefetch_round(s + nsg, 0);
#endif
#if RN_NODE_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      const int i = r * NG + grp;
      if (i < dcount && !(RN_NODE_PROBE & 2)) {
        const float *njr = nj + (size_t)d_bl[i] * 2 * FP + c0;
        Vec4<float> xf = load4<float>(bufP + grp * LDQ + c0), xc = load4<float>(bufP + grp * LDQ + FP + c0);
        const Vec4<float> af = load4<float>(njr), ac = load4<float>(njr + FP);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xf.v[k] = fmaf(xf.v[k], inv1, af.v[k]);
          xc.v[k] = fmaf(xc.v[k], inv1, ac.v[k]);
        }
        if constexpr (ZM) store4(gated + (size_t)i * LDG + c0, ln_gate_zero_mean<LG>(xf, xc, pf, pc, inv2n));
        else store4(gated + (size_t)i * LDG + c0, ln_gate<LG, PAD>(xf, xc, pf, pc, inv2n, nvalid));
      }
#if RN_NODE_PRIO
      __builtin_amdgcn_s_setprio(RN_NODE_PRIO);
#endif
#if !RN_NODE_REGRING
      dma_wait();
      if (r + 1 < nrounds || s + nsg < a.S) split_landed_tiles();
#endif
      if (!(RN_NODE_PROBE & 16) || r + 1 == nrounds)
      __syncthreads();  // S2: bufP may be rewritten, next round's operand rows landed; after the last round: gated complete
    }
    // ---- per atom: sum over its in-edges (ascending, as the reference's scatter), LayerNorm, residual
    for (int n = grp; n < ((RN_NODE_PROBE & 4) ? 0 : j1 - j0); n += NG) {
      const int i0 = g.in_ptr[j0 + n] - di0, i1 = g.in_ptr[j0 + n + 1] - di0;
      Vec4<float> acc{{0.f, 0.f, 0.f, 0.f}};
      for (int i = i0; i < i1; ++i) {
        const Vec4<float> v = load4<float>(gated + (size_t)i * LDG + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc.v[k] += v.v[k];
      }
      const LnParams<float> pn{load4<float>(s_fg + c0), load4<float>(s_fb + c0)};
      const Vec4<float> ln = ln_row<LG, PAD>(acc, pn, invn, nvalid);
      const Vec4<float> old = load4<float>(a.node_in + (nrow0 + j0 + n) * FP + c0);
      Vec4<float> out;
#pragma unroll
      for (int k = 0; k < 4; ++k) out.v[k] = fast_tanh(old.v[k] + ln.v[k]);
      store4(a.node_out + (nrow0 + j0 + n) * FP + c0, out);
    }
  }
}

size_t node_fused_lds_bytes(const Graph &g) { return node_fused_lds(g.nt_max_in_rows, g.nt_max_nodes).total; }
size_t node_fused_lds_bytes(int tile_in_rows, int tile_nodes) { return node_fused_lds(tile_in_rows, tile_nodes).total; }

void launch_node_fused(const float *edge, const float *node_in, const float *npc1, float *node_out, int S,
                       const Graph &g, Dims d, const PassW<float> &w, bool f16, bool centred, hipStream_t st) {
  if (S == 0 || g.N == 0) return;
  NodeFusedArgs a{edge, node_in, npc1, node_out, S, g, d, w};
  const bool pad = d.Fn != d.FnP;
#if RN_EXPERIMENTS
  if (f16 && node_fused_wave_tiles()) {  // opt-in wave-autonomous form (experiments/)
    launch_node_wave(a, st);
    return;
  }
#endif
  const size_t lds = node_fused_lds_bytes(g);
  auto kern = f16 ? (centred ? (pad ? &node_block_fused_kernel<true, true, true> : &node_block_fused_kernel<false, true, true>)
                             : (pad ? &node_block_fused_kernel<true, true> : &node_block_fused_kernel<false, true>))
                  : (pad ? &node_block_fused_kernel<true, false> : &node_block_fused_kernel<false, false>);
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
  int nsg = per_cu * cus / g.nt_num;
  nsg = nsg < 1 ? 1 : (nsg > S ? S : nsg);
  kern<<<(unsigned)nsg * (unsigned)g.nt_num, 256, lds, st>>>(a);
}

// ============================================================================ readout MLP
// _to_polarizability_embedding (_gnn.py:532-539) in one launch: Linear -> BatchNorm(eval, folded)
// -> ssp -> Linear -> ssp -> Linear(12) per edge row.  The three weight matrices sit in LDS
// transposed; every wave walks its own 16-row tiles: A fragments of the first product straight
// from HBM, the hidden rows pass from the C layout of one product to the A layout of the next
// through a wave-private LDS slab (a wave's LDS operations complete in order: no barrier).
// Replaces three projection launches that wrote and re-read two [S*E, 64] intermediates.
struct ReadoutFusedArgs {
  const float *edge;  // [M, FP]
  float *pol;         // [M, pol_stride]: columns 0..15 written (12 real)
  int64_t M;
  ReadoutW<float> w;
  int pol_stride;     // 32 (the unfused chain's layout) or 16 (whole 64-byte rows)
};

// PRE (with F16): the edge rows are split-f16 pairs (kernels.hpp: launch_geom_rbf_pairs) -- the first layer's operand as fetched
// NW waves per workgroup share one copy of the weights in LDS (41 KiB as split f16) and own a 4.3 KiB slab each: eight waves
// = 77 KiB, two workgroups per CU = four waves per SIMD at 88 VGPRs (four waves per workgroup were two per SIMD: the
// weights' copy per workgroup set the occupancy).
template <bool F16, bool PRE = false, int NW = (F16 ? 8 : 4)>
__global__ __launch_bounds__(64 * NW, 2) void readout_fused_kernel(ReadoutFusedArgs a, int tiles_per_wave) {
  constexpr int LDW = FP + 4;  // floats: row stride of the transposed f32 weights and of the slabs
  constexpr int LDH = FP + 8;  // halves: row stride of the transposed split-f16 weights
  // transposed weights [n][k]: f32 (LDW) or split f16, hi then lo (LDH)
  constexpr int kW64 = F16 ? 2 * FP * LDH * 2 : FP * LDW * 4, kW16 = F16 ? 2 * 16 * LDH * 2 : 16 * LDW * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char ro_smem[];
  unsigned char *w0_raw = ro_smem, *w3_raw = w0_raw + kW64, *w5_raw = w3_raw + kW64;
  float *slab_all = reinterpret_cast<float *>(w5_raw + kW16);
  float *s_scale0 = slab_all + NW * 16 * LDW, *s_shift0 = s_scale0 + FP, *s_b3 = s_shift0 + FP, *s_b5 = s_b3 + FP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  auto put = [&](unsigned char *raw, int rows, int n, int k, float v) {
    if constexpr (F16) {
      _Float16 *h = reinterpret_cast<_Float16 *>(raw), *l = h + rows * LDH;
      const _Float16 hi = (_Float16)v;
      h[n * LDH + k] = hi;
      l[n * LDH + k] = (_Float16)(v - (float)hi);
    } else {
      reinterpret_cast<float *>(raw)[n * LDW + k] = v;
    }
  };
  // split-f16: power-of-two prescaled weights (kernels.hpp: mfma_prescale), inverse folded into the epilogues
  const float ps0 = F16 ? a.w.mfma_scale[0] : 1.0f, inv0 = F16 ? a.w.mfma_scale[1] : 1.0f;
  const float ps3 = F16 ? a.w.mfma_scale[2] : 1.0f, inv3 = F16 ? a.w.mfma_scale[3] : 1.0f;
  const float ps5 = F16 ? a.w.mfma_scale[4] : 1.0f, inv5 = F16 ? a.w.mfma_scale[5] : 1.0f;
  for (int i = tid; i < FP * FP; i += 64 * NW) {
    const int k = i / FP, n = i % FP;  // W?T is [K][N] row-major
    put(w0_raw, FP, n, k, ps0 * a.w.W0T[i]);
    put(w3_raw, FP, n, k, ps3 * a.w.W3T[i]);
  }
  for (int i = tid; i < FP * 16; i += 64 * NW) {
    const int k = i / 16, n = i % 16;
    put(w5_raw, 16, n, k, ps5 * a.w.W5T[k * 32 + n]);
  }
  if (tid < FP) {
    s_scale0[tid] = a.w.scale0[tid] * inv0;
    s_shift0[tid] = a.w.shift0[tid];
    s_b3[tid] = a.w.b3[tid];
    if (tid < 16) s_b5[tid] = a.w.b5[tid];
  }
  __syncthreads();
  float *slab = slab_all + wave * 16 * LDW;
  const int64_t num_tiles = (a.M + 15) / 16;
  const int64_t first = ((int64_t)blockIdx.x * NW + wave) * tiles_per_wave;

  // one 16 x 64 times 64 x (16 NT) product: A in registers (k = 16 quad + s), B rows from LDS
  f16x8 ah[2], al[2];  // split-f16 image of the A rows in `af` (F16 only; refreshed by `operand`)
  auto operand = [&](const float (&af)[KS]) {
    if constexpr (F16) {
      split_f16x8(af, ah[0], al[0]);
      split_f16x8(af + 8, ah[1], al[1]);
    }
  };
  // PRE: this lane's 64 bytes of a pair row (k = 16 quad .. + 15) are [hi x8][lo x8] of the two K = 32 slices
  auto operand_pairs = [&](const float (&af)[KS]) {
    union { float f[4]; f16x8 h; } u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) u[j].f[i] = af[4 * j + i];
    ah[0] = u[0].h; al[0] = u[1].h; ah[1] = u[2].h; al[1] = u[3].h;
  };
  auto product = [&](const float (&af)[KS], const unsigned char *raw, int rows, int nt, f32x4 &acc) {
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (F16) {
      const _Float16 *h = reinterpret_cast<const _Float16 *>(raw) + (nt * 16 + l15) * LDH + quad * KS;
      const _Float16 *l = h + rows * LDH;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        acc = mfma_split3(ah[s], al[s], *reinterpret_cast<const f16x8 *>(h + 8 * s),
                          *reinterpret_cast<const f16x8 *>(l + 8 * s), acc);
    } else {
      const float *wp = reinterpret_cast<const float *>(raw) + (nt * 16 + l15) * LDW + quad * KS;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b = *reinterpret_cast<const float4 *>(wp + 4 * j);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * j], b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * j + 1], b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * j + 2], b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * j + 3], b.w, acc, 0, 0, 0);
      }
    }
  };
  auto load_slab = [&](float (&af)[KS]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = *reinterpret_cast<const float4 *>(slab + l15 * LDW + quad * KS + 4 * j);
      af[4 * j] = v.x; af[4 * j + 1] = v.y; af[4 * j + 2] = v.z; af[4 * j + 3] = v.w;
    }
  };

  float nxt[KS];
  auto fetch = [&](int64_t tile) {
    int64_t row = tile * 16 + l15;
    if (row >= a.M) row = a.M - 1;
    const float *p = a.edge + row * FP + quad * KS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = *reinterpret_cast<const float4 *>(p + 4 * j);
      nxt[4 * j] = v.x; nxt[4 * j + 1] = v.y; nxt[4 * j + 2] = v.z; nxt[4 * j + 3] = v.w;
    }
  };
  if (first < num_tiles) fetch(first);
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int64_t tile = first + it;
    if (tile >= num_tiles) break;
    float af[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) af[k] = nxt[k];
    if (it + 1 < tiles_per_wave && tile + 1 < num_tiles) fetch(tile + 1);
    f32x4 acc;
    // h1 = ssp(BN(edge W0^T))
    if constexpr (PRE) operand_pairs(af);
    else operand(af);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      product(af, w0_raw, FP, nt, acc);
      const int col = nt * 16 + l15;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        slab[(4 * quad + rr) * LDW + col] = ssp_fast(acc[rr] * s_scale0[col] + s_shift0[col]);
    }
    load_slab(af);
    // h2 = ssp(h1 W3^T + b3)
    operand(af);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      product(af, w3_raw, FP, nt, acc);
      const int col = nt * 16 + l15;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) slab[(4 * quad + rr) * LDW + col] = ssp_fast(fmaf(acc[rr], inv3, s_b3[col]));
    }
    load_slab(af);
    // pol = h2 W5^T + b5
    operand(af);
    product(af, w5_raw, 16, 0, acc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int64_t row = tile * 16 + 4 * quad + rr;
      if (row < a.M) a.pol[row * a.pol_stride + l15] = fmaf(acc[rr], inv5, s_b5[l15]);
    }
  }
}

void launch_readout_fused(const float *edge, int64_t M, const ReadoutW<float> &w, float *pol, bool f16,
                          hipStream_t st, bool pair_rows, int pol_stride) {
  if (M == 0) return;
  ReadoutFusedArgs a{edge, pol, M, w, pol_stride};
  const int64_t tiles = (M + 15) / 16;
  const int tpw = 8;
  // dynamic LDS: the three weight copies + one slab per wave + the four small vectors (readout_fused_kernel)
  auto launch = [&](auto kern, int nw, bool f16w) {
    constexpr int LDW = FP + 4, LDH = FP + 8;
    const size_t lds = (f16w ? (size_t)2 * (2 * FP * LDH * 2) + 2 * 16 * LDH * 2 : (size_t)2 * FP * LDW * 4 + 16 * LDW * 4) +
                       (size_t)nw * 16 * LDW * 4 + (3 * FP + 16) * 4;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned blocks = (unsigned)((tiles + (int64_t)nw * tpw - 1) / ((int64_t)nw * tpw));
    kern<<<blocks, 64 * nw, lds, st>>>(a, tpw);
  };
  if (f16 && pair_rows) launch(&readout_fused_kernel<true, true>, 8, true);
  else if (f16) launch(&readout_fused_kernel<true>, 8, true);
  else launch(&readout_fused_kernel<false>, 4, false);
}

size_t edge_fused_lds_bytes(const Graph &g) {
  return fused_lds(g.max_tile_out_rows, g.max_tile_in_rows, g.max_tile_nodes).total;
}
// LDS footprint of a tile with `rows` out-edges, `in_rows` in-edges and `nodes` atoms.
size_t edge_fused_lds_bytes(int rows, int in_rows, int nodes) { return fused_lds(rows, in_rows, nodes).total; }

bool edge_fused_supported(const Graph &g, Dims d) {
  return d.FnP == 64 && d.FeP == 64 && g.E > 0 && edge_fused_lds_bytes(g) <= kFusedLdsBudget;
}

}  // namespace rn
