// Fused NodeBlock, atom-owning form (gfx950).  _NodeBlock.forward, /root/reference/ramannoodle/pmodel/torch/_gnn.py:122-151:
//   c1 = c1_linear([node[b_e] | edge_e]) -> LayerNorm(2Fn) -> sigmoid(filter) * tanh(core), summed over the edges e entering
//   each atom (scatter_add, ascending edge order), LayerNorm(Fn), residual tanh.
//
// node_block_fused_kernel (kernels_fused.hip) takes a tile's in-edges sixteen CONSECUTIVE rows at a time; the rows of one
// atom then sit in several lane groups and rounds, so the pre-activations cross the LDS to reach a lane group per row, every
// gated row goes to LDS again and a per-atom pass sums them afterwards -- which bounds the tile at 40 KiB / (272 B per
// row) = 72 rows = 4.5 rounds of which 5 are paid, and costs two barriers and six 16-byte LDS transfers per lane and round.
// Here a tile is SIXTEEN ATOMS and round r takes the r-th in-edge of each of them (operand rows are gathered by edge id
// anyway, so any order costs the same): row l of every round's MFMA tile is atom l of the tile.  Consequences:
//   * each wave multiplies 16 filter columns AND the 16 core columns that gate them (two column tiles of the weight that are
//     64 apart), so a lane's two accumulators hold four (filter, core) pairs of its atom: the gate runs on the accumulators
//     where they are.  The only thing that crosses waves is the row's sum of squares (LayerNorm), 4 B per wave and row;
//   * the per-atom sum is four registers per lane, accumulated in ascending in-edge order -- the reference's order -- with
//     no LDS buffer and no per-atom pass; the atom's node term W_n node + b is the C operand of the round's first MFMAs
//     (the same sixteen rows every round), not an addition per round;
//   * rounds per tile = the largest in-degree among its atoms: no partial round for a crystal of uniform coordination
//     (288 rounds per 256-atom structure of config 3 instead of 320);
//   * LDS is 30 KiB whatever the degree (+ 64 B per round of topology): the operand rows travel through a FOUR-deep ring of
//     LDS-DMA tiles (two to three rounds of requests in flight per workgroup, no registers held), ONE s_barrier per round;
//   * the row has zero mean by construction (c1_linear centred on the host, kernels.hpp: c1_WeT_c) and LayerNorm is
//     invariant under the power-of-two MFMA prescale s (eps scaled by s^2), so the gate phase is sum of squares -> rsqrt ->
//     four gates, with the exp2 factors folded into the LayerNorm parameters.
// Every LDS access inside the round loop is inline assembly (fused_common.hpp) and the barrier is a bare s_barrier behind an
// explicit s_waitcnt: a compiler-visible LDS access or __syncthreads() would drain vmcnt(0), i.e. the whole request ring.
#include "fused_common.hpp"
#include <type_traits>

namespace rn {
namespace {
constexpr int NA_D = 4;  // operand tiles in the LDS ring (requests are issued NA_D - 1 rounds ahead)
static_assert((NA_D & (NA_D - 1)) == 0, "ring slots are addressed with a mask");

struct NodeAtomLds {
  size_t atile, land, part, lnp, tab, total;
};
__host__ __device__ inline NodeAtomLds node_atom_lds(int max_deg) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  NodeAtomLds L;
  size_t off = 0;
  L.atile = off; off += (size_t)NA_D * NG * FP * 4;
  L.land = off; off += (size_t)4 * 256 * 16;  // per lane: the next frame's node term (2 x 16 B), old row of this | the next frame
  L.part = off; off += (size_t)4 * NG * 4 * 4;  // [2] round + [2] epilogue buffers of [16 rows][4 waves] partial sums
  L.lnp = off; off += (size_t)2 * FP * 4;
  L.tab = off; off += up(((size_t)NG * (size_t)(max_deg > 0 ? max_deg : 1) + NG) * 4);
  L.total = off;
  return L;
}

#ifndef RN_NA_PROBE
#define RN_NA_PROBE 0  // timing experiments only (results wrong): 1 no MFMA, 2 no gate, 4 no split
#endif
#ifndef RN_NA_WGS
#define RN_NA_WGS 4  // workgroups per CU the kernel is compiled for (register budget 512 / RN_NA_WGS per lane)
#endif

// sum over the four 16-lane rows of a wave (lanes l, l + 16, l + 32, l + 48), on the VALU
__device__ __forceinline__ float sum_quads(float v) { return sum_xor32(sum_xor16(v)); }
// one wave's part of a row sum -> LDS; after the barrier every lane reads the four parts of its row
__device__ __forceinline__ void part_put(unsigned buf, int l15, int quad, int wave, float v) {
  if (quad == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(buf + (unsigned)(l15 * 4 + wave) * 4u), "v"(v) : "memory");
}
__device__ __forceinline__ float part_get(unsigned buf, int l15) {
  f32x4 p;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p) : "v"(buf + (unsigned)l15 * 16u) : "memory");
  return (p[0] + p[1]) + (p[2] + p[3]);
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace

// PRE: the edge rows arrive as split-f16 pairs (kernels.hpp: launch_geom_rbf_pairs) -- the operand tile is MFMA-ready as it lands
template <bool PAD, bool PRE>
__global__ __launch_bounds__(256, PRE ? RN_NA_WGS : 3) void node_block_atom_kernel(NodeFusedArgs a) {  // (float32 rows: the in-place split needs ~12 more registers)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Graph &g = a.g;
  const NodeAtomLds L = node_atom_lds(g.na_max_deg);
  float *atile = reinterpret_cast<float *>(smem_raw + L.atile);  // [NA_D][16][64] swizzled operand rows, split in place
  float *land = reinterpret_cast<float *>(smem_raw + L.land);    // [4][256] 16-byte slots: what each lane asked for itself
  float *part = reinterpret_cast<float *>(smem_raw + L.part);
  float *lnp = reinterpret_cast<float *>(smem_raw + L.lnp);      // final_norm weight | bias
  int *tab = reinterpret_cast<int *>(smem_raw + L.tab);          // [R][16] edge id of (round, atom); then the 16 degrees

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;
  const int frow = tid >> 4;                 // the row this lane fetches (wave w brings rows 4w .. 4w+3)
  const int fcol = 16 * wave + 4 * quad;     // this lane's filter columns fcol .. fcol+3; its core columns are FP + the same
  const unsigned atile_a = lds_addr(atile), lnp_a = lds_addr(lnp), tab_a = lds_addr(tab), part_a = lds_addr(part);
  const unsigned land_a = lds_addr(land) + (unsigned)tid * 16u;
  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.na_num;
  const int sg = logical / g.na_num, nsg = gridDim.x / g.na_num;
  const int j0 = tile * NG, natoms = min(NG, g.N - j0);

  int *degs = tab + (size_t)NG * max(g.na_max_deg, 1);
  if (tid < NG) degs[tid] = tid < natoms ? g.in_ptr[j0 + tid + 1] - g.in_ptr[j0 + tid] : 0;
  for (int c = tid; c < 2 * FP; c += 256) lnp[c] = c < FP ? a.w.final_norm.g[c] : a.w.final_norm.b[c - FP];
  __syncthreads();
  int R = 0;
  for (int n = 0; n < NG; ++n) R = max(R, degs[n]);
  R = __builtin_amdgcn_readfirstlane(R);
  for (int idx = tid; idx < R * NG; idx += 256) {  // (an atom with fewer in-edges repeats its last one: fetched, not added)
    const int n = idx & 15, r = idx >> 4, dn = degs[n];
    tab[idx] = dn > 0 ? g.in_edge[g.in_ptr[j0 + n] + min(r, dn - 1)] : 0;
  }
  const int deg_own = degs[l15];
  const bool own_valid = l15 < natoms;

  WaveB<true> bW;  // B fragments of the edge part of the centred c1_linear, resident, prescaled by s1
  const float s1 = a.w.mfma_scale_c[6];
  bW.load2(a.w.c1_WeT_c, 2 * FP, 16 * wave, FP + 16 * wave, l15, quad, s1);

  // LayerNorm(2Fn) parameters of this lane's 4 + 4 columns times the gate's exp2 factors:
  //   sigmoid(yf) tanh(yc) = (e2 - 1) / ((1 + e1)(1 + e2)),  e1 = 2^(-log2e yf),  e2 = 2^(2 log2e yc)
  float gfm[4], bfm[4], gcm[4], bcm[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    gfm[k] = -kLog2e * a.w.c1_norm.g[fcol + k];
    bfm[k] = -kLog2e * a.w.c1_norm.b[fcol + k];
    gcm[k] = 2.0f * kLog2e * a.w.c1_norm.g[FP + fcol + k];
    bcm[k] = 2.0f * kLog2e * a.w.c1_norm.b[FP + fcol + k];
  }
  const float kClamp = 15.0f * 2.0f * kLog2e;  // tanh is 1 to 13 digits beyond |yc| = 15; keeps e2 finite
  const int nvalid = min(max(a.d.Fn - fcol, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fn), invn = 1.0f / (float)a.d.Fn;
  const float eps_s = 1e-5f * s1 * s1;  // LayerNorm(s x) with eps s^2 = LayerNorm(x) with eps (s a power of two: exact)
  __syncthreads();
  if (sg >= a.S) return;
  const int nframes = (a.S - sg + nsg - 1) / nsg;

  // this lane's columns of its atom's node term (filter | core) and of the atom's old embedding, frame f of this workgroup
  const int arow = j0 + min(l15, natoms - 1);
  auto seed_ptr = [&](int f, int t) { return a.npc1 + ((int64_t)(sg + f * nsg) * g.N + arow) * (2 * FP) + t * FP + fcol; };
  auto old_ptr = [&](int f) { return a.node_in + ((int64_t)(sg + f * nsg) * g.N + arow) * FP + fcol; };
  f32x4 seed[2];
  {
    const float4 s0 = *reinterpret_cast<const float4 *>(seed_ptr(0, 0)), s1v = *reinterpret_cast<const float4 *>(seed_ptr(0, 1));
    seed[0] = f32x4{s0.x, s0.y, s0.z, s0.w} * s1;
    seed[1] = f32x4{s1v.x, s1v.y, s1v.z, s1v.w} * s1;
    // (consumed before the first request: a later first use would make the compiler drain the requests with it)
    asm volatile("" : "+v"(seed[0]), "+v"(seed[1]));
  }
  // What a frame needs besides its operand rows travels by LDS-DMA into 16-byte slots that only the requesting lane reads
  // (after its own vmcnt wait: no barrier involved): the node term of the NEXT frame, requested during round 0 of this one,
  // and the old rows, which stay in their slot (one per frame parity) until the frame's epilogue.  Registers in flight
  // across the round loop would be the compiler's to copy or spill before the data is there.
  auto request_old = [&](int f) { dma16(old_ptr(f), land + (2 + (f & 1)) * 1024 + wave * 256); };
  auto request_frame = [&](int f) {
    dma16(seed_ptr(f, 0), land + wave * 256);
    dma16(seed_ptr(f, 1), land + 1024 + wave * 256);
    request_old(f);
  };

  // LayerNorm(Fn) of the atom's sum (two passes, as ln_row), residual tanh.  Two barriers, every wave of the workgroup.
  auto epilogue = [&](int f, const f32x4 &sum, const f32x4 *old_given) {
    part_put(part_a + 2 * 256u, l15, quad, wave, sum_quads((sum[0] + sum[1]) + (sum[2] + sum[3])));
    wg_barrier();
    const float mean = part_get(part_a + 2 * 256u, l15) * invn;
    f32x4 dv = sum - mean;
    if (PAD) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i >= nvalid) dv[i] = 0.f;
    }
    part_put(part_a + 3 * 256u, l15, quad, wave, sum_quads((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3])));
    f32x4 fg, fb, oldv;
    lds_read3(lnp_a + (unsigned)fcol * 4u, lnp_a + (unsigned)(FP + fcol) * 4u, land_a + (unsigned)(2 + (f & 1)) * 4096u, fg, fb, oldv);
    if (old_given) oldv = *old_given;
    wg_barrier();
    const float rstd = fast_rsq(part_get(part_a + 3 * 256u, l15) * invn + 1e-5f);
    if (own_valid) {
      float4 out;
      out.x = fast_tanh(oldv[0] + (dv[0] * rstd * fg[0] + fb[0]));
      out.y = fast_tanh(oldv[1] + (dv[1] * rstd * fg[1] + fb[1]));
      out.z = fast_tanh(oldv[2] + (dv[2] * rstd * fg[2] + fb[2]));
      out.w = fast_tanh(oldv[3] + (dv[3] * rstd * fg[3] + fb[3]));
      *reinterpret_cast<float4 *>(a.node_out + ((int64_t)(sg + f * nsg) * g.N + j0 + l15) * FP + fcol) = out;
    }
  };

  if (R == 0) {  // (a tile without in-edges: LayerNorm of a zero sum)
    for (int f = 0; f < nframes; ++f) {
      const float4 o = *reinterpret_cast<const float4 *>(old_ptr(f));
      const f32x4 ov = {o.x, o.y, o.z, o.w};
      epilogue(f, f32x4{0.f, 0.f, 0.f, 0.f}, &ov);
    }
    return;
  }

  // ---- the request ring: step k = (frame, round) in launch order goes to slot k & (NA_D - 1).  Wave w brings rows
  // 4w..4w+3 (lane: row frow, 16-byte piece l15 ^ frow -- the swizzle load_pair_a undoes).  Past the last frame the same
  // request re-fetches rows of the last frame into a slot nobody reads, so that every round is one instruction sequence
  // and the vmcnt arithmetic below holds to the end.
  int ff = 0, fr = 0;
  const int piece = (l15 ^ frow) & 15;
  auto request = [&](int slot) {
    const int e = lds_read1(tab_a + (unsigned)(fr * NG + frow) * 4u);
    const int s = sg + min(ff, nframes - 1) * nsg;
    dma16(a.edge + ((int64_t)s * g.E + e) * FP + 4 * piece, atile + slot * (NG * FP) + wave * 256);
    if (++fr == R) {
      fr = 0;
      ++ff;
    }
  };
  request_old(0);
#pragma unroll
  for (int j = 0; j < NA_D - 1; ++j) request(j);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA_D - 2) : "memory");
  if (!PRE && !(RN_NA_PROBE & 4)) split_own_pair(atile_a, frow, l15);  // this lane's 16 bytes of step 0
  wg_barrier();

  // One round with the ring slot J = k & (NA_D - 1) as a compile-time constant: the loop below is unrolled over the ring, so
  // every LDS address of a round is a loop-invariant register plus an immediate offset.
  unsigned afrag[4];  // this lane's four fragment addresses in ring slot 0
#pragma unroll
  for (int j = 0; j < 4; ++j) afrag[j] = atile_a + (unsigned)l15 * (FP * 4) + (unsigned)(((4 * quad + j) ^ l15) & 15) * 16u;
  int r = 0, f = 0;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  auto round = [&](auto slot_tag) {
    constexpr int J = decltype(slot_tag)::value;
    // ---- matrix phase: 16 filter + 16 core columns of the round's 16 rows, the node term as the C operand
    f32x4 acc[2] = {seed[0], seed[1]};
    if (!(RN_NA_PROBE & 1)) {
      f16x8 ah[2], al[2];
      load_pair_a_at<J * NG * FP * 4>(afrag, ah, al);
      bW.product_split(ah, al, acc);
    }
    // this wave's part of |row|^2 (the row has zero mean: that is the variance's numerator)
    const unsigned pbuf = part_a + (unsigned)(J & 1) * 256u;
    {
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) q = fmaf(acc[0][i], acc[0][i], fmaf(acc[1][i], acc[1][i], q));
      part_put(pbuf, l15, quad, wave, sum_quads(q));
    }
    // ---- the next round's rows: requested NA_D - 1 rounds ago, one younger request may still be in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA_D - 3) : "memory");
    if (!PRE && !(RN_NA_PROBE & 4)) split_own_pair(atile_a + (unsigned)(((J + 1) & (NA_D - 1)) * NG * FP) * 4u, frow, l15);
    wg_barrier();
    request((J + NA_D - 1) & (NA_D - 1));                 // into the slot the previous round multiplied from
    if (r == 0 && f + 1 < nframes) request_frame(f + 1);  // landed by round 2 (the waits above), read after the last
    // ---- gate phase on the accumulators: LayerNorm(2Fn) of a zero-mean row, sigmoid * tanh, sum in edge order
    if (!(RN_NA_PROBE & 2)) {
      const float rstd = fast_rsq(fmaf(part_get(pbuf, l15), inv2n, eps_s));
      f32x4 out;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e1 = fast_exp2(fmaf(acc[0][i] * rstd, gfm[i], bfm[i]));
        const float yc = __builtin_amdgcn_fmed3f(fmaf(acc[1][i] * rstd, gcm[i], bcm[i]), -kClamp, kClamp);
        const float e2 = fast_exp2(yc);
        const float t = 1.0f + e2;
        out[i] = (e2 - 1.0f) * fast_rcp(fmaf(e1, t, t));
      }
      if (r < deg_own) sum += out;
    } else {
      sum += acc[0] + acc[1];
    }
    if (++r < R) return true;
    // ---- the frame's last round: LayerNorm(Fn), residual, store; the next frame's node term out of its LDS slots
    epilogue(f, sum, nullptr);
    sum = f32x4{0.f, 0.f, 0.f, 0.f};
    r = 0;
    if (++f == nframes) return false;
    if (R < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (requested fewer than two waits ago)
    f32x4 n0, n1;
    lds_read2(land_a, land_a + 4096u, n0, n1);
    seed[0] = n0 * s1;
    seed[1] = n1 * s1;
    return true;
  };
  static_assert(NA_D == 4, "the round loop is unrolled over a four-slot ring");
  for (;;) {
    if (!round(std::integral_constant<int, 0>{})) break;
    if (!round(std::integral_constant<int, 1>{})) break;
    if (!round(std::integral_constant<int, 2>{})) break;
    if (!round(std::integral_constant<int, 3>{})) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's re-fetches
}

size_t node_atom_lds_bytes(int max_deg) { return node_atom_lds(max_deg).total; }

void launch_node_atom(const float *edge, const float *node_in, const float *npc1, float *node_out, int S, const Graph &g,
                      Dims d, const PassW<float> &w, hipStream_t st, bool pair_rows) {
  if (S == 0 || g.N == 0) return;
  NodeFusedArgs a{edge, node_in, npc1, node_out, S, g, d, w};
  const bool pad = d.Fn != d.FnP;
  const size_t lds = node_atom_lds_bytes(g.na_max_deg);
  auto kern = pair_rows ? (pad ? &node_block_atom_kernel<true, true> : &node_block_atom_kernel<false, true>)
                        : (pad ? &node_block_atom_kernel<true, false> : &node_block_atom_kernel<false, false>);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, RN_NA_WGS);
  int nsg = per_cu * cus / g.na_num;
  nsg = nsg < 1 ? 1 : (nsg > S ? S : nsg);
  kern<<<(unsigned)nsg * (unsigned)g.na_num, 256, lds, st>>>(a);
}

}  // namespace rn
