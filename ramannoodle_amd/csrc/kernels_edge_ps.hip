// Fused EdgeBlock with ROLE-SPECIALISED waves (gfx950, float32 model, Fn and Fe padded to FP = 64, split-f16 MFMA).
// The math is _EdgeBlock.forward (/root/reference/ramannoodle/pmodel/torch/_gnn.py:223-228, 270-291, 351) in the
// factorised form of kernels_fused.hip; what changes is WHO does what.
//
// edge_block_fused_kernel runs four identical waves per workgroup, two workgroups per CU: every wave carries the
// resident weight fragments (64 VGPRs) AND the ~100-register triplet loop, so only two waves fit a SIMD and its
// VALU idles 37 % of the cycles (profiles/r03/perf_sq_counters.txt).  Here ONE 768-thread workgroup owns a CU:
//
//   waves 0-3  PRODUCERS (one per SIMD).  Wave p owns the 32 output columns 32p..32p+31 of every projection with
//              W4, W5 and the c2 weight resident as split-f16 fragments (96 VGPRs).  Per step: retire the LDS-DMA
//              of its operand rows, turn the slots it fetched into [hi|lo] halves, request the next step's rows,
//              then by MFMA  P'_d = W4 edge_d + Wj node[b_d] + Wk node[a_d] + bias  and the c2 pre-activation
//              for the round's 16 destination edges, and  Q'_e = W5 edge_e + Wi node[b_e]  for the 16-row source
//              tiles the coming rounds need.  The node terms seed the accumulators, so an MFMA result IS the row.
//   waves 4-11 CONSUMERS (two per SIMD) in TWO SETS of four (round 5): waves 4-7 take the even rounds, waves 8-11 the odd
//              ones.  A wave owns FOUR destinations of its round, one per 16-lane group with the destination's whole
//              triplet list (lane q of a group: columns 4q..4q+3 of the filter and of the core half): P' row -> registers,
//              the c2 branch (gate(LN(c2 pre-activation)) -> LN, independent of the triplets) right away, then the
//              triplet loop on the ring's rows, LayerNorm, residual tanh, store.  No weights: ~130 VGPRs.
//              (Round 4 gave every destination TWO lane groups of one wave, half of the triplets each, so that all eight
//               waves worked on every round: the per-destination prologue ran twice, the epilogue crossed lane groups,
//               and every round was one hand-over of all twelve waves.  A set now has two round periods for its round,
//               a destination's prologue / epilogue runs once on 16 lanes, and 17 triplets cost 17 loop slots, not 18.)
//
// Three waves per SIMD, and no s_barrier after start-up: the roles meet through LDS words (an LDS atomic add to
// signal, a relaxed poll with s_sleep to wait; a wave's LDS operations are performed in order):
//   c_split / c_norm  the four producers among themselves (operand tiles split; projections written)
//   c_ready           round g may be consumed (published by the LAST producer to finish the step, once it has
//                     completed the new rows' |q|^2 from the four producers' parts); its top bit = "a bounded wait ran
//                     out somewhere in this workgroup": every row stored after that is NaN
//   c_free[g & 1]     consumer waves that have TAKEN round g's P' and c2 rows into registers (right after the round
//                     starts): a producer overwrites those buffers for round g + 2 after all four waves of the set did
//   c_rd[g & 1]       consumer waves that FINISHED round g (no longer read the ring)
// Q' lives in a RING of source-row tiles (PS_NRT x 16 rows): destinations are sorted by their atom and so are the
// source rows, so round g needs a sliding window of rows.  The step of round g writes its new tiles once round
// g - back - 1 is finished (back = Graph::pt_back: 3 when the ring has the room, else 2), i.e. while rounds g - back .. g - 1
// may still be reading: the host checks, by running the same schedule, that no tile those rounds read is overwritten
// then (edge_ps_tile_ok).
//
// No LayerNorm mean anywhere: c3_linear / c2_linear are centred over their output columns on the host
// (api.hip: centred_ops), LN(x) = LN(x - mean x) and mean x is linear in the inputs, so the projections come out
// with zero row mean and the statistics need |p|^2, |q|^2 and p.q only -- which is what lets a producer wave
// finish its 32 columns of a row without seeing the other 96.
#include <algorithm>
#include <vector>

#include "fused_common.hpp"

namespace rn {

struct EdgePsArgs {
  const float *edge_in;
  float *edge_out;
  const float *node;  // updated node embedding [S*N, FP]
  const float *np3;   // [S*N, 6FP] = node * centred(Wi | Wj | Wk) + (0 | centred bias | 0)
  float *agg_out;     // optional: the pre-LayerNorm triplet sums [S*E, FP]
  int *fail;          // set when a bounded wait ran out (a protocol bug; the kernel still terminates)
  int S;
  Graph g;
  Dims d;
  PassW<float> w;
};

// Timing build (RN_BUILD_TAG=timing RN_EXTRA_FLAGS=-DRN_PS_TIMING=1): producer wave 0 and consumer wave 4 of workgroup 0
// accumulate the shader-clock cycles of their phases into fail[16 + 2 i .. ] (int64 pairs: cycles, count); api.hip
// prints them after a synchronising call.  The product build compiles none of it.
#ifndef RN_PS_TIMING
#define RN_PS_TIMING 0
#endif
// Timing-only probe builds (results wrong by construction): -DRN_PS_PROBE=1 consumers skip the triplet loop (what is left is
// the producers' pace), 2 producers skip their MFMA products, 4 consumers skip the epilogue arithmetic, 8 producers skip the
// node-term loads, 16 the operand split, 32 the LDS-DMA requests, 64 the wait of the split sync, 128 the Q' tiles.
#ifndef RN_PS_PROBE
#define RN_PS_PROBE 0
#endif
#ifndef RN_PS_SLEEP
#define RN_PS_SLEEP 1  // s_sleep between two polls of a signalling word (0: poll back to back)
#endif
#ifndef RN_PS_PRIO
#define RN_PS_PRIO 3   // s_setprio of the producer waves (consumers: 0 / RN_PS_LPRIO in the triplet loop, RN_PS_CPRIO outside)
#endif
#ifndef RN_PS_CPRIO
#define RN_PS_CPRIO 2  // s_setprio of a consumer wave outside its triplet loop (+0.6 %: profiles/r04/edge_ps_experiments.txt)
#endif
#ifndef RN_PS_LG8
#define RN_PS_LG8 1     // 1 (not GRAM): EIGHT lanes per destination, eight filter + eight core columns per lane, eight
#endif                  // destinations per wave: what a triplet costs whatever the columns a lane holds -- the row address, the
                        // 16-lane (now 8-lane, three-step) reduction of the cross term, the variance, the rsq -- is paid once
                        // per 16 gates instead of once per 8
#ifndef RN_PS_TICKET
#define RN_PS_TICKET 1  // 1: a consumer wave takes the next QUARTER round (four destinations) off a shared counter instead of
#endif                  // owning every other round with three fixed partners: the wave slots the arbiter serves first simply
                        // take more quarters, and nobody waits for a round while a slower wave still works on the one before
#ifndef RN_PS_LPRIO
#define RN_PS_LPRIO 1  // s_setprio of a consumer wave in the SECOND half of its triplet loop (first half: 0).  The two consumer
#endif                 // waves of a SIMD work on consecutive rounds; at equal priority the arbiter serves the older wave slot
                       // first, so one set ran ahead and then waited for its next round while the other ran alone at half
                       // the issue rate.  The wave that is further into its round -- the one the producers' ring guard waits
                       // for -- now goes first, and the sets leapfrog.
#if RN_PS_TIMING
// (accumulated in wave-uniform locals -- SGPRs -- and written once when the role's loop ends: a read-modify-write of global
//  memory per phase made the producers wait for their own instrumentation, profiles/r05/edge_ps_experiments.txt)
#define PS_T0() long long _t = (long long)__builtin_readcyclecounter(); ++tn
#define PS_TICK(i)                                                                       \
  do {                                                                                   \
    const long long _n = (long long)__builtin_readcyclecounter();                        \
    tl[(i) & 15] += (unsigned)(_n - _t);                                                 \
    _t = (long long)__builtin_readcyclecounter();                                        \
  } while (0)
#define PS_TFLUSH(first, last, shift)                                                    \
  do {                                                                                   \
    if (timed)                                                                           \
      for (int _i = (first); _i <= (last); ++_i) {                                       \
        tacc[2 * (_i + (shift))] += (long long)tl[_i & 15];                              \
        tacc[2 * (_i + (shift)) + 1] += (long long)tn;                                   \
      }                                                                                  \
  } while (0)
#else
#define PS_T0() do {} while (0)
#define PS_TICK(i) do {} while (0)
#define PS_TFLUSH(first, last, shift) do {} while (0)
#endif

namespace {
constexpr int PS_THREADS = 768;
constexpr int PS_PROD = 4;   // producer waves
constexpr int PS_ND = 16;    // destinations per round (two per consumer wave)
#ifndef RN_PS_NRT
#define RN_PS_NRT 8   // (9 fits the CU's LDS too and buys a fifth round of lookahead: level, 6.63-6.66 against 6.65-6.67 ms per launch)
#endif
constexpr int PS_NRT = RN_PS_NRT;  // ring capacity, in 16-row source tiles (GRAM instantiations: PS_NRT_GRAM)
constexpr int PS_NRT_GRAM = 7;  // ... with the Gram tables in LDS next to it: one tile less, slots wrap by compare-and-subtract
constexpr int PS_GW = 3;     // Gram tables: source tiles a round's window may span
constexpr int PS_GSTRIDE = PS_GW * 16 * 16;  // floats of one producer's partial table for one round: [tile][e][d]
constexpr int PS_MAXNEW = 2;             // source tiles a step produces at most
constexpr int PS_TILE = 16 * FP;         // floats of one operand tile
constexpr int PS_BUF = (2 + PS_MAXNEW) * PS_TILE;  // one DMA buffer: edge_d | node[b] | PS_MAXNEW x edge_e

constexpr int PS_CSET = 4;   // consumer waves per set (even rounds: waves 4-7, odd rounds: waves 8-11)
constexpr unsigned PS_FAILBIT = 0x80000000u;  // in the C_READY word: a bounded wait ran out, everything stored from now on is NaN

// byte offsets of the signalling words
enum { C_SPLIT = 0, C_NORM = 16, C_READY = 32, C_FREE0 = 48, C_FREE1 = 64, C_RD0 = 80, C_RD1 = 96, C_TICKET = 112 };

struct PsLds {
  size_t ring, qnp, bufP, bufC, atile, lnp, ints, sync, gram, total;
};
__host__ __device__ inline PsLds ps_lds(int maxR, int maxD, bool gram) {
  auto up = [](size_t b) { return (b + 15) & ~size_t(15); };
  const size_t maxR16 = ((size_t)maxR + 15) & ~size_t(15), rounds = ((size_t)maxD + PS_ND - 1) / PS_ND;
  PsLds L;
  size_t off = 0;
  const size_t ring_rows = (size_t)(gram ? PS_NRT_GRAM : PS_NRT) * 16;
  L.ring = off; off += ring_rows * LDQ * 4;
  L.qnp = off; off += ring_rows * 4 * 4;
  L.bufP = off; off += (size_t)2 * PS_ND * LDQ * 4;
  L.bufC = off; off += (size_t)2 * PS_ND * LDQ * 4;
  L.atile = off; off += ((size_t)2 * PS_BUF + PS_TILE) * 4;  // two buffers + the node[a] tile they share
  // per-column tables: what the producers read every step (the c2 bias, the Q' fold, the P' fold of the cross term) and,
  // where the Gram tables do not need the room, the consumers' LayerNorm / fold tables (eight-lane form: read per quarter)
  L.lnp = off; off += (size_t)(gram ? 6 : 16) * FP * 4;
  L.ints = off; off += up((maxR16 + 7 * (size_t)maxD + 2 * rounds + 8) * 4);
  L.sync = off; off += 128;
  L.gram = off; off += gram ? (size_t)2 * PS_PROD * PS_GSTRIDE * 4 : 0;  // [2 rounds][4 producers][PS_GW tiles][16 e][16 d] partial p.q
  L.total = off;
  return L;
}

// ---- the production schedule (host and device run the same arithmetic) ------------------------------------
// A workgroup serves one atom tile for the frames sg, sg + nsg, ... ("units").  A unit is `nrounds` rounds of 16
// destinations and `nrt` source tiles; hi[r] = last source tile round r reads.  Source tiles are produced in order
// (monotonic index j: unit j / nrt, tile j % nrt) at most PS_MAXNEW per step: a step produces what its own round
// still misses and runs ahead into what the NEXT round needs; tiles round 0 of the first unit needs come from
// destination-less prologue steps.
struct PsStep {
  int has_dest;  // the step serves round (unit u, local round r), global round index g
  int u, r, g;
  int ntiles, tile0;  // source tiles produced: monotonic indices tile0 .. tile0 + ntiles - 1
  int tu0, tt0, tu1, tt1;  // their units and local tile indices (the second pair repeats the first when there is one tile)
  int slot0, slot1;   // the ring slots (16-row tiles) they go to: tile index mod the ring's capacity
  int ubs;            // ring slot of local tile 0 of the step's unit u (the round's window: slots ubs + lo .. ubs + hi, wrapped)
};
struct PsSched {
  int nrounds, nrt, nunits;
  int P, pu, pt;  // tiles scheduled so far; unit and local index of the next one (P = pu nrt + pt)
  int u, r, g;    // next destination round
  int h0, hc;     // hi[0]; hi[r] of the next destination round
  int ring, ps, ubs;  // ring capacity in tiles; slot of the next tile (P mod ring); slot of unit u's local tile 0
};
__host__ __device__ inline int ps_wrap(int x, int ring) { return x >= ring ? x - ring : x; }  // (x < 2 ring)
__host__ __device__ inline void ps_sched_init(PsSched &s, int nrounds, int nrt, int nunits, int hi0, int ring) {
  s.nrounds = nrounds;
  s.nrt = nrt;
  s.nunits = nunits;
  s.P = s.pu = s.pt = s.u = s.r = s.g = 0;
  s.h0 = s.hc = hi0;
  s.ring = ring;
  s.ps = s.ubs = 0;
}
__host__ __device__ inline void ps_sched_take(PsSched &s, PsStep &st, int n) {  // the next n (<= 2) tiles go to this step
  st.ntiles = n;
  st.tile0 = s.P;
  st.tu0 = st.tu1 = s.pu;
  st.tt0 = st.tt1 = s.pt;
  st.slot0 = st.slot1 = s.ps;
  if (n >= 1) {
    ++s.P;
    s.ps = ps_wrap(s.ps + 1, s.ring);
    if (++s.pt == s.nrt) { s.pt = 0; ++s.pu; }
  }
  if (n >= 2) {
    st.tu1 = s.pu;
    st.tt1 = s.pt;
    st.slot1 = s.ps;
    ++s.P;
    s.ps = ps_wrap(s.ps + 1, s.ring);
    if (++s.pt == s.nrt) { s.pt = 0; ++s.pu; }
  }
}
// `hi` = the tile's hi[] table (called at most once, for the round after the one scheduled).  Returns false when the
// stream is exhausted.
template <typename HiFn>
__host__ __device__ inline bool ps_sched_next(PsSched &s, HiFn hi, PsStep &st) {
  if (s.u >= s.nunits) return false;
  if (s.g == 0 && s.P < s.h0 + 1) {  // prologue: what the very first round reads
    const int miss = s.h0 + 1 - s.P;
    st.has_dest = 0;
    st.u = st.r = st.g = 0;
    st.ubs = 0;
    ps_sched_take(s, st, miss < PS_MAXNEW ? miss : PS_MAXNEW);
    return true;
  }
  const int need_now = s.u * s.nrt + s.hc + 1;
  int u2 = s.u, r2 = s.r + 1;
  if (r2 == s.nrounds) { r2 = 0; ++u2; }
  const int h2 = r2 == 0 ? s.h0 : hi(r2);
  const int need_next = u2 < s.nunits ? u2 * s.nrt + h2 + 1 : need_now;
  int n = need_next - s.P;
  n = n < 0 ? 0 : (n > PS_MAXNEW ? PS_MAXNEW : n);
  if (n < need_now - s.P) n = need_now - s.P;  // (edge_ps_tile_ok: never more than PS_MAXNEW)
  st.has_dest = 1;
  st.u = s.u;
  st.r = s.r;
  st.g = s.g;
  st.ubs = s.ubs;
  ps_sched_take(s, st, n);
  if (u2 != s.u) {  // (nrt may exceed the ring: the host check refuses such a tile, the arithmetic stays defined)
    int b = s.ubs + s.nrt % s.ring;
    s.ubs = ps_wrap(b, s.ring);
  }
  s.u = u2;
  s.r = r2;
  s.hc = h2;
  ++s.g;
  return true;
}

// ---- LDS signalling -------------------------------------------------------------------------------------
// In inline assembly on purpose: around a compiler-visible atomic, volatile access or fence hipcc (ROCm 7.2) drains
// vmcnt(0) whenever an LDS-DMA or a store may be outstanding -- for a producer that is the next step's operand rows
// (a full HBM round trip per step), for a consumer the previous round's output stores.  LDS operations of a wave are
// performed in order, and the waits below are on lgkmcnt only.
// Bounded: a wait that has not been satisfied after ~2^21 polls (tens of milliseconds; a real one takes microseconds)
// reports `code` through *fail, sets the top bit of the workgroup's C_READY word -- every wave that stores a row after seeing
// it stores NaN, so an entry point that does not synchronise (and so cannot look at *fail) never hands back plausible numbers
// -- and lets the wave go on: a protocol bug produces an error, not a hung GPU.  Returns the value read (wave-uniform).
__device__ __forceinline__ unsigned ps_wait_ge(unsigned word, unsigned target, unsigned ready_word, int *fail, int code) {
  for (unsigned spins = 0;; ++spins) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(word) : "memory");
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    if (v >= target) return v;
    // (once any wave of the workgroup has given up, the others follow within a thousand polls instead of 2^21 each)
    bool give_up = spins > (1u << 21);
    if ((spins & 1023u) == 1023u && !give_up) {
      unsigned rv;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rv) : "v"(ready_word) : "memory");
      give_up = ((unsigned)__builtin_amdgcn_readfirstlane((int)rv) & PS_FAILBIT) != 0;
    }
    if (give_up) {
      *fail = code;
      // (the store counts in vmcnt like a load: drain, so that no counted wait behind this point returns early)
      asm volatile("ds_or_b32 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(ready_word), "v"(PS_FAILBIT) : "memory");
      return v | PS_FAILBIT;
    }
#if RN_PS_SLEEP
    __builtin_amdgcn_s_sleep(RN_PS_SLEEP);
#endif
  }
}
// every LDS access of this wave is complete before the count moves
__device__ __forceinline__ void ps_arrive(unsigned word, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(word), "v"(1u) : "memory");
}
// the same, returning the count BEFORE this wave's arrival (wave-uniform): the last arriver knows it is the last
__device__ __forceinline__ unsigned ps_arrive_ticket(unsigned word, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned old = 0;
  if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(word), "v"(1u) : "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)old);
}
// eight floats (two f32x4) -> two bf16 halves each: hi = bf16(x) (round to nearest even), lo = bf16(x - hi): 16 significant
// bits over the whole f32 exponent range.  20 VALU: v_cvt_pk_bf16_f32 x 4, the halves back to f32 by shift / mask, packed subtract.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16x8(const f32x4 &x0, const f32x4 &x1, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 v = j < 2 ? f32x2{x0[2 * j], x0[2 * j + 1]} : f32x2{x1[2 * j - 4], x1[2 * j - 3]};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const bf16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), bf16x2);
    hi[2 * j] = h[0];
    hi[2 * j + 1] = h[1];
    lo[2 * j] = l[0];
    lo[2 * j + 1] = l[1];
  }
}
// (an unsigned maximum, not a write: the failure bit of C_READY survives later publications)
__device__ __forceinline__ void ps_publish(unsigned word, unsigned value, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) asm volatile("ds_max_u32 %0, %1" ::"v"(word), "v"(value) : "memory");
}
}  // namespace

// PRE: edge rows in and out are split-f16 pairs (kernels.hpp: launch_geom_rbf_pairs): the producers' edge tiles are MFMA-ready
// as they land, the consumers read / write the (hi, lo) halves of their two columns
// GRAM: the LayerNorm cross term p.q of every (destination, source row) pair of a round on the matrix pipe (the producers'
// Gram phase below) instead of a 128-column dot product and a 16-lane reduction per triplet in the consumers' loop
// F16 = false: every matrix product on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, RN_POTGNN_MFMA=f32 or the range guard's
// fallback): f32 weight fragments, f32 operand tiles as they land, no prescales; PRE and GRAM are then false
template <bool PAD, bool PRE, bool GRAM, bool F16 = true>
__global__ __launch_bounds__(PS_THREADS) void edge_block_ps_kernel(EdgePsArgs a) {
  static_assert(F16 || (!PRE && !GRAM), "the exact-f32 instantiation keeps float32 rows and the in-loop cross term");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NRT = GRAM ? PS_NRT_GRAM : PS_NRT, RING = NRT * 16;
  constexpr bool LG8 = RN_PS_LG8 && !GRAM;  // eight-lane destination groups (below)
  constexpr unsigned CPR = LG8 ? 2u : (unsigned)PS_CSET;  // consumer waves that arrive per round (c_free, c_rd)
  // ring row / tile slot arithmetic: a mask for the power-of-two ring, compare-and-subtract (x < 2 capacity) otherwise
  auto wrap_row = [](unsigned x) -> unsigned {
    if constexpr ((RING & (RING - 1)) == 0) return x & (unsigned)(RING - 1);
    else return min(x, x - (unsigned)RING);  // (unsigned: x - RING wraps to a huge value when x < RING)
  };
  const Graph &g = a.g;
  const PsLds L = ps_lds(g.pt_max_out_rows, g.pt_max_in_rows, GRAM);
  auto wsc = [&](int i) { return F16 ? a.w.mfma_scale_c[i] : 1.0f; };  // the split-f16 products' power-of-two prescale pairs (s, 1 / s)
  float *ring = reinterpret_cast<float *>(smem_raw + L.ring);   // [PS_RING][LDQ] folded source rows, |q|^2 in the pad
  float *qnp = reinterpret_cast<float *>(smem_raw + L.qnp);     // [PS_RING][4] per-producer parts of |q|^2 / 2Fe
  float *bufP = reinterpret_cast<float *>(smem_raw + L.bufP);   // [2][16][LDQ] P' rows of a round
  float *bufC = reinterpret_cast<float *>(smem_raw + L.bufC);   // [2][16][LDQ] c2 pre-activations of a round
  float *atile = reinterpret_cast<float *>(smem_raw + L.atile); // 2 x PS_BUF operand tiles + node[a] tile
  float *na_tile = atile + 2 * PS_BUF;
  float *lnp = reinterpret_cast<float *>(smem_raw + L.lnp);
  // per-column tables the PRODUCERS read every step: the centred c2 bias, the fold of a Q' row on its way to the ring, and
  // the fold 1 / gamma * 2 / 2Fe of a P' row for the cross term (the consumers keep theirs in registers)
  float *s_c2b = lnp, *s_g3q = lnp + 2 * FP, *s_igp = lnp + 4 * FP;
  // (not GRAM) the consumers' tables: p gamma fold, c2_norm_1, c2_norm_2, c3_norm_2
  float *s_g3p = lnp + 6 * FP, *s_c21g = lnp + 8 * FP, *s_c21b = lnp + 10 * FP, *s_c22g = lnp + 12 * FP, *s_c22b = lnp + 13 * FP,
        *s_c3g = lnp + 14 * FP, *s_c3b = lnp + 15 * FP;
  int *ints = reinterpret_cast<int *>(smem_raw + L.ints);
  unsigned *sync = reinterpret_cast<unsigned *>(smem_raw + L.sync);
  const unsigned sync_a = lds_addr(sync);  // LDS byte address of the signalling words
  const unsigned atile_a = lds_addr(atile), lnp_a = lds_addr(lnp), qnp_a = lds_addr(qnp), ints_a = lds_addr(ints);
  const unsigned bufP_a = lds_addr(bufP), ring_a = lds_addr(ring);
  float *gram_f = reinterpret_cast<float *>(smem_raw + L.gram);  // [2 rounds][4 producers][PS_GW tiles][16 e][16 d] partial cross terms
  const int maxD = g.pt_max_in_rows, maxR16 = (g.pt_max_out_rows + 15) & ~15;
  int *qb = ints;  // [maxR16] b_e of the tile's source rows
  int *d_edge = qb + maxR16, *d_a = d_edge + maxD, *d_b = d_a + maxD, *d_rb = d_b + maxD, *d_cnt = d_rb + maxD,
      *d_skip = d_cnt + maxD, *d_re = d_skip + maxD;
  int *hi_t = d_re + maxD;                          // [rounds] last source tile a round reads
  int *lo_t = hi_t + (maxD + PS_ND - 1) / PS_ND;    // [rounds] first source tile a round reads (hi + 1: none)
  int *misc = lo_t + (maxD + PS_ND - 1) / PS_ND;    // [0] nrt
  const int hi_off = maxR16 + 7 * maxD;             // index of hi_t[0] in ints
  const int lo_off = hi_off + (maxD + PS_ND - 1) / PS_ND;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, quad = lane >> 4;

  // Workgroups of one frame group share node / np3 rows: keep them on one XCD (its L2).
  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tile = logical % g.pt_num;
  const int sg = logical / g.pt_num, nsg = gridDim.x / g.pt_num;
  const int j0 = g.pt_begin[tile], j1 = g.pt_begin[tile + 1];
  const int eo0 = g.out_ptr[j0], R = g.out_ptr[j1] - eo0;
  const int di0 = g.in_ptr[j0], D = g.in_ptr[j1] - di0;
  const int nrounds = (D + PS_ND - 1) / PS_ND;
  const int back = g.pt_back;

  // ---- once per launch: LayerNorm parameters, the tile topology, the round -> source tile table
  for (int c = tid; c < 2 * FP; c += PS_THREADS) {
    s_c2b[c] = a.w.c2_bias_c[c] * wsc(4);  // (in the c2 weight's prescale: the accumulator's seed as it is)
    const float gam = a.w.c3_norm_1s.g[c];  // c3_norm_1's scale times the gate's exp2 factor (-log2e | 2 log2e)
    // The producers leave their accumulators in the weights' power-of-two prescale: P' rows carry 1/inv4, Q' rows are
    // multiplied by gamma / s5 as they go to the ring, c2 rows carry 1/invc2 (LayerNorm does not see it: eps scaled)
    s_g3q[c] = gam * wsc(3);
    // pd = p / gamma * (2 / 2Fe) from a stored P' row: pd . (q gamma) is the cross term of the variance
    s_igp[c] = ((c % FP) < a.d.Fe) ? wsc(1) / gam * (1.0f / (float)a.d.Fe) : 0.0f;
    if constexpr (!GRAM) {
      s_g3p[c] = gam * wsc(1);
      s_c21g[c] = a.w.c2_norm_1.g[c];
      s_c21b[c] = a.w.c2_norm_1.b[c];
      if (c < FP) {
        s_c22g[c] = a.w.c2_norm_2.g[c];
        s_c22b[c] = a.w.c2_norm_2.b[c];
        s_c3g[c] = a.w.c3_norm_2.g[c];
        s_c3b[c] = a.w.c3_norm_2.b[c];
      }
    }
  }
  for (int r = tid; r < maxR16; r += PS_THREADS) qb[r] = g.edge_b[eo0 + min(r, max(R - 1, 0))];
  for (int i = tid; i < D; i += PS_THREADS) {
    const int dst = g.in_edge[di0 + i];
    const int ad = g.edge_a[dst], bd = g.edge_b[dst];
    const int rb = g.out_ptr[bd] - eo0, re = g.out_ptr[bd + 1] - eo0;
    const int rev = g.rev_edge[dst];  // edge (b_d -> a_d): its triplet (i == k) is excluded
    d_edge[i] = dst;
    d_a[i] = ad;
    d_b[i] = bd;
    d_rb[i] = rb;
    d_cnt[i] = (re - rb) - (rev >= 0 ? 1 : 0);
    d_skip[i] = rev >= 0 ? rev - eo0 : re;
    d_re[i] = re;
  }
  if (tid < 32) sync[tid] = 0u;
  __syncthreads();
  if (tid < nrounds) {  // destinations are sorted by atom, so the rows a round reads end where its last atom's do
    int top = 0, first = 1 << 30;
    const int last = min((tid + 1) * PS_ND, D);
    for (int i = 0; i < last; ++i) {
      top = max(top, d_re[i]);
      if (i >= tid * PS_ND && d_re[i] > d_rb[i]) first = min(first, d_rb[i]);
    }
    hi_t[tid] = (top + 15) / 16 - 1;
    lo_t[tid] = first == (1 << 30) ? (top + 15) / 16 : first / 16;
  }
  __syncthreads();
  if (tid == 0) misc[0] = max(hi_t[nrounds > 0 ? nrounds - 1 : 0] + 1, 1);
  __syncthreads();
  if (D == 0 || sg >= a.S) return;
  const int nrt = __builtin_amdgcn_readfirstlane(misc[0]);
  const int nunits = (a.S - sg + nsg - 1) / nsg;

  if (wave < PS_PROD) {
    // =========================================================================================== PRODUCER
    __builtin_amdgcn_s_setprio(RN_PS_PRIO);
    const int colbase = wave * 32;  // this wave's 32 of the 128 pre-activation columns
    // (uniform values: kept in SGPRs -- as VGPR operands of packed multiplies each would cost a register pair)
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const float s4 = uni(wsc(0));
    const float s5 = uni(wsc(2)), inv5 = uni(wsc(3));
    const float sc2 = uni(wsc(4));
    WaveB<F16> bW4, bW5, bWc;
    bW4.load(a.w.c3_WeT_c, 4 * FP, colbase, l15, quad, s4);
    bW5.load(a.w.c3_WeT_c + 2 * FP, 4 * FP, colbase, l15, quad, s5);
    bWc.load(a.w.c2_WT_c, 2 * FP, colbase, l15, quad, sc2);
    const float qscale = uni(inv5 * inv5 / (float)(2 * a.d.Fe));  // |q|^2 / 2Fe from the prescaled accumulators

    // LDS-DMA of a step's operand rows: wave w brings rows 4w..4w+3 of every tile; slot (row, piece p) receives
    // global piece p ^ row (the XOR swizzle load_split_a undoes).  Always the same five requests -- a step without
    // destinations or with fewer source tiles re-fetches a valid row into a tile nobody reads -- so that every
    // step is one instruction sequence.
    auto request = [&](const PsStep &st, int buf, int ln) {
      const int row = 4 * wave + (ln >> 4);
      const int piece = ((ln & 15) ^ row) & 15;
      float *dst = atile + buf * PS_BUF + wave * 256;
      // (indices first, through ONE assembly read: an LDS read the compiler can see between two requests would get a
      //  vmcnt(0) in front of it and serialise them)
      const unsigned ia = ints_a + (unsigned)(maxR16 + min(st.r * PS_ND + row, D - 1)) * 4u;  // &d_edge[i]
      int de, da, db;
      lds_read1x3(ia, ia + (unsigned)maxD * 4u, ia + 2u * (unsigned)maxD * 4u, de, da, db);
      const int s = sg + st.u * nsg;
      const float *eb = a.edge_in + (int64_t)s * g.E * FP, *nb = a.node + (int64_t)s * g.N * FP;  // (uniform)
      const float *eb0 = a.edge_in + (int64_t)(sg + min(st.tu0, nunits - 1) * nsg) * g.E * FP;
      const float *eb1 = a.edge_in + (int64_t)(sg + min(st.tu1, nunits - 1) * nsg) * g.E * FP;
      const unsigned r0 = (unsigned)(eo0 + min(st.tt0 * 16 + row, R - 1)), r1 = (unsigned)(eo0 + min(st.tt1 * 16 + row, R - 1));
      dma16(eb + ((unsigned)de * FP + 4 * piece), dst);
      dma16(nb + ((unsigned)db * FP + 4 * piece), dst + PS_TILE);
      dma16(nb + ((unsigned)da * FP + 4 * piece), na_tile + wave * 256);
      dma16(eb0 + (r0 * FP + 4 * piece), dst + 2 * PS_TILE);
      dma16(eb1 + (r1 * FP + 4 * piece), dst + 3 * PS_TILE);
    };
    // every lane turns the 16-byte slots IT fetched into its shares of the [hi x8] / [lo x8] slot pairs (fused_common.hpp:
    // write_pair) in place: the edge rows, and
    // node[b] * node[a] (the c2 operand) in the node[b] tile
    auto split_landed = [&](int buf, int ln) {
      const int row = 4 * wave + (ln >> 4), phys = ln & 15;
      const unsigned tb = atile_a + (unsigned)(buf * PS_BUF) * 4u;
      const unsigned sa = tb + (unsigned)(wave * 256 + ln * 4) * 4u;
      f32x4 u, v, w;
      if constexpr (!PRE && F16) {
        lds_read3(sa, sa + 2 * PS_TILE * 4, sa + 3 * PS_TILE * 4, u, v, w);
        write_pair(tb, row, phys, u);
        write_pair(tb + 2 * PS_TILE * 4, row, phys, v);
        write_pair(tb + 3 * PS_TILE * 4, row, phys, w);
      }
      lds_read2(sa + PS_TILE * 4, atile_a + (unsigned)(2 * PS_BUF + wave * 256 + ln * 4) * 4u, u, v);
      if constexpr (F16) {
        write_pair(tb + PS_TILE * 4, row, phys, u * v);
      } else {  // (float32 tiles: the product replaces node[b]'s slot as it is)
        const f32x4 pr = u * v;
        lds_write4(sa + PS_TILE * 4, float4{pr[0], pr[1], pr[2], pr[3]});
      }
    };
    static_assert(PS_MAXNEW == 2, "split_landed / request are written for two source tiles per step");

    auto hi = [&](int r) { return __builtin_amdgcn_readfirstlane(lds_read1(ints_a + (unsigned)(hi_off + r) * 4u)); };  // hi_t[r]
    PsSched sched;
    ps_sched_init(sched, nrounds, nrt, nunits, __builtin_amdgcn_readfirstlane(hi_t[0]), NRT);
    PsStep cur, nxt;
    bool have = ps_sched_next(sched, hi, cur);
    if (have) request(cur, 0, lane);
#if RN_PS_TIMING
    long long *tacc = reinterpret_cast<long long *>(a.fail + 16);
    const bool timed = blockIdx.x == 0 && wave == 0 && lane == 0;
    unsigned tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tn = 0;
#endif
    for (unsigned k = 0; have; ++k) {
      PS_T0();
      const int buf = (int)(k & 1u);
      int ln = launder(lane);
      // ---- A: this step's operand rows have landed (requested one step ago)
      dma_wait();
      PS_TICK(0);
      // node terms of this step's destinations: four 16-byte loads per lane, issued now and added to the product where that is
      // finished (`seeds_landed` below) -- in flight under the split, the sync, the next requests and the first MFMAs; the
      // source tile's two follow when the P' registers are free, in flight under the c2 and Q' products.
      // Inline assembly: the compiler would wait for a load it can see with vmcnt(0) at its first use, i.e. for the five
      // LDS-DMA requests issued after it as well; in order, "all but the five youngest" is exactly these loads.
      f32x4 kP[2], jP[2], qS[2];
      const float *q_src;  // this lane's node term of the step's first source tile: requested when the P' registers are free
      int ringrow0;
      {
        const int l15 = ln & 15, mycol = colbase + 4 * (ln >> 4);
        const float *np3_s = a.np3 + (int64_t)(sg + cur.u * nsg) * g.N * (6 * FP);  // (uniform) this step's frame
        const int i = min(cur.r * PS_ND + l15, D - 1);
        const unsigned ok = (unsigned)d_a[i] * (6 * FP) + 4 * FP + mycol, oj = (unsigned)d_b[i] * (6 * FP) + 2 * FP + mycol;
        const float *np3_q = a.np3 + (int64_t)(sg + min(cur.tu0, nunits - 1) * nsg) * g.N * (6 * FP);
        const unsigned oq = (unsigned)qb[min(cur.tt0 * 16 + l15, R - 1)] * (6 * FP) + mycol;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (RN_PS_PROBE & 8) {
            kP[t] = jP[t] = f32x4{0.f, 0.f, 0.f, 0.f} + (float)(ok + oj);
          } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kP[t]) : "v"(np3_s + ok + 16 * t) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(jP[t]) : "v"(np3_s + oj + 16 * t) : "memory");
          }
        }
        q_src = np3_q + oq;
        ringrow0 = cur.slot0 * 16 + l15;
      }
      const bool have_next = ps_sched_next(sched, hi, nxt);  // (its one table read hides under the loads above)
      if (!(RN_PS_PROBE & 16)) split_landed(buf, ln);
      PS_TICK(1);
      ps_arrive(sync_a + C_SPLIT, ln);
      if (!(RN_PS_PROBE & 64)) ps_wait_ge(sync_a + C_SPLIT, 4u * (k + 1u), sync_a + C_READY, a.fail, 1);
      PS_TICK(2);
      // ---- B: request the next step's rows (its buffer and the node[a] tile are free: every producer is past step k - 1)
      PS_TICK(3);
      ln = launder(ln);
      // Always five requests, also after the last step (its own rows again, into the buffer nobody will read): the wait below
      // is then ONE statement on one path.  With two variants behind a branch the compiler copied the loaded registers into
      // the merge point's registers in front of one of the waits -- i.e. before the data was there.
      if (!(RN_PS_PROBE & 32)) request(have_next ? nxt : cur, buf ^ 1, ln);
      // the node terms are there once at most the five requests above are outstanding
      // (the product's accumulators are operands too: the wait stays BEHIND the MFMAs it is meant to be covered by)
      auto seeds_landed = [&](f32x4 (&acc)[2]) {
        if (RN_PS_PROBE & 32)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(kP[0]), "+v"(kP[1]), "+v"(jP[0]), "+v"(jP[1]), "+v"(acc[0]), "+v"(acc[1])::"memory");
        else
          asm volatile("s_waitcnt vmcnt(5)" : "+v"(kP[0]), "+v"(kP[1]), "+v"(jP[0]), "+v"(jP[1]), "+v"(acc[0]), "+v"(acc[1])::"memory");
      };
      // Every load issued in assembly is waited for on EVERY path, by a statement that names its registers: a register that
      // is loaded and then never read (a step without destinations, a step without a source tile) would be free for the
      // compiler to hand to something else while the load is still in flight.  So the waits are unconditional and only the
      // arithmetic around them is not.
      PS_TICK(4);
      // ---- the P' / c2 buffers of round g were last read by round g - 2: its set has taken the rows into registers
      if (cur.has_dest)
        ps_wait_ge(sync_a + ((cur.g & 1) ? C_FREE1 : C_FREE0), CPR * (unsigned)(cur.g >> 1), sync_a + C_READY, a.fail, 2);
      PS_TICK(5);
      ln = launder(ln);
      const int l15 = ln & 15, quad = ln >> 4, mycol = colbase + 4 * quad;  // + 16 t: the four columns of tile t this lane ends up with
      const unsigned tb_a = atile_a + (unsigned)(buf * PS_BUF) * 4u;
      const int slot0 = (cur.g & 1) * PS_ND;
      {
        f32x4 accP[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (cur.has_dest) {
          if constexpr (F16) {
            f16x8 ah[2], al[2];
            load_pair_a(tb_a, l15, quad, ah, al);
            if (!(RN_PS_PROBE & 2)) bW4.product_split(ah, al, accP);
          } else {
            float af[KS];
            f32x4 unused0, unused1;
            load_f32_a2(tb_a, l15, quad, af, lnp_a, lnp_a, unused0, unused1);
            if (!(RN_PS_PROBE & 2)) bW4.product(af, accP);
          }
        }
        seeds_landed(accP);
        if (cur.has_dest) {
#pragma unroll
          for (int t = 0; t < 2; ++t)  // row l15, columns mycol + 16 t .. + 3; the node terms in the weights' prescale
            *reinterpret_cast<f32x4 *>(bufP + (slot0 + l15) * LDQ + mycol + 16 * t) = (kP[t] + jP[t]) * s4 + accP[t];
        }
      }
      // the first source tile's node term: requested now that the P' registers are free, in flight under the c2 and Q' products
      if (RN_PS_PROBE & 8) {
        qS[0] = qS[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qS[0]) : "v"(q_src) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qS[1]) : "v"(q_src + 16) : "memory");
      }
      if (cur.has_dest) {
        f32x4 accC[2];
        if constexpr (F16) {
          f16x8 ah[2], al[2];
          load_pair_a2(tb_a + PS_TILE * 4, l15, quad, ah, al, lnp_a + (unsigned)mycol * 4u, lnp_a + (unsigned)(mycol + 16) * 4u,
                       accC[0], accC[1]);  // + the centred c2 bias
          if (!(RN_PS_PROBE & 2)) bWc.product_split(ah, al, accC);
        } else {
          float af[KS];
          load_f32_a2(tb_a + PS_TILE * 4, l15, quad, af, lnp_a + (unsigned)mycol * 4u, lnp_a + (unsigned)(mycol + 16) * 4u, accC[0], accC[1]);
          if (!(RN_PS_PROBE & 2)) bWc.product(af, accC);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<f32x4 *>(bufC + (slot0 + l15) * LDQ + mycol + 16 * t) = accC[t];
      }
      PS_TICK(6);
      const int ntl = (RN_PS_PROBE & 128) ? 0 : cur.ntiles;
      // source tile n of the step: product, node term, |q|^2 part, scaled row to the ring
      auto q_tile_finish = [&](f32x4 (&accQ)[2], const f32x4 (&g3v)[2], int ringrow) {
#pragma unroll
        for (int t = 0; t < 2; ++t) accQ[t] = qS[t] * s5 + accQ[t];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) ss = fmaf(accQ[t][e], accQ[t][e], ss);
        ss = sum_xor32(sum_xor16(ss));  // the row's four lanes l15 + 16 quad
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4 *>(ring + ringrow * LDQ + mycol + 16 * t) = accQ[t] * g3v[t];
        if (quad == 0) qnp[ringrow * 4 + wave] = ss * qscale;
      };
      {
        f32x4 accQ[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, g3v[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (ntl > 0) {
          if constexpr (F16) {
            f16x8 ah[2], al[2];
            load_pair_a2(tb_a + (unsigned)(2 * PS_TILE) * 4u, l15, quad, ah, al, lnp_a + (unsigned)(2 * FP + mycol) * 4u,
                         lnp_a + (unsigned)(2 * FP + mycol + 16) * 4u, g3v[0], g3v[1]);  // + s_g3q
            if (!(RN_PS_PROBE & 2)) bW5.product_split(ah, al, accQ);
          } else {
            float af[KS];
            load_f32_a2(tb_a + (unsigned)(2 * PS_TILE) * 4u, l15, quad, af, lnp_a + (unsigned)(2 * FP + mycol) * 4u,
                        lnp_a + (unsigned)(2 * FP + mycol + 16) * 4u, g3v[0], g3v[1]);
            if (!(RN_PS_PROBE & 2)) bW5.product(af, accQ);
          }
        }
        // (issued after this step's requests: everything of this wave has landed then -- the requests are ~2000 cycles old)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qS[0]), "+v"(qS[1]), "+v"(accQ[0]), "+v"(accQ[1])::"memory");
        PS_TICK(7);
        // ---- the ring: this step's tiles overwrite rows that rounds <= g - back - 1 read (edge_ps_tile_ok) -- that round must
        // be done.  Behind the product: hidden under it.
        if (cur.has_dest && cur.g > back) {
          const int gdone = cur.g - back - 1;  // (the rounds of its parity up to it: (gdone >> 1) + 1)
          ps_wait_ge(sync_a + ((gdone & 1) ? C_RD1 : C_RD0), CPR * (unsigned)((gdone >> 1) + 1), sync_a + C_READY, a.fail, 3);
        }
        PS_TICK(9);
        if (ntl > 0) q_tile_finish(accQ, g3v, ringrow0);
      }
      if (ntl > 1) {  // a second tile in one step is rare (once per unit): its node terms are fetched here, by ordinary loads
        const float *np3_q = a.np3 + (int64_t)(sg + min(cur.tu1, nunits - 1) * nsg) * g.N * (6 * FP);
        const unsigned oq = (unsigned)lds_read1(ints_a + (unsigned)min(cur.tt1 * 16 + l15, R - 1) * 4u) * (6 * FP) + mycol;  // qb[]
#pragma unroll
        for (int t = 0; t < 2; ++t) qS[t] = *reinterpret_cast<const f32x4 *>(np3_q + oq + 16 * t);
        f32x4 accQ[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, g3v[2];
        if constexpr (F16) {
          f16x8 ah[2], al[2];
          load_pair_a2(tb_a + (unsigned)(3 * PS_TILE) * 4u, l15, quad, ah, al, lnp_a + (unsigned)(2 * FP + mycol) * 4u,
                       lnp_a + (unsigned)(2 * FP + mycol + 16) * 4u, g3v[0], g3v[1]);
          if (!(RN_PS_PROBE & 2)) bW5.product_split(ah, al, accQ);
        } else {
          float af[KS];
          load_f32_a2(tb_a + (unsigned)(3 * PS_TILE) * 4u, l15, quad, af, lnp_a + (unsigned)(2 * FP + mycol) * 4u,
                      lnp_a + (unsigned)(2 * FP + mycol + 16) * 4u, g3v[0], g3v[1]);
          if (!(RN_PS_PROBE & 2)) bW5.product(af, accQ);
        }
        q_tile_finish(accQ, g3v, cur.slot1 * 16 + l15);
      }
      if constexpr (GRAM) {
        // ---- the Gram phase: p.q of the round's 16 destinations with every source row of the round's window, this wave's 32
        // columns of it, on the matrix pipe.  Operands: the P' rows this wave stored above (its own LDS stores: in order) times
        // 1 / gamma * 2 / 2Fe, and the folded Q' rows (q gamma) of the window's tiles out of the ring -- this wave's columns of
        // them are its own stores too, of this step or of earlier ones, so the phase needs no other wave.  Both are in the
        // model's own units (no bounded range), so they are split into two bf16 halves each (16 significant bits, the f32
        // exponent range) and multiplied as lo hi + hi lo + hi hi with f32 accumulation: the cross term to ~2^-16 of
        // sum |p_c q_c|, a few 1e-7 of the variance it enters.  The lane ends with (e = row l15 of the tile, d = 4 quad .. + 3):
        // one 16-byte store into this producer's partial table; the consumers add the four tables (their prologue).
        if (cur.has_dest) {
          const unsigned prow_a = bufP_a + (unsigned)((slot0 + l15) * LDQ + mycol) * 4u;
          const unsigned ig_a = lnp_a + (unsigned)(4 * FP + mycol) * 4u;
          const int lo = __builtin_amdgcn_readfirstlane(lds_read1(ints_a + (unsigned)(lo_off + cur.r) * 4u));
          const int hiw = __builtin_amdgcn_readfirstlane(lds_read1(ints_a + (unsigned)(hi_off + cur.r) * 4u));
          // always PS_GW tiles, as one straight instruction sequence: a window of fewer tiles multiplies its last one again
          // (into a table slot nobody reads); every operand row of the phase is requested before the first is waited for
          unsigned qa[PS_GW];
#pragma unroll
          for (int k = 0; k < PS_GW; ++k) {
            int sl = cur.ubs + min(lo + k, max(hiw, lo));  // (< 3 NRT: edge_ps_tile_ok)
            sl = sl >= NRT ? sl - NRT : sl;
            sl = sl >= NRT ? sl - NRT : sl;
            qa[k] = ring_a + (unsigned)((sl * 16 + l15) * LDQ + mycol) * 4u;
          }
          static_assert(PS_GW == 3, "the Gram phase is written for three window tiles");
          f32x4 p0, p1, i0, i1, q0[PS_GW], q1[PS_GW];
          asm volatile(
              "ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:64\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %11 offset:64\n\t"
              "ds_read_b128 %4, %12\n\tds_read_b128 %5, %12 offset:64\n\tds_read_b128 %6, %13\n\tds_read_b128 %7, %13 offset:64\n\t"
              "ds_read_b128 %8, %14\n\tds_read_b128 %9, %14 offset:64\n\ts_waitcnt lgkmcnt(0)"
              : "=&v"(p0), "=&v"(p1), "=&v"(i0), "=&v"(i1), "=&v"(q0[0]), "=&v"(q1[0]), "=&v"(q0[1]), "=&v"(q1[1]), "=&v"(q0[2]), "=&v"(q1[2])
              : "v"(prow_a), "v"(ig_a), "v"(qa[0]), "v"(qa[1]), "v"(qa[2])
              : "memory");
          bf16x8 ph, pl, qh[PS_GW], ql[PS_GW];
          split_bf16x8(p0 * i0, p1 * i1, ph, pl);
#pragma unroll
          for (int k = 0; k < PS_GW; ++k) split_bf16x8(q0[k], q1[k], qh[k], ql[k]);
          f32x4 gacc[PS_GW];
#pragma unroll
          for (int k = 0; k < PS_GW; ++k) gacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, qh[k], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < PS_GW; ++k) gacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, ql[k], gacc[k], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < PS_GW; ++k) gacc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, qh[k], gacc[k], 0, 0, 0);
          // (plain stores, like the P' / c2 / ring rows: the compiler then covers the MFMA -> LDS-write hazard itself; an asm
          //  ds_write of the accumulators went out before the chain had finished)
          const int gout = ((cur.g & 1) * PS_PROD + wave) * PS_GSTRIDE + l15 * 16 + 4 * quad;
#pragma unroll
          for (int k = 0; k < PS_GW; ++k) *reinterpret_cast<f32x4 *>(gram_f + gout + k * 256) = gacc[k];
        }
      }
      PS_TICK(7);
      // ---- the LAST producer to get here completes |q|^2 of the step's new rows (every producer's part is in LDS by then)
      // and publishes the round; the others go straight on to the next step
      if (ps_arrive_ticket(sync_a + C_NORM, ln) == 4u * k + 3u) {
        if (ln < 16 * cur.ntiles) {
          const int rrow = ((ln >> 4) ? cur.slot1 : cur.slot0) * 16 + l15;
          f32x4 v, v2;
          lds_read2(qnp_a + (unsigned)rrow * 16u, qnp_a + (unsigned)rrow * 16u, v, v2);
          ring[rrow * LDQ + 2 * FP] = (v[0] + v[1]) + (v[2] + v[3]);
        }
        if (cur.has_dest) ps_publish(sync_a + C_READY, (unsigned)cur.g + 1u, ln);
      }
      PS_TICK(8);
      cur = nxt;
      have = have_next;
    }
    dma_wait();  // (the last step's re-fetch)
    PS_TFLUSH(0, 9, 0);
    return;
  }

  // ============================================================================================= CONSUMER
  const int cw = wave - PS_PROD;            // 0..7
  if constexpr (LG8) {
    // ---- eight lanes per destination: lane q of a group owns columns 4 q .. 4 q + 3 and 32 + 4 q .. 32 + 4 q + 3 ("chunks" 0, 1)
    // of the filter and of the core half, a wave owns EIGHT destinations -- half a round -- and takes the next half round off
    // the ticket counter.  Same arithmetic per column as the sixteen-lane form below; what changes is that a triplet's fixed
    // cost (row address, cross-term reduction -- three DPP steps now --, variance, rsq) is spread over sixteen gates per lane.
    const int l7 = lane & 7, grp = lane >> 3;
    const float inv2n = 1.0f / (float)(2 * a.d.Fe), invn = 1.0f / (float)a.d.Fe;
    const float spscale = wsc(1) * wsc(1) * inv2n;  // P' rows arrive prescaled (see s_g3q)
    const float eps_c2 = 1e-5f * wsc(4) * wsc(4);   // so do the c2 rows
    int nval[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) nval[j] = min(max(a.d.Fe - (4 * l7 + 32 * j), 0), 4);
    f32x2 bf2[2][2], bc2[2][2];  // c3_norm_1's shift with the exp2 scale of the gate folded in
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const Vec4<float> bf = load4<float>(a.w.c3_norm_1s.b + 4 * l7 + 32 * j), bc = load4<float>(a.w.c3_norm_1s.b + FP + 4 * l7 + 32 * j);
      bf2[j][0] = f32x2{bf.v[0], bf.v[1]};
      bf2[j][1] = f32x2{bf.v[2], bf.v[3]};
      bc2[j][0] = f32x2{bc.v[0], bc.v[1]};
      bc2[j][1] = f32x2{bc.v[2], bc.v[3]};
      // (consumed here: left pending, these loads get their vmcnt waits INSIDE the triplet loop -- their first use)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(bf2[j][0]), "+v"(bf2[j][1]), "+v"(bc2[j][0]), "+v"(bc2[j][1]));
    }
    const float *ringc = ring + 4 * l7;
    const int sdelta = 2 * FP - 4 * l7;  // from this lane's first filter columns of a row to the row's |q|^2
    const unsigned total_rounds = (unsigned)nunits * (unsigned)nrounds;
    // LayerNorm of a 64-column row spread over the group's eight lanes, two chunks of four columns per lane
    auto ln64 = [&](const Vec4<float> (&x)[2], const float *gt, const float *bt, Vec4<float> (&y)[2]) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) sum += (x[j].v[0] + x[j].v[1]) + (x[j].v[2] + x[j].v[3]);
      const float mean = lg_sum<8>(sum) * invn;
      float dv[2][4], qq = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dv[j][k] = x[j].v[k] - mean;
          if (PAD && k >= nval[j]) dv[j][k] = 0.f;
          qq = fmaf(dv[j][k], dv[j][k], qq);
        }
      const float rstd = fast_rsq(lg_sum<8>(qq) * invn + 1e-5f);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const Vec4<float> gv = load4<float>(gt + 4 * l7 + 32 * j), bv = load4<float>(bt + 4 * l7 + 32 * j);
#pragma unroll
        for (int k = 0; k < 4; ++k) y[j].v[k] = dv[j][k] * rstd * gv.v[k] + bv.v[k];
      }
    };
    bool poisoned = false;  // a bounded wait ran out in this workgroup: store NaN from here on
#if RN_PS_TIMING
    long long *tacc = reinterpret_cast<long long *>(a.fail + 16);
    const bool timed = blockIdx.x == 0 && (cw & 3) == 0 && lane == 0;  // waves 4 and 8
    unsigned tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tn = 0;
#endif
    for (;;) {
      PS_T0();
      const unsigned tk = ps_arrive_ticket(sync_a + C_TICKET, lane);
      const unsigned gr = tk >> 1;
      if (gr >= total_rounds) break;
      const int slot = 8 * (int)(tk & 1u) + grp;  // the destination of the round this 8-lane group owns
      const int u = (int)(gr / (unsigned)nrounds), r = (int)(gr - (unsigned)u * (unsigned)nrounds);
      const int ub = (int)(((unsigned)u * (unsigned)(nrt % NRT)) % (unsigned)NRT) * 16;  // ring row of the unit's first source row
      poisoned |= (ps_wait_ge(sync_a + C_READY, gr + 1u, sync_a + C_READY, a.fail, 4) & PS_FAILBIT) != 0;
      PS_TICK(10);
      const int64_t erow0 = (int64_t)(sg + u * nsg) * g.E;
      const bool active = r * PS_ND + slot < D;
      const int i = min(r * PS_ND + slot, D - 1);  // (a lane group beyond the tile's last destination: zero triplets of a valid one)
      const int64_t drow = erow0 + d_edge[i];
      // ---- the P' row -> registers, folded for the loop: pd = p / gamma * (2 / 2Fe), pg = p * gamma;
      //      var + eps = pd.qg + (|p|^2 / 2Fe + eps) + |q|^2 / 2Fe
      f32x2 pf2[2][2], pc2[2][2], pdf2[2][2], pdc2[2][2];
      float spe;
      {
        const float *prow = bufP + ((int)(gr & 1u) * PS_ND + slot) * LDQ;
        float sp = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int cj = 4 * l7 + 32 * j;
          const Vec4<float> xf = load4<float>(prow + cj), xc = load4<float>(prow + FP + cj);
          const Vec4<float> gf = load4<float>(s_g3p + cj), gc = load4<float>(s_g3p + FP + cj);
          const Vec4<float> jf = load4<float>(s_igp + cj), jc = load4<float>(s_igp + FP + cj);
#pragma unroll
          for (int k = 0; k < 4; ++k) sp += xf.v[k] * xf.v[k] + xc.v[k] * xc.v[k];  // zero-mean by construction
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            pdf2[j][hh] = f32x2{xf.v[2 * hh] * jf.v[2 * hh], xf.v[2 * hh + 1] * jf.v[2 * hh + 1]};
            pdc2[j][hh] = f32x2{xc.v[2 * hh] * jc.v[2 * hh], xc.v[2 * hh + 1] * jc.v[2 * hh + 1]};
            pf2[j][hh] = f32x2{xf.v[2 * hh] * gf.v[2 * hh], xf.v[2 * hh + 1] * gf.v[2 * hh + 1]};
            pc2[j][hh] = f32x2{xc.v[2 * hh] * gc.v[2 * hh], xc.v[2 * hh + 1] * gc.v[2 * hh + 1]};
          }
        }
        spe = lg_sum<8>(sp) * spscale + 1e-5f;
      }
      // ---- c2: gate(LayerNorm(c2_linear(node[b]*node[a]))) -> LayerNorm   (_gnn.py:223-228), before the loop: the round's
      // P' / c2 buffers then go back to the producers.  Zero-mean pre-activation row (centred weights), exact zeros in its
      // padded columns: its variance is the plain sum of squares (in the weights' prescale: eps scaled).
      Vec4<float> c2v[2];
      {
        const float *crow = bufC + ((int)(gr & 1u) * PS_ND + slot) * LDQ;
        Vec4<float> xf[2], xc[2];
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          xf[j] = load4<float>(crow + 4 * l7 + 32 * j);
          xc[j] = load4<float>(crow + FP + 4 * l7 + 32 * j);
#pragma unroll
          for (int k = 0; k < 4; ++k) q += xf[j].v[k] * xf[j].v[k] + xc[j].v[k] * xc[j].v[k];
        }
        const float rstd2 = fast_rsq(lg_sum<8>(q) * inv2n + eps_c2);
        Vec4<float> g2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int cj = 4 * l7 + 32 * j;
          const Vec4<float> gf = load4<float>(s_c21g + cj), bf = load4<float>(s_c21b + cj);
          const Vec4<float> gc = load4<float>(s_c21g + FP + cj), bc = load4<float>(s_c21b + FP + cj);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            g2[j].v[k] = (RN_PS_PROBE & 4) ? xf[j].v[k] + xc[j].v[k]
                                           : gate(xf[j].v[k] * rstd2 * gf.v[k] + bf.v[k], xc[j].v[k] * rstd2 * gc.v[k] + bc.v[k]);
        }
        if (RN_PS_PROBE & 4) {
          c2v[0] = g2[0];
          c2v[1] = g2[1];
        } else {
          ln64(g2, s_c22g, s_c22b, c2v);
        }
      }
      const int rb = d_rb[i], cnt = active ? d_cnt[i] : 0, rskip = d_skip[i];
      const int rbase = (int)wrap_row(wrap_row((unsigned)(ub + rb)));  // ring row of the destination's first source row
      ps_arrive(sync_a + ((gr & 1u) ? C_FREE1 : C_FREE0), lane);  // this wave holds what it needs of round gr's P' / c2 buffers
      PS_TICK(11);
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      {
        auto triplet = [&](const float *qr, float (&sumk)[8]) {
          f32x2 qf2[2][2], qc2[2][2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float4 qfv = *reinterpret_cast<const float4 *>(qr + 32 * j), qcv = *reinterpret_cast<const float4 *>(qr + FP + 32 * j);
            qf2[j][0] = f32x2{qfv.x, qfv.y};
            qf2[j][1] = f32x2{qfv.z, qfv.w};
            qc2[j][0] = f32x2{qcv.x, qcv.y};
            qc2[j][1] = f32x2{qcv.z, qcv.w};
          }
          const float qs = qr[sdelta];
          f32x2 d2 = pdf2[0][0] * qf2[0][0], d3 = pdc2[0][0] * qc2[0][0];
          d2 = __builtin_elementwise_fma(pdf2[0][1], qf2[0][1], d2);
          d3 = __builtin_elementwise_fma(pdc2[0][1], qc2[0][1], d3);
          d2 = __builtin_elementwise_fma(pdf2[1][0], qf2[1][0], d2);
          d3 = __builtin_elementwise_fma(pdc2[1][0], qc2[1][0], d3);
          d2 = __builtin_elementwise_fma(pdf2[1][1], qf2[1][1], d2);
          d3 = __builtin_elementwise_fma(pdc2[1][1], qc2[1][1], d3);
          d2 += d3;
          const float dot = lg_sum<8>(d2.x + d2.y);
          float ve = dot + (spe + qs);
          ve = ve > 1e-5f ? ve : 1e-5f;
          const float rstd = fast_rsq(ve);
          const f32x2 rstd2 = {rstd, rstd}, one2 = {1.0f, 1.0f};
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const f32x2 xf2 = __builtin_elementwise_fma(pf2[j][hh] + qf2[j][hh], rstd2, bf2[j][hh]);
              const f32x2 xc2 = __builtin_elementwise_fma(pc2[j][hh] + qc2[j][hh], rstd2, bc2[j][hh]);
              const f32x2 e1 = {fast_exp2(xf2.x), fast_exp2(xf2.y)}, e2 = {fast_exp2(xc2.x), fast_exp2(xc2.y)};
              const f32x2 t2 = e2 + one2;  // (1 + e1)(1 + e2) = t2 + e1 t2: one fma
              const f32x2 den = __builtin_elementwise_fma(e1, t2, t2);
              const f32x2 rd = {fast_rcp(den.x), fast_rcp(den.y)};
              f32x2 sk = {sumk[4 * j + 2 * hh], sumk[4 * j + 2 * hh + 1]};
              sk = __builtin_elementwise_fma(e2 - one2, rd, sk);
              sumk[4 * j + 2 * hh] = sk.x;
              sumk[4 * j + 2 * hh + 1] = sk.y;
            }
        };
#if RN_PS_CPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        float acc2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int tskip = rskip - rb;
        auto row_of = [&](int t) { return ringc + __umul24(wrap_row((unsigned)(rbase + t + (t >= tskip ? 1 : 0))), (unsigned)LDQ); };
        int t = (RN_PS_PROBE & 1) ? cnt : 0;
        const int tmid = RN_PS_LPRIO ? ((cnt >> 1) & ~1) : 0;
        for (; t + 1 < tmid; t += 2) {
          triplet(row_of(t), acc);
          triplet(row_of(t + 1), acc2);
        }
#if RN_PS_LPRIO
        __builtin_amdgcn_s_setprio(RN_PS_LPRIO);
#endif
        for (; t + 1 < cnt; t += 2) {
          triplet(row_of(t), acc);
          triplet(row_of(t + 1), acc2);
        }
        if (t < cnt) triplet(row_of(t), acc);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += acc2[k];
      }
      PS_TICK(12);
#if RN_PS_CPRIO
      __builtin_amdgcn_s_setprio(RN_PS_CPRIO);
#endif
      // the destination's own row (the residual): requested behind the loop (eight registers less across it), consumed after c3's LayerNorm
      f32x4 old4[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cj = 4 * l7 + 32 * j;
        if constexpr (PRE) {  // columns cj .. cj + 3 of group m = cj / 8: four f16 of the hi slot, four of the lo slot
          const char *pp = reinterpret_cast<const char *>(a.edge_in + drow * FP + (cj >> 3) * 8) + (cj & 7) * 2;
          const f16x4 hh = *reinterpret_cast<const f16x4 *>(pp), ll = *reinterpret_cast<const f16x4 *>(pp + 16);
#pragma unroll
          for (int k = 0; k < 4; ++k) old4[j][k] = (float)hh[k] + (float)ll[k];
        } else {
          old4[j] = *reinterpret_cast<const f32x4 *>(a.edge_in + drow * FP + cj);
        }
      }
      // this wave no longer reads the ring rows of round gr; arrivals ordered by round (see the sixteen-lane form)
      poisoned |= (ps_wait_ge(sync_a + ((gr & 1u) ? C_RD1 : C_RD0), CPR * (gr >> 1), sync_a + C_READY, a.fail, 6) & PS_FAILBIT) != 0;
      ps_arrive(sync_a + ((gr & 1u) ? C_RD1 : C_RD0), lane);
      if (active) {
        Vec4<float> a4[2], c3[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          a4[j] = Vec4<float>{{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]}};
          if (a.agg_out) store4(a.agg_out + drow * FP + 4 * l7 + 32 * j, a4[j]);
        }
        if (RN_PS_PROBE & 4) {
          c3[0] = a4[0];
          c3[1] = a4[1];
        } else {
          ln64(a4, s_c3g, s_c3b, c3);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int cj = 4 * l7 + 32 * j;
          f32x4 y;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            y[k] = (RN_PS_PROBE & 4) ? old4[j][k] + c2v[j].v[k] + c3[j].v[k] : fast_tanh(old4[j][k] + c2v[j].v[k] + c3[j].v[k]);
            if (poisoned) y[k] = __int_as_float(0x7fc00000);
          }
          if constexpr (PRE) {
            char *pp = reinterpret_cast<char *>(a.edge_out + drow * FP + (cj >> 3) * 8) + (cj & 7) * 2;
            f16x4 hh, ll;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              hh[k] = (_Float16)y[k];
              ll[k] = (_Float16)(y[k] - (float)hh[k]);
            }
            *reinterpret_cast<f16x4 *>(pp) = hh;
            *reinterpret_cast<f16x4 *>(pp + 16) = ll;
          } else {
            *reinterpret_cast<f32x4 *>(a.edge_out + drow * FP + cj) = y;
          }
        }
      }
      PS_TICK(13);
    }
    PS_TFLUSH(10, 13, (cw >> 2) ? 6 : 0);  // (wave 4: slots 10-13, wave 8: 16-19)
    return;
  }
  const int cset = cw >> 2;                 // 0: the even global rounds, 1: the odd ones
  const int c0 = 4 * l15;                   // lane l15 of a group owns columns c0 .. c0 + 3 of the filter and of the core half
  const int nvalid = min(max(a.d.Fe - c0, 0), 4);
  const float inv2n = 1.0f / (float)(2 * a.d.Fe), invn = 1.0f / (float)a.d.Fe;
  const float spscale = wsc(1) * wsc(1) * inv2n;  // P' rows arrive prescaled (see s_g3q)
  const float eps_c2 = 1e-5f * wsc(4) * wsc(4);   // so do the c2 rows
  f32x2 bf2[2], bc2[2];  // c3_norm_1's shift with the exp2 scale of the gate folded in
  {
    const Vec4<float> bf = load4<float>(a.w.c3_norm_1s.b + c0), bc = load4<float>(a.w.c3_norm_1s.b + FP + c0);
    bf2[0] = f32x2{bf.v[0], bf.v[1]};
    bf2[1] = f32x2{bf.v[2], bf.v[3]};
    bc2[0] = f32x2{bc.v[0], bc.v[1]};
    bc2[1] = f32x2{bc.v[2], bc.v[3]};
    // (consumed here: left pending, these two loads get their vmcnt waits INSIDE the triplet loop -- their first use -- where
    //  every round's own loads and stores are outstanding too)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bf2[0]), "+v"(bf2[1]), "+v"(bc2[0]), "+v"(bc2[1]));
  }
  // LayerNorm parameters and fold tables of this lane's four columns: registers (a consumer has ~40 to spare), straight from
  // the weight blob -- not LDS reads per round, and no LDS copy of the tables either
  LnParams<float> k_c3 = {load4<float>(a.w.c3_norm_2.g + c0), load4<float>(a.w.c3_norm_2.b + c0)};
  LnParams<float> k_c21f = {load4<float>(a.w.c2_norm_1.g + c0), load4<float>(a.w.c2_norm_1.b + c0)};
  LnParams<float> k_c21c = {load4<float>(a.w.c2_norm_1.g + FP + c0), load4<float>(a.w.c2_norm_1.b + FP + c0)};
  LnParams<float> k_c22 = {load4<float>(a.w.c2_norm_2.g + c0), load4<float>(a.w.c2_norm_2.b + c0)};
  Vec4<float> g3f = load4<float>(a.w.c3_norm_1s.g + c0), g3c = load4<float>(a.w.c3_norm_1s.g + FP + c0);
  {  // (consumed here, like bf2 / bc2 above)
    auto pin = [](Vec4<float> &x) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3])); };
    pin(k_c3.g); pin(k_c3.b); pin(k_c21f.g); pin(k_c21f.b); pin(k_c21c.g); pin(k_c21c.b); pin(k_c22.g); pin(k_c22.b);
    pin(g3f); pin(g3c);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // p gamma from a stored P' row (which carries the weights' prescale)
    g3f.v[k] *= wsc(1);
    g3c.v[k] *= wsc(1);
  }
  // (not GRAM) the fold of a P' row for the cross term, 1 / gamma * 2 / 2Fe: the producers' table (written before the barrier)
  Vec4<float> igf = load4<float>(s_igp + c0), igc = load4<float>(s_igp + FP + c0);
  const float *ringc = ring + c0;
  const int sdelta = 2 * FP - c0;  // from this lane's filter columns of a row to the row's |q|^2
  // Which round a wave works on.  RN_PS_TICKET: the next quarter round off the workgroup's counter (quarter q = destinations
  // 4 (q & 3) .. + 3 of global round q >> 2); else the wave's set owns every other round and the wave a fixed quarter of it.
  // Global round gr = (unit u, local round r); ub = ring row of unit u's first source row.
  const unsigned total_rounds = (unsigned)nunits * (unsigned)nrounds;
  auto locate = [&](unsigned gr, int &u, int &r, int &ub) {  // (uniform arithmetic, once per round)
    u = (int)(gr / (unsigned)nrounds);
    r = (int)(gr - (unsigned)u * (unsigned)nrounds);
    ub = (int)(((unsigned)u * (unsigned)(nrt % NRT)) % (unsigned)NRT) * 16;
  };
  int u = 0, r = 0, ub = 0;
  bool poisoned = false;  // a bounded wait ran out in this workgroup: store NaN from here on
#if RN_PS_TIMING
  long long *tacc = reinterpret_cast<long long *>(a.fail + 16);
  const bool timed = blockIdx.x == 0 && (cw & 3) == 0 && lane == 0;  // one wave of either set
  unsigned tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tn = 0;
#endif
  for (unsigned turn = 0;; ++turn) {
    PS_T0();
    unsigned gr;
    int slot;  // the destination of the round this 16-lane group owns
    if (RN_PS_TICKET) {
      const unsigned q = ps_arrive_ticket(sync_a + C_TICKET, lane);
      gr = q >> 2;
      slot = 4 * (int)(q & 3u) + quad;
    } else {
      gr = 2u * turn + (unsigned)cset;
      slot = 4 * (cw & 3) + quad;
    }
    if (gr >= total_rounds) break;
    locate(gr, u, r, ub);
    poisoned |= (ps_wait_ge(sync_a + C_READY, gr + 1u, sync_a + C_READY, a.fail, 4) & PS_FAILBIT) != 0;
    PS_TICK(10);
    const int64_t erow0 = (int64_t)(sg + u * nsg) * g.E;
    const bool active = r * PS_ND + slot < D;
    const int i = min(r * PS_ND + slot, D - 1);  // (a lane group beyond the tile's last destination: zero triplets of a valid one)
    const int64_t drow = erow0 + d_edge[i];
    // the destination's own row (the residual): requested now, consumed in the epilogue
    f32x4 old4;
    if constexpr (PRE) {  // columns c0 .. c0 + 3 of group m = c0 / 8: four f16 of the hi slot, four of the lo slot
      const char *pp = reinterpret_cast<const char *>(a.edge_in + drow * FP + (c0 >> 3) * 8) + (c0 & 7) * 2;
      const f16x4 hh = *reinterpret_cast<const f16x4 *>(pp), ll = *reinterpret_cast<const f16x4 *>(pp + 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) old4[k] = (float)hh[k] + (float)ll[k];
    } else {
      old4 = *reinterpret_cast<const f32x4 *>(a.edge_in + drow * FP + c0);
    }
    // ---- the P' row -> registers, folded for the loop
    f32x2 pf2[2], pc2[2], pdf2[2], pdc2[2];
    float spe;
    {
      const float *prow = bufP + ((int)(gr & 1u) * PS_ND + slot) * LDQ;
      const Vec4<float> xf = load4<float>(prow + c0), xc = load4<float>(prow + FP + c0);
      float sp = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sp += xf.v[k] * xf.v[k] + xc.v[k] * xc.v[k];  // zero-mean by construction
      sp = lg_sum<LG>(sp);
      // pd = p / gamma * (2 / 2Fe), pg = p * gamma;  var + eps = pd.qg + (|p|^2 / 2Fe + eps) + |q|^2 / 2Fe
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if constexpr (!GRAM) {
          pdf2[hh] = f32x2{xf.v[2 * hh] * igf.v[2 * hh], xf.v[2 * hh + 1] * igf.v[2 * hh + 1]};
          pdc2[hh] = f32x2{xc.v[2 * hh] * igc.v[2 * hh], xc.v[2 * hh + 1] * igc.v[2 * hh + 1]};
        }
        pf2[hh] = f32x2{xf.v[2 * hh] * g3f.v[2 * hh], xf.v[2 * hh + 1] * g3f.v[2 * hh + 1]};
        pc2[hh] = f32x2{xc.v[2 * hh] * g3c.v[2 * hh], xc.v[2 * hh + 1] * g3c.v[2 * hh + 1]};
      }
      spe = sp * spscale + 1e-5f;
    }
    // ---- c2: gate(LayerNorm(c2_linear(node[b]*node[a]))) -> LayerNorm   (_gnn.py:223-228).  It does not depend on the
    // triplets, so it runs here and the round's P' / c2 buffers go back to the producers before the loop starts.  The
    // pre-activation row has zero mean (centred weights) and exact zeros in its padded columns: its variance is the plain
    // sum of squares (in the weights' prescale: eps scaled).
    Vec4<float> c2v;
    {
      const float *crow = bufC + ((int)(gr & 1u) * PS_ND + slot) * LDQ;
      const Vec4<float> xf = load4<float>(crow + c0), xc = load4<float>(crow + FP + c0);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) q += xf.v[k] * xf.v[k] + xc.v[k] * xc.v[k];
      const float rstd2 = fast_rsq(lg_sum<LG>(q) * inv2n + eps_c2);
      Vec4<float> g2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g2.v[k] = (RN_PS_PROBE & 4) ? xf.v[k] + xc.v[k]
                                    : gate(xf.v[k] * rstd2 * k_c21f.g.v[k] + k_c21f.b.v[k], xc.v[k] * rstd2 * k_c21c.g.v[k] + k_c21c.b.v[k]);
      }
      c2v = (RN_PS_PROBE & 4) ? g2 : ln_row<LG, PAD, float>(g2, k_c22, invn, nvalid);
    }
    const int rb = d_rb[i], cnt = active ? d_cnt[i] : 0, rskip = d_skip[i];
    const int rbase = (int)wrap_row(wrap_row((unsigned)(ub + rb)));  // ring row of the destination's first source row
    // ---- GRAM: 1 / sqrt(var + eps) of every (this destination, source row) pair, sixteen source rows per instruction: lane j
    // of the group takes source rows j and j + 16 of the destination's list (the reverse edge's row among them: computed,
    // never used), adds the four producers' partial cross terms, |p|^2 and |q|^2, and keeps the two results; the loop
    // fetches the one it needs from the lane that has it (ds_bpermute: the LDS crossbar, no LDS memory).
    float rs0 = 0.f, rs1 = 0.f;
    if constexpr (GRAM) {
      const int nrow = d_re[i] - rb;  // (<= 32: edge_ps_gram_ok)
      const int lo_r = __builtin_amdgcn_readfirstlane(lo_t[r]);
      const float *gtab = reinterpret_cast<const float *>(smem_raw + L.gram) + (int)(gr & 1u) * (PS_PROD * PS_GSTRIDE) + slot;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int er = min(l15 + 16 * pass, max(nrow - 1, 0));
        const int urow = rb + er;                                     // row of the unit
        const int tl = min(max((urow >> 4) - lo_r, 0), PS_GW - 1);    // tile of the round's window
        const float *gp = gtab + (tl * 16 + (urow & 15)) * 16;
        const float qs = ring[wrap_row((unsigned)(rbase + er)) * LDQ + 2 * FP];
#ifdef RN_PS_GRAM_DEBUG  // (debug builds: the cross term by a serial dot product instead of the tables)
        float gsum = 0.f;
        {
          const float *prow = bufP + ((int)(gr & 1u) * PS_ND + slot) * LDQ, *qrow = ring + wrap_row((unsigned)(rbase + er)) * LDQ;
          for (int c = 0; c < 2 * FP; ++c) gsum = fmaf(prow[c] * s_igp[c], qrow[c], gsum);
        }
        {  // (debug: the first few mismatches between the tables and the serial dot product go to fail[128 ..])
          const float gt = (gp[0] + gp[PS_GSTRIDE]) + (gp[2 * PS_GSTRIDE] + gp[3 * PS_GSTRIDE]);
          if (active && l15 + 16 * pass < nrow && fabsf(gt - gsum) > 1e-4f * (fabsf(gsum) + 1e-3f)) {
            const int k = atomicAdd(a.fail + 127, 1);
            if (k < 12) {
              float *dbg = reinterpret_cast<float *>(a.fail + 128 + 12 * k);
              dbg[0] = gsum; dbg[1] = gt; dbg[2] = gp[0]; dbg[3] = gp[PS_GSTRIDE]; dbg[4] = gp[2 * PS_GSTRIDE]; dbg[5] = gp[3 * PS_GSTRIDE];
              dbg[6] = (float)slot; dbg[7] = (float)er; dbg[8] = (float)tl; dbg[9] = (float)urow; dbg[10] = (float)r; dbg[11] = (float)gr;
            }
          }
        }
        float ve = gsum + (spe + qs);
#else
        float ve = (gp[0] + gp[PS_GSTRIDE]) + (gp[2 * PS_GSTRIDE] + gp[3 * PS_GSTRIDE]) + (spe + qs);
#endif
        ve = ve > 1e-5f ? ve : 1e-5f;
        (pass ? rs1 : rs0) = fast_rsq(ve);
      }
    }
    ps_arrive(sync_a + ((gr & 1u) ? C_FREE1 : C_FREE0), lane);  // this wave holds what it needs of round gr's P' / c2 buffers
    PS_TICK(11);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    {
      // `er` (GRAM): index of the source row in the destination's list = which lane of the group holds its rstd, and in which register
      auto triplet = [&](const float *qr, int er, float (&sumk)[4]) {
        const float4 qfv = *reinterpret_cast<const float4 *>(qr), qcv = *reinterpret_cast<const float4 *>(qr + FP);
        const f32x2 qf2[2] = {{qfv.x, qfv.y}, {qfv.z, qfv.w}}, qc2[2] = {{qcv.x, qcv.y}, {qcv.z, qcv.w}};
        float rstd;
        if constexpr (GRAM) {
          const int src = ((lane & 48) | (er & 15)) << 2;
          rstd = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(er < 16 ? rs0 : rs1)));
        } else {
          const float qs = qr[sdelta];
          f32x2 d2 = pdf2[0] * qf2[0];
          f32x2 d3 = pdc2[0] * qc2[0];
          d2 = __builtin_elementwise_fma(pdf2[1], qf2[1], d2);
          d3 = __builtin_elementwise_fma(pdc2[1], qc2[1], d3);
          d2 += d3;
          const float dot = lg_sum<LG>(d2.x + d2.y);
          float ve = dot + (spe + qs);
          ve = ve > 1e-5f ? ve : 1e-5f;
          rstd = fast_rsq(ve);
        }
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
#if RN_PS_CPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      // two independent triplets per iteration.  Source row of triplet t: rb + t, plus one from the reverse edge on
      // (the numbering jumps over it); its ring row is (that + the unit's ring base) wrapped -- a compare, an add-with-carry,
      // the wrap and one 24-bit multiply per triplet, no branches and no running pointer to wrap
      float acc2[4] = {0.f, 0.f, 0.f, 0.f};
      const int tskip = rskip - rb;
      auto er_of = [&](int t) { return t + (t >= tskip ? 1 : 0); };
      auto row_of = [&](int er) { return ringc + __umul24(wrap_row((unsigned)(rbase + er)), (unsigned)LDQ); };
      int t = (RN_PS_PROBE & 1) ? cnt : 0;
      const int tmid = RN_PS_LPRIO ? ((cnt >> 1) & ~1) : 0;
      for (; t + 1 < tmid; t += 2) {
        const int e0 = er_of(t), e1 = er_of(t + 1);
        triplet(row_of(e0), e0, acc);
        triplet(row_of(e1), e1, acc2);
      }
#if RN_PS_LPRIO
      __builtin_amdgcn_s_setprio(RN_PS_LPRIO);
#endif
      for (; t + 1 < cnt; t += 2) {
        const int e0 = er_of(t), e1 = er_of(t + 1);
        triplet(row_of(e0), e0, acc);
        triplet(row_of(e1), e1, acc2);
      }
      if (t < cnt) {
        const int e0 = er_of(t);
        triplet(row_of(e0), e0, acc);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += acc2[k];
    }
    PS_TICK(12);
#if RN_PS_CPRIO
    __builtin_amdgcn_s_setprio(RN_PS_CPRIO);
#endif
    // this wave no longer reads the ring rows of round gr (all its LDS reads are complete: ps_arrive waits for them).  The
    // counter is cumulative over the rounds of gr's parity, and a producer reads "count >= 4 n" as "the first n of them are
    // finished": so a wave adds its arrival for round gr only when every arrival for the earlier ones is in (it almost always is).
    poisoned |= (ps_wait_ge(sync_a + ((gr & 1u) ? C_RD1 : C_RD0), CPR * (gr >> 1), sync_a + C_READY, a.fail, 6) & PS_FAILBIT) != 0;
    ps_arrive(sync_a + ((gr & 1u) ? C_RD1 : C_RD0), lane);
    if (active) {
      const Vec4<float> a4 = {{acc[0], acc[1], acc[2], acc[3]}};
      if (a.agg_out) store4(a.agg_out + drow * FP + c0, a4);
      const Vec4<float> c3 = (RN_PS_PROBE & 4) ? a4 : ln_row<LG, PAD, float>(a4, k_c3, invn, nvalid);
      f32x4 y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        y[k] = (RN_PS_PROBE & 4) ? old4[k] + c2v.v[k] + c3.v[k] : fast_tanh(old4[k] + c2v.v[k] + c3.v[k]);
        if (poisoned) y[k] = __int_as_float(0x7fc00000);
      }
      if constexpr (PRE) {
        char *pp = reinterpret_cast<char *>(a.edge_out + drow * FP + (c0 >> 3) * 8) + (c0 & 7) * 2;
        f16x4 hh, ll;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hh[k] = (_Float16)y[k];
          ll[k] = (_Float16)(y[k] - (float)hh[k]);
        }
        *reinterpret_cast<f16x4 *>(pp) = hh;
        *reinterpret_cast<f16x4 *>(pp + 16) = ll;
      } else {
        *reinterpret_cast<f32x4 *>(a.edge_out + drow * FP + c0) = y;
      }
    }
    PS_TICK(13);
  }
  PS_TFLUSH(10, 13, cset ? 6 : 0);  // (set A: slots 10-13, set B: 16-19)
}

// ---- host side ---------------------------------------------------------------------------------------------
size_t edge_ps_lds_bytes(int rows, int in_rows, bool gram) { return ps_lds(rows, in_rows, gram).total; }
int edge_ps_ring_tiles(bool gram) { return gram ? PS_NRT_GRAM : PS_NRT; }
int edge_ps_gram_window() { return PS_GW; }

// Runs the producers' schedule for one tile (first destination index and end row per destination, sorted by atom)
// and checks what the kernel takes for granted: never more than PS_MAXNEW source tiles in one step, and no ring slot is
// rewritten while a round that may still be in flight (g - back .. g - 1 at the step of round g) reads the tile it holds.
// back = 2 is the least the two consumer sets need; back = 3 (Graph::pt_back, when the ring has the room) lets the producers
// run one more round ahead of the slower set: 7.44 -> 6.91 ms per launch on the benchmark cell.
bool edge_ps_tile_ok(const int *rb, const int *re, int D, int back, int ring, int *window) {
  if (D <= 0) return true;
  const int nrounds = (D + PS_ND - 1) / PS_ND;
  std::vector<int> hi(nrounds), lo(nrounds);
  int top = 0;
  for (int r = 0; r < nrounds; ++r) {
    const int last = std::min((r + 1) * PS_ND, D);
    int first_row = 1 << 30;
    for (int i = r * PS_ND; i < last; ++i) {
      top = std::max(top, re[i]);
      if (re[i] > rb[i]) first_row = std::min(first_row, rb[i]);
    }
    hi[r] = (top + 15) / 16 - 1;
    lo[r] = first_row == (1 << 30) ? (hi[r] + 1) : first_row / 16;
  }
  const int nrt = std::max(hi[nrounds - 1] + 1, 1);
  if (nrt > 2 * ring) return false;  // (the kernel wraps ring positions by at most two subtractions)
  const int units = 3 + back;
  PsSched s;
  if (window) {  // most source tiles a round's window spans (what the Gram tables must hold)
    int w = 0;
    for (int r = 0; r < nrounds; ++r) w = std::max(w, hi[r] - std::min(lo[r], hi[r] + 1) + 1);
    *window = w;
  }
  ps_sched_init(s, nrounds, nrt, units, hi[0], ring);
  PsStep st;
  auto hif = [&](int r) { return hi[r]; };
  while (ps_sched_next(s, hif, st)) {
    if (st.ntiles > PS_MAXNEW) return false;
    if (!st.has_dest) {
      if (st.tile0 + st.ntiles > ring) return false;
      continue;
    }
    // the consumer sets may still be reading rounds g - back .. g - 1 (the step waits for round g - back - 1 only): the first
    // tile of round g - back and everything after it must survive this step
    const int g1 = std::max(st.g - back, 0), u1 = g1 / nrounds, r1 = g1 % nrounds;
    const int oldest = u1 * nrt + std::min(lo[r1], hi[r1] + 1);
    if (st.tile0 + st.ntiles - oldest > ring) return false;
    if (st.u * nrt + hi[st.r] + 1 > st.tile0 + st.ntiles) return false;  // the round's own rows exist
  }
  return true;
}

void launch_edge_ps(const float *edge_in, float *edge_out, const float *node, const float *np3, float *agg_out, int S,
                    const Graph &g, Dims d, const PassW<float> &w, int *fail, hipStream_t st, bool pair_rows, bool f16) {
  if (S == 0 || g.E == 0) return;
  EdgePsArgs a{edge_in, edge_out, node, np3, agg_out, fail, S, g, d, w};
  const bool gram = g.pt_gram != 0 && f16;
  if (!f16) pair_rows = false;
  const size_t lds = ps_lds(g.pt_max_out_rows, g.pt_max_in_rows, gram).total;
  const bool pad = d.Fe != d.FeP;
  auto kern = !f16 ? (pad ? &edge_block_ps_kernel<true, false, false, false> : &edge_block_ps_kernel<false, false, false, false>)
              : gram ? (pair_rows ? (pad ? &edge_block_ps_kernel<true, true, true> : &edge_block_ps_kernel<false, true, true>)
                                  : (pad ? &edge_block_ps_kernel<true, false, true> : &edge_block_ps_kernel<false, false, true>))
                     : (pair_rows ? (pad ? &edge_block_ps_kernel<true, true, false> : &edge_block_ps_kernel<false, true, false>)
                                  : (pad ? &edge_block_ps_kernel<true, false, false> : &edge_block_ps_kernel<false, false, false>));
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int nsg = cus / g.pt_num;  // one workgroup per CU
  nsg = nsg < 1 ? 1 : (nsg > S ? S : nsg);
  kern<<<(unsigned)nsg * (unsigned)g.pt_num, PS_THREADS, lds, st>>>(a);
}

}  // namespace rn
