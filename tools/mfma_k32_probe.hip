// Probe (gfx950): is a chain of v_mfma_f32_16x16x32_f16 (new on gfx950) reproducible when its operands
// come from LDS and other waves of the SIMD run transcendental-heavy VALU code -- the situation inside
// the fused EdgeBlock kernel (profiles/r02/determinism.txt)?  The kernel is launched several times on the
// same input; any lane whose checksum differs between launches is counted.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_k32_probe.hip -o tools/mfma_k32_probe && tools/mfma_k32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K32>
__device__ __forceinline__ f32x4 split3(const f16x8 &ah, const f16x8 &al, const f16x8 &bh, const f16x8 &bl, f32x4 acc) {
  if (K32) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al.lo, bh.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al.hi, bh.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah.lo, bl.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah.hi, bl.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah.lo, bh.lo, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(ah.hi, bh.hi, acc, 0, 0, 0);
}

// waves 0..3: MFMA role (operands from LDS, results to LDS and into a running checksum);
// waves 4..7 (CORUN): the VALU mix of the triplet loop, one such wave on every SIMD beside an MFMA wave
template <int K32, int CORUN>
__global__ __launch_bounds__(512) void probe(const _Float16 *frag, float *out, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 tile[4][16 * 64 * 2];  // per wave: [hi|lo] fragments, 4 KB
  __shared__ __attribute__((aligned(16))) float res[4][16 * 132];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= 4) {
    if (!CORUN) return;
    float a = 0.001f * lane + 0.5f, b = 0.3f, s = 0.f;
    for (int it = 0; it < iters * 24; ++it) {
      const float e1 = __builtin_amdgcn_exp2f(a * 0.25f - 1.0f), e2 = __builtin_amdgcn_exp2f(b * 0.5f - 0.5f);
      const float t2 = 1.0f + e2;
      s = fmaf(e2 - 1.0f, __builtin_amdgcn_rcpf(fmaf(e1, t2, t2)), s);
      a = fmaf(a, 0.999f, 0.001f);
      b = fmaf(b, 1.001f, -0.0007f);
    }
    out[(size_t)blockIdx.x * 512 + tid] = s;
    return;
  }
  const int l15 = lane & 15, quad = lane >> 4;
  // B fragments (weights), resident: 2 tiles x 2 K-slices x (hi, lo)
  f16x8 bh[2][2], bl[2][2];
  const f16x8 *src = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
  for (int t = 0; t < 2; ++t)
    for (int s = 0; s < 2; ++s) {
      bh[t][s] = src[((wave * 4 + t * 2 + s) * 2 + 0) * 64 + lane];
      bl[t][s] = src[((wave * 4 + t * 2 + s) * 2 + 1) * 64 + lane];
    }
  // this wave's operand tile in LDS
  for (int i = lane; i < 16 * 64 * 2 / 8; i += 64)
    reinterpret_cast<f16x8 *>(tile[wave])[i] = src[2048 + wave * 256 + i];
  f32x4 sum[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it) {
    f16x8 ah[2], al[2];
    const f16x8 *tp = reinterpret_cast<const f16x8 *>(tile[wave]);
    const int rot = it & 3;
    for (int s = 0; s < 2; ++s) {
      ah[s] = tp[(l15 * 8 + ((quad * 2 + s + rot) & 7))];
      al[s] = tp[128 + (l15 * 8 + ((quad * 2 + s + rot) & 7))];
    }
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int s = 0; s < 2; ++s)
      for (int t = 0; t < 2; ++t) acc[t] = split3<K32>(bh[t][s], bl[t][s], ah[s], al[s], acc[t]);
    for (int t = 0; t < 2; ++t) {
      *reinterpret_cast<f32x4 *>(res[wave] + l15 * 132 + 16 * t + 4 * quad) = acc[t];
      sum[t] += acc[t];
    }
    // read back what another lane wrote (as the VALU phase of the real kernel does)
    const f32x4 other = *reinterpret_cast<const f32x4 *>(res[wave] + ((l15 + 1) & 15) * 132 + 4 * quad);
    sum[0] += other * 0.125f;
  }
  float total = 0.f;
  for (int t = 0; t < 2; ++t)
    for (int j = 0; j < 4; ++j) total += sum[t][j];
  out[(size_t)blockIdx.x * 512 + tid] = total;
}

// A dependent chain whose links live in DIFFERENT registers (every intermediate is used again later, so
// the accumulator cannot be updated in place): x = A1 B1, y = A2 B2 + x, z = A3 B3 + y.  PAD puts 64 idle
// cycles after every MFMA; the padded and the plain build must agree bit for bit.
template <int K32, int PAD>
__global__ __launch_bounds__(256) void chain(const _Float16 *frag, float *out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f16x8 *src = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
  f16x8 a[3], b[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = src[(wave * 8 + i) * 64 + lane];
    b[i] = src[2048 + (wave * 8 + i) * 64 + lane];
  }
  f32x4 sum = {0, 0, 0, 0};
  const f32x4 zero = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 x, y, z;
    if (K32) {
      x = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], zero, 0, 0, 0);
      if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(x));
      y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[1], x, 0, 0, 0);
      if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(y));
      z = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], b[2], y, 0, 0, 0);
      if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(z));
    } else {
      x = __builtin_amdgcn_mfma_f32_16x16x16f16(a[0].lo, b[0].lo, zero, 0, 0, 0);
      if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(x));
      y = __builtin_amdgcn_mfma_f32_16x16x16f16(a[1].lo, b[1].lo, x, 0, 0, 0);
      if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(y));
      z = __builtin_amdgcn_mfma_f32_16x16x16f16(a[2].lo, b[2].lo, y, 0, 0, 0);
      if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(z));
    }
    sum += x * 0.5f + y * 0.25f + z;  // x and y stay live next to z
    a[0][it & 7] += (_Float16)0.001f;  // the operands change, nothing can be hoisted
  }
  out[(size_t)blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
}

// The exact instruction pattern the compiler emits inside the fused kernel (kernels_fused.hip, K = 32 build):
//   T  = A1 B1            D  = A2 B2 + T     (SrcC = the previous result, DIFFERENT destination)
//   T  = A3 B1  (again)   D  = A2 B1 + D     E = A4 B2 + T
// back to back in inline assembly (PAD = 0) or with 64 idle cycles after every instruction (PAD = 1).
template <int K32, int PAD>
__global__ __launch_bounds__(256) void pattern(const _Float16 *frag, float *out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f16x8 *src = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
  f16x8 a1 = src[(wave * 8 + 0) * 64 + lane], a2 = src[(wave * 8 + 1) * 64 + lane], a3 = src[(wave * 8 + 2) * 64 + lane],
        a4 = src[(wave * 8 + 3) * 64 + lane], b1 = src[2048 + (wave * 8) * 64 + lane], b2 = src[2048 + (wave * 8 + 1) * 64 + lane];
  f32x4 sum = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 t, d, e;
#define NOPS "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
    if (K32) {
      if (PAD)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %3, %7, 0\n\t" NOPS "v_mfma_f32_16x16x32_f16 %1, %4, %8, %0\n\t" NOPS
                     "v_mfma_f32_16x16x32_f16 %0, %5, %7, 0\n\t" NOPS "v_mfma_f32_16x16x32_f16 %1, %4, %7, %1\n\t" NOPS
                     "v_mfma_f32_16x16x32_f16 %2, %6, %8, %0\n\t" NOPS
                     : "=&v"(t), "=&v"(d), "=&v"(e) : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b1), "v"(b2));
      else
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %3, %7, 0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %8, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %5, %7, 0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %7, %1\n\t"
                     "v_mfma_f32_16x16x32_f16 %2, %6, %8, %0\n\t" NOPS
                     : "=&v"(t), "=&v"(d), "=&v"(e) : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b1), "v"(b2));
    } else {
      const f16x4 a1l = a1.lo, a2l = a2.lo, a3l = a3.lo, a4l = a4.lo, b1l = b1.lo, b2l = b2.lo;
      if (PAD)
        asm volatile("v_mfma_f32_16x16x16_f16 %0, %3, %7, 0\n\t" NOPS "v_mfma_f32_16x16x16_f16 %1, %4, %8, %0\n\t" NOPS
                     "v_mfma_f32_16x16x16_f16 %0, %5, %7, 0\n\t" NOPS "v_mfma_f32_16x16x16_f16 %1, %4, %7, %1\n\t" NOPS
                     "v_mfma_f32_16x16x16_f16 %2, %6, %8, %0\n\t" NOPS
                     : "=&v"(t), "=&v"(d), "=&v"(e) : "v"(a1l), "v"(a2l), "v"(a3l), "v"(a4l), "v"(b1l), "v"(b2l));
      else
        asm volatile("v_mfma_f32_16x16x16_f16 %0, %3, %7, 0\n\tv_mfma_f32_16x16x16_f16 %1, %4, %8, %0\n\t"
                     "v_mfma_f32_16x16x16_f16 %0, %5, %7, 0\n\tv_mfma_f32_16x16x16_f16 %1, %4, %7, %1\n\t"
                     "v_mfma_f32_16x16x16_f16 %2, %6, %8, %0\n\t" NOPS
                     : "=&v"(t), "=&v"(d), "=&v"(e) : "v"(a1l), "v"(a2l), "v"(a3l), "v"(a4l), "v"(b1l), "v"(b2l));
    }
#undef NOPS
    sum += d + e * 0.5f + t * 0.25f;
    a1[it & 7] += (_Float16)0.001f;
  }
  out[(size_t)blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
}

// Operand registers written by VALU moves immediately before the MFMA that reads them (the K = 32 build
// assembles its 4-register operands from halves of two LDS reads this way), back to back vs padded.
template <int PAD, int K32 = 1>
__global__ __launch_bounds__(256) void movfeed(const _Float16 *frag, float *out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f16x8 *src = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 x = *reinterpret_cast<const u32x4 *>(&src[(wave * 8 + 0) * 64 + lane]);
  u32x4 y = *reinterpret_cast<const u32x4 *>(&src[(wave * 8 + 1) * 64 + lane]);
  f16x8 b1 = src[2048 + (wave * 8) * 64 + lane];
  f32x4 sum = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 d, e;
    const unsigned x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
#define NOPS "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define GAP(n) "s_nop " #n "\n\t"
#define FEED(MFMA, G)                                                                                                    \
  asm volatile("v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\tv_mov_b32 v202, %6\n\tv_mov_b32 v203, %7\n\t" G             \
               MFMA " %0, v[200:203], %10, 0\n\t" NOPS                                                                   \
               "v_mov_b32 v200, %4\n\tv_mov_b32 v201, %5\n\tv_mov_b32 v202, %8\n\tv_mov_b32 v203, %9\n\t" G             \
               MFMA " %1, v[200:203], %10, %0\n\t" NOPS                                                                  \
               : "=&v"(d), "=&v"(e) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(b1) \
               : "v200", "v201", "v202", "v203")
#define FEED16(G)                                                                                                        \
  asm volatile("v_mov_b32 v200, %2\n\tv_mov_b32 v201, %3\n\t" G                                                          \
               "v_mfma_f32_16x16x16_f16 %0, v[200:201], %6, 0\n\t" NOPS                                                   \
               "v_mov_b32 v200, %4\n\tv_mov_b32 v201, %5\n\t" G                                                          \
               "v_mfma_f32_16x16x16_f16 %1, v[200:201], %6, %0\n\t" NOPS                                                  \
               : "=&v"(d), "=&v"(e) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(b1.lo) : "v200", "v201")
    if (K32) {
      if (PAD == 0) FEED("v_mfma_f32_16x16x32_f16", "");
      else if (PAD == 1) FEED("v_mfma_f32_16x16x32_f16", GAP(0));
      else if (PAD == 2) FEED("v_mfma_f32_16x16x32_f16", GAP(1));
      else if (PAD == 3) FEED("v_mfma_f32_16x16x32_f16", GAP(2));
      else if (PAD == 4) FEED("v_mfma_f32_16x16x32_f16", GAP(3));
      else if (PAD == 6) FEED("v_mfma_f32_16x16x32_f16", GAP(5));
      else if (PAD == 8) FEED("v_mfma_f32_16x16x32_f16", GAP(7));
      else if (PAD == 12) FEED("v_mfma_f32_16x16x32_f16", GAP(11));
      else if (PAD == 16) FEED("v_mfma_f32_16x16x32_f16", GAP(15));
      else FEED("v_mfma_f32_16x16x32_f16", NOPS);
    } else {
      if (PAD == 0) FEED16("");
      else if (PAD == 1) FEED16(GAP(0));
      else if (PAD == 2) FEED16(GAP(1));
      else if (PAD == 4) FEED16(GAP(3));
      else FEED16(NOPS);
    }
#undef FEED
#undef FEED16
#undef GAP
#undef NOPS
    sum += d * 0.5f + e;
    x[it & 3] += 0x00010001u * (it & 1);  // nudge the halves: the operands change every iteration
  }
  out[(size_t)blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
}

// MFMA result stored to LDS by ds_write_b128 N wait states after the MFMA was issued (the fused kernel
// stores P' / c2 / Q' tiles this way), read back after a full wait.
template <int PAD, int K32, int CONSUMER = 0, int CORUN = 0>
__global__ __launch_bounds__(512) void storeafter(const _Float16 *frag, float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[256 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= 4) {  // (CORUN launches 512 threads) a wave per SIMD that keeps the matrix pipe busy
    const f16x8 *srcc = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
    f16x8 ca = srcc[((wave - 4) * 8 + 4) * 64 + lane], cb = srcc[2048 + ((wave - 4) * 8 + 4) * 64 + lane];  // (< 4096 per block)
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int it = 0; it < iters * 6; ++it) {
      if (CORUN == 1) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca, cb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca, cb, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca, cb, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca, cb, acc3, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x16f16(ca.lo, cb.lo, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x16f16(ca.lo, cb.lo, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x16f16(ca.lo, cb.lo, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x16f16(ca.lo, cb.lo, acc3, 0, 0, 0);
      }
    }
    const f32x4 t = acc0 + acc1 + acc2 + acc3;
    out[(size_t)gridDim.x * 256 + (size_t)blockIdx.x * 256 + (tid - 256)] = t[0] + t[1] + t[2] + t[3];
    return;
  }
  const f16x8 *src = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
  f16x8 a = src[(wave * 8 + 0) * 64 + lane], b = src[2048 + (wave * 8) * 64 + lane];
  const unsigned addr = (unsigned)(size_t)(buf + tid * 4);  // LDS byte address of this lane's 16 bytes
  f32x4 sum = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 d, back = {0, 0, 0, 0};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk = {0, 0};
    const f32x2 one2 = {1.0f, 2.0f};
#define NOPS "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define GAP(n) "s_nop " #n "\n\t"
#define STORE(MFMA, A, B, G)                                                                                 \
  if (CONSUMER == 0)                                                                                          \
    asm volatile(MFMA " %0, %2, %3, 0\n\t" G "ds_write_b128 %4, %0\n\t" NOPS "s_waitcnt lgkmcnt(0)\n\t"        \
                 "ds_read_b128 %1, %4\n\ts_waitcnt lgkmcnt(0)\n\t" NOPS                                        \
                 : "=&v"(d), "=&v"(back) : "v"(A), "v"(B), "v"(addr) : "memory");                             \
  else if (CONSUMER == 1)                                                                                     \
    asm volatile(MFMA " v[204:207], %1, %2, 0\n\t" G "v_add_f32 %0, v207, %3\n\t" NOPS                          \
                 : "=&v"(back[0]) : "v"(A), "v"(B), "v"(1.0f) : "v204", "v205", "v206", "v207");              \
  else                                                                                                        \
    asm volatile(MFMA " v[204:207], %1, %2, 0\n\t" G "v_pk_add_f32 %0, v[206:207], %3\n\t" NOPS                 \
                 : "=&v"(pk) : "v"(A), "v"(B), "v"(one2) : "v204", "v205", "v206", "v207")
    if (K32) {
      if (PAD == 0) STORE("v_mfma_f32_16x16x32_f16", a, b, "");
      else if (PAD == 2) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(1));
      else if (PAD == 4) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(3));
      else if (PAD == 6) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(5));
      else if (PAD == 7) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(6));
      else if (PAD == 8) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(7));
      else if (PAD == 9) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(8));
      else if (PAD == 10) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(9));
      else if (PAD == 12) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(11));
      else if (PAD == 16) STORE("v_mfma_f32_16x16x32_f16", a, b, GAP(15));
      else STORE("v_mfma_f32_16x16x32_f16", a, b, NOPS);
    } else {
      if (PAD == 0) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, "");
      else if (PAD == 2) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(1));
      else if (PAD == 4) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(3));
      else if (PAD == 6) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(5));
      else if (PAD == 7) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(6));
      else if (PAD == 8) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(7));
      else if (PAD == 9) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(8));
      else if (PAD == 10) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(9));
      else if (PAD == 12) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(11));
      else if (PAD == 16) STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, GAP(15));
      else STORE("v_mfma_f32_16x16x16_f16", a.lo, b.lo, NOPS);
    }
#undef STORE
#undef GAP
#undef NOPS
    sum += back;
    sum[0] += pk[0] + pk[1];
    a[it & 7] += (_Float16)0.001f;
  }
  out[(size_t)blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
}
template <int PAD, int K32, int CONSUMER = 0, int CORUN = 0>
static long store_bad(const _Float16 *frag, float *out, int blocks, int iters, const std::vector<float> &ref) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> cur(n);
  storeafter<PAD, K32, CONSUMER, CORUN><<<blocks, CORUN ? 512 : 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(cur.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  long bad = 0;
  for (size_t i = 0; i < n; ++i) bad += std::memcmp(&ref[i], &cur[i], 4) != 0;
  return bad;
}
template <int K32, int CONSUMER = 0>
static void run_store(const char *name, const _Float16 *frag, float *out, int blocks, int iters) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> ref(n);
  storeafter<64, K32, CONSUMER><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  printf("  %s after N wait states  N=0: %ld  2: %ld  4: %ld  6: %ld  7: %ld  8: %ld  10: %ld  12: %ld  16: %ld\n", name,
         store_bad<0, K32, CONSUMER>(frag, out, blocks, iters, ref), store_bad<2, K32, CONSUMER>(frag, out, blocks, iters, ref),
         store_bad<4, K32, CONSUMER>(frag, out, blocks, iters, ref), store_bad<6, K32, CONSUMER>(frag, out, blocks, iters, ref),
         store_bad<7, K32, CONSUMER>(frag, out, blocks, iters, ref), store_bad<8, K32, CONSUMER>(frag, out, blocks, iters, ref),
         store_bad<10, K32, CONSUMER>(frag, out, blocks, iters, ref), store_bad<12, K32, CONSUMER>(frag, out, blocks, iters, ref),
         store_bad<16, K32, CONSUMER>(frag, out, blocks, iters, ref));
}

// The measured MFMA issued right behind PRE other MFMAs (a backed-up matrix pipe, as in the real kernels'
// product chains), then N wait states, then a VALU read of its result.
template <int PAD, int K32, int PRE>
__global__ __launch_bounds__(256) void queued(const _Float16 *frag, float *out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f16x8 *src = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
  f16x8 a = src[(wave * 8 + 0) * 64 + lane], b = src[2048 + (wave * 8) * 64 + lane];
  float sum = 0.f;
  for (int it = 0; it < iters; ++it) {
    float r;
#define NOPS "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define GAP(n) "s_nop " #n "\n\t"
#define PRE32 "v_mfma_f32_16x16x32_f16 v[208:211], %1, %2, 0\n\tv_mfma_f32_16x16x32_f16 v[212:215], %1, %2, 0\n\t" \
              "v_mfma_f32_16x16x32_f16 v[216:219], %1, %2, 0\n\tv_mfma_f32_16x16x32_f16 v[220:223], %1, %2, 0\n\t"
#define PRE16 "v_mfma_f32_16x16x16_f16 v[208:211], %1, %2, 0\n\tv_mfma_f32_16x16x16_f16 v[212:215], %1, %2, 0\n\t" \
              "v_mfma_f32_16x16x16_f16 v[216:219], %1, %2, 0\n\tv_mfma_f32_16x16x16_f16 v[220:223], %1, %2, 0\n\t"
#define Q(PREFIX, MFMA, A, B, G)                                                                              \
  asm volatile(NOPS PREFIX MFMA " v[204:207], %1, %2, 0\n\t" G "v_add_f32 %0, v207, %3\n\t" NOPS                \
               : "=&v"(r) : "v"(A), "v"(B), "v"(1.0f)                                                          \
               : "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214",      \
                 "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223")
#define LADDER(PREFIX, MFMA, A, B)                                   \
  if (PAD == 7) Q(PREFIX, MFMA, A, B, GAP(6));                       \
  else if (PAD == 8) Q(PREFIX, MFMA, A, B, GAP(7));                  \
  else if (PAD == 9) Q(PREFIX, MFMA, A, B, GAP(8));                  \
  else if (PAD == 10) Q(PREFIX, MFMA, A, B, GAP(9));                 \
  else if (PAD == 12) Q(PREFIX, MFMA, A, B, GAP(11));                \
  else if (PAD == 16) Q(PREFIX, MFMA, A, B, GAP(15));                \
  else if (PAD == 24) Q(PREFIX, MFMA, A, B, GAP(15) GAP(7));         \
  else if (PAD == 32) Q(PREFIX, MFMA, A, B, GAP(15) GAP(15));        \
  else Q(PREFIX, MFMA, A, B, NOPS NOPS)
    const f16x4 alo = a.lo, blo = b.lo;
    if (K32) {
      if (PRE == 0) { LADDER("", "v_mfma_f32_16x16x32_f16", a, b); }
      else if (PRE == 4) { LADDER(PRE32, "v_mfma_f32_16x16x32_f16", a, b); }
      else { LADDER(PRE32 PRE32, "v_mfma_f32_16x16x32_f16", a, b); }
    } else {
      if (PRE == 0) { LADDER("", "v_mfma_f32_16x16x16_f16", alo, blo); }
      else if (PRE == 4) { LADDER(PRE16, "v_mfma_f32_16x16x16_f16", alo, blo); }
      else { LADDER(PRE16 PRE16, "v_mfma_f32_16x16x16_f16", alo, blo); }
    }
#undef LADDER
#undef Q
#undef PRE16
#undef PRE32
#undef GAP
#undef NOPS
    sum += r;
    a[it & 7] += (_Float16)0.001f;
  }
  out[(size_t)blockIdx.x * 256 + tid] = sum;
}
template <int PAD, int K32, int PRE>
static long queued_bad(const _Float16 *frag, float *out, const std::vector<float> &ref) {
  const size_t n = (size_t)1024 * 256;
  std::vector<float> cur(n);
  queued<PAD, K32, PRE><<<1024, 256>>>(frag, out, 300);
  hipDeviceSynchronize();
  hipMemcpy(cur.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  long bad = 0;
  for (size_t i = 0; i < n; ++i) bad += std::memcmp(&ref[i], &cur[i], 4) != 0;
  return bad;
}
template <int K32, int PRE>
static void run_queued(const char *name, const _Float16 *frag, float *out) {
  const size_t n = (size_t)1024 * 256;
  std::vector<float> ref(n);
  queued<200, K32, PRE><<<1024, 256>>>(frag, out, 300);
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  printf("  %s, %d MFMAs issued right before it -> v_add_f32 after N wait states  N=7: %ld  8: %ld  9: %ld  10: %ld  12: %ld  16: %ld  24: %ld  32: %ld\n",
         name, PRE, queued_bad<7, K32, PRE>(frag, out, ref), queued_bad<8, K32, PRE>(frag, out, ref),
         queued_bad<9, K32, PRE>(frag, out, ref), queued_bad<10, K32, PRE>(frag, out, ref),
         queued_bad<12, K32, PRE>(frag, out, ref), queued_bad<16, K32, PRE>(frag, out, ref),
         queued_bad<24, K32, PRE>(frag, out, ref), queued_bad<32, K32, PRE>(frag, out, ref));
}

// WAR: the MFMA reads v[204:207] as SrcC (or v[200:203] as SrcA); N wait states later a VALU move
// overwrites those registers.  The MFMA result must not change.
template <int PAD, int K32, int WHICH>
__global__ __launch_bounds__(256) void warprobe(const _Float16 *frag, float *out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f16x8 *src = reinterpret_cast<const f16x8 *>(frag) + (size_t)(blockIdx.x % 64) * 4096;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 x = *reinterpret_cast<const u32x4 *>(&src[(wave * 8 + 0) * 64 + lane]);
  f16x8 b1 = src[2048 + (wave * 8) * 64 + lane];
  f32x4 sum = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 d;
    const unsigned x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    const float c0 = 0.5f + it, c1 = 1.5f, c2 = 2.5f, c3 = 3.5f, junk = 1.0e6f;
#define NOPS "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define GAP(n) "s_nop " #n "\n\t"
#define CLOB(G, MFMA, AREG, BOP)                                                                                         \
  asm volatile("v_mov_b32 v200, %1\n\tv_mov_b32 v201, %2\n\tv_mov_b32 v202, %3\n\tv_mov_b32 v203, %4\n\t"               \
               "v_mov_b32 v204, %6\n\tv_mov_b32 v205, %7\n\tv_mov_b32 v206, %8\n\tv_mov_b32 v207, %9\n\t" NOPS           \
               MFMA " %0, " AREG ", %5, v[204:207]\n\t" G                                                               \
               "v_mov_b32 v204, %10\n\tv_mov_b32 v205, %10\n\tv_mov_b32 v206, %10\n\tv_mov_b32 v207, %10\n\t"           \
               "v_mov_b32 v200, %10\n\tv_mov_b32 v201, %10\n\tv_mov_b32 v202, %10\n\tv_mov_b32 v203, %10\n\t" NOPS       \
               : "=&v"(d) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(BOP), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(junk) \
               : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207")
    (void)WHICH;
    const f16x4 b1lo = b1.lo;
    if (K32) {
      if (PAD == 0) CLOB("", "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 1) CLOB(GAP(0), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 2) CLOB(GAP(1), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 3) CLOB(GAP(2), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 4) CLOB(GAP(3), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 5) CLOB(GAP(4), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 6) CLOB(GAP(5), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 8) CLOB(GAP(7), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else if (PAD == 12) CLOB(GAP(11), "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
      else CLOB(NOPS, "v_mfma_f32_16x16x32_f16", "v[200:203]", b1);
    } else {
      if (PAD == 0) CLOB("", "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 1) CLOB(GAP(0), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 2) CLOB(GAP(1), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 3) CLOB(GAP(2), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 4) CLOB(GAP(3), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 5) CLOB(GAP(4), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 6) CLOB(GAP(5), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 8) CLOB(GAP(7), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else if (PAD == 12) CLOB(GAP(11), "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
      else CLOB(NOPS, "v_mfma_f32_16x16x16_f16", "v[200:201]", b1lo);
    }
#undef CLOB
#undef GAP
#undef NOPS
    sum += d;
    x[it & 3] += 0x00010001u * (it & 1);
  }
  out[(size_t)blockIdx.x * 256 + tid] = sum[0] + sum[1] + sum[2] + sum[3];
}
template <int PAD, int K32>
static long war_bad(const _Float16 *frag, float *out, int blocks, int iters, const std::vector<float> &ref) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> cur(n);
  warprobe<PAD, K32, 0><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(cur.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  long bad = 0;
  for (size_t i = 0; i < n; ++i) bad += std::memcmp(&ref[i], &cur[i], 4) != 0;
  return bad;
}
template <int K32>
static void run_war(const char *name, const _Float16 *frag, float *out, int blocks, int iters) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> ref(n);
  warprobe<64, K32, 0><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  printf("  %s operands overwritten by v_mov after N wait states  N=0: %ld  1: %ld  2: %ld  3: %ld  4: %ld  5: %ld  6: %ld  8: %ld  12: %ld\n", name,
         war_bad<0, K32>(frag, out, blocks, iters, ref), war_bad<1, K32>(frag, out, blocks, iters, ref),
         war_bad<2, K32>(frag, out, blocks, iters, ref), war_bad<3, K32>(frag, out, blocks, iters, ref),
         war_bad<4, K32>(frag, out, blocks, iters, ref), war_bad<5, K32>(frag, out, blocks, iters, ref),
         war_bad<6, K32>(frag, out, blocks, iters, ref), war_bad<8, K32>(frag, out, blocks, iters, ref),
         war_bad<12, K32>(frag, out, blocks, iters, ref));
}

template <int PAD, int K32>
static long movfeed_bad(const _Float16 *frag, float *out, int blocks, int iters, const std::vector<float> &ref) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> cur(n);
  movfeed<PAD, K32><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(cur.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  long bad = 0;
  for (size_t i = 0; i < n; ++i) bad += std::memcmp(&ref[i], &cur[i], 4) != 0;
  return bad;
}
static void run_movfeed(const _Float16 *frag, float *out, int blocks, int iters) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> ref(n);
  movfeed<64, 1><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  printf("operand registers written by v_mov, then N wait states (s_nop N-1), then the MFMA that reads them;\n"
         "lanes (of %zu) whose result differs from the 64-wait-state build:\n", n);
  printf("  v_mfma_f32_16x16x32_f16  N=0: %ld  N=1: %ld  N=2: %ld  N=3: %ld  N=4: %ld  N=6: %ld  N=8: %ld  N=12: %ld  N=16: %ld\n",
         movfeed_bad<0, 1>(frag, out, blocks, iters, ref), movfeed_bad<1, 1>(frag, out, blocks, iters, ref),
         movfeed_bad<2, 1>(frag, out, blocks, iters, ref), movfeed_bad<3, 1>(frag, out, blocks, iters, ref),
         movfeed_bad<4, 1>(frag, out, blocks, iters, ref), movfeed_bad<6, 1>(frag, out, blocks, iters, ref),
         movfeed_bad<8, 1>(frag, out, blocks, iters, ref), movfeed_bad<12, 1>(frag, out, blocks, iters, ref),
         movfeed_bad<16, 1>(frag, out, blocks, iters, ref));
  movfeed<64, 0><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(ref.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  printf("  v_mfma_f32_16x16x16_f16  N=0: %ld  N=1: %ld  N=2: %ld  N=4: %ld\n", movfeed_bad<0, 0>(frag, out, blocks, iters, ref),
         movfeed_bad<1, 0>(frag, out, blocks, iters, ref), movfeed_bad<2, 0>(frag, out, blocks, iters, ref),
         movfeed_bad<4, 0>(frag, out, blocks, iters, ref));
}

template <int K32>
static void run_pattern(const char *name, const _Float16 *frag, float *out, int blocks, int iters) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> plain(n), padded(n);
  pattern<K32, 0><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(plain.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  pattern<K32, 1><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(padded.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  long bad = 0, nan = 0;
  for (size_t i = 0; i < n; ++i) {
    bad += std::memcmp(&plain[i], &padded[i], 4) != 0;
    nan += !(padded[i] == padded[i]);
  }
  printf("%-44s lanes where back-to-back differs from padded: %ld of %zu (non-finite: %ld)\n", name, bad, n, nan);
}

template <int K32>
static void run_chain(const char *name, const _Float16 *frag, float *out, int blocks, int iters) {
  const size_t n = (size_t)blocks * 256;
  std::vector<float> plain(n), padded(n);
  chain<K32, 0><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(plain.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  chain<K32, 1><<<blocks, 256>>>(frag, out, iters);
  hipDeviceSynchronize();
  hipMemcpy(padded.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
  long bad = 0;
  for (size_t i = 0; i < n; ++i) bad += std::memcmp(&plain[i], &padded[i], 4) != 0;
  printf("%-44s lanes where the plain chain differs from the padded one: %ld of %zu\n", name, bad, n);
}

template <int K32, int CORUN>
static void run(const char *name, const _Float16 *frag, float *out, size_t n, int blocks, int iters, int reps) {
  std::vector<float> first(n), cur(n);
  long bad_total = 0;
  for (int r = 0; r <= reps; ++r) {
    hipMemset(out, 0, n * sizeof(float));
    probe<K32, CORUN><<<blocks, 512>>>(frag, out, iters);
    hipDeviceSynchronize();
    hipMemcpy(cur.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
    if (r == 0) {
      first = cur;
      continue;
    }
    long bad = 0;
    for (size_t i = 0; i < n; ++i) bad += std::memcmp(&first[i], &cur[i], 4) != 0;
    bad_total += bad;
  }
  printf("%-44s lanes differing from the first launch, summed over %d launches: %ld of %zu\n", name, reps, bad_total,
         n * (size_t)reps);
}

int main() {
  const int blocks = 2048, iters = 2000, reps = 4;
  const size_t n = (size_t)blocks * 512;
  std::vector<_Float16> h((size_t)64 * 4096 * 8);
  unsigned state = 12345u;
  for (auto &v : h) {
    state = state * 1664525u + 1013904223u;
    v = (_Float16)(((int)(state >> 16) % 2001 - 1000) * 1e-3f);
  }
  _Float16 *frag;
  float *out;
  hipMalloc(&frag, h.size() * sizeof(_Float16));
  hipMalloc(&out, 2 * n * sizeof(float));
  hipMemcpy(frag, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
  run<0, 0>("K=16 x2 per product, MFMA waves only", frag, out, n, blocks, iters, reps);
  run<0, 1>("K=16 x2 per product, beside VALU waves", frag, out, n, blocks, iters, reps);
  run<1, 0>("K=32, MFMA waves only", frag, out, n, blocks, iters, reps);
  run<1, 1>("K=32, beside VALU waves", frag, out, n, blocks, iters, reps);
  run_chain<0>("K=16 chain through different registers", frag, out, 1024, 500);
  run_chain<1>("K=32 chain through different registers", frag, out, 1024, 500);
  run_pattern<0>("K=16 compiler pattern, inline asm", frag, out, 1024, 500);
  run_pattern<1>("K=32 compiler pattern, inline asm", frag, out, 1024, 500);
  run_movfeed(frag, out, 1024, 500);
  printf("MFMA result consumed by an LDS store; lanes (of 262144) that differ from the 64-wait-state build:\n");
  run_queued<0, 0>("16x16x16_f16", frag, out);
  run_queued<0, 4>("16x16x16_f16", frag, out);
  run_queued<0, 8>("16x16x16_f16", frag, out);
  run_queued<1, 0>("16x16x32_f16", frag, out);
  run_queued<1, 4>("16x16x32_f16", frag, out);
  run_queued<1, 8>("16x16x32_f16", frag, out);
  run_war<0>("16x16x16_f16", frag, out, 1024, 300);
  run_war<1>("16x16x32_f16", frag, out, 1024, 300);
  {  // the same consumers while a second wave per SIMD keeps the matrix pipe busy (N = wait states given)
    const size_t n = (size_t)1024 * 256;
    std::vector<float> ref(n);
    auto reference = [&](auto kernel) {
      kernel<<<1024, 256>>>(frag, out, 300);
      hipDeviceSynchronize();
      hipMemcpy(ref.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
    };
    reference(storeafter<64, 1, 1, 0>);
    printf("  16x16x32_f16 -> v_add_f32, matrix pipe shared with another wave issuing 16x16x32:  N=8: %ld  9: %ld  10: %ld  12: %ld  16: %ld\n",
           store_bad<8, 1, 1, 1>(frag, out, 1024, 300, ref), store_bad<9, 1, 1, 1>(frag, out, 1024, 300, ref),
           store_bad<10, 1, 1, 1>(frag, out, 1024, 300, ref), store_bad<12, 1, 1, 1>(frag, out, 1024, 300, ref),
           store_bad<16, 1, 1, 1>(frag, out, 1024, 300, ref));
    reference(storeafter<64, 0, 1, 0>);
    printf("  16x16x16_f16 -> v_add_f32, matrix pipe shared with another wave issuing 16x16x16:  N=7: %ld  8: %ld  9: %ld  10: %ld  12: %ld\n",
           store_bad<7, 0, 1, 2>(frag, out, 1024, 300, ref), store_bad<8, 0, 1, 2>(frag, out, 1024, 300, ref),
           store_bad<9, 0, 1, 2>(frag, out, 1024, 300, ref), store_bad<10, 0, 1, 2>(frag, out, 1024, 300, ref),
           store_bad<12, 0, 1, 2>(frag, out, 1024, 300, ref));
    reference(storeafter<64, 1, 0, 0>);
    printf("  16x16x32_f16 -> ds_write_b128, matrix pipe shared:  N=6: %ld  7: %ld  8: %ld  10: %ld  12: %ld\n",
           store_bad<6, 1, 0, 1>(frag, out, 1024, 300, ref), store_bad<7, 1, 0, 1>(frag, out, 1024, 300, ref),
           store_bad<8, 1, 0, 1>(frag, out, 1024, 300, ref), store_bad<10, 1, 0, 1>(frag, out, 1024, 300, ref),
           store_bad<12, 1, 0, 1>(frag, out, 1024, 300, ref));
  }
  run_store<0, 0>("16x16x16_f16 -> ds_write_b128", frag, out, 1024, 300);
  run_store<1, 0>("16x16x32_f16 -> ds_write_b128", frag, out, 1024, 300);
  run_store<0, 1>("16x16x16_f16 -> v_add_f32    ", frag, out, 1024, 300);
  run_store<1, 1>("16x16x32_f16 -> v_add_f32    ", frag, out, 1024, 300);
  run_store<0, 2>("16x16x16_f16 -> v_pk_add_f32 ", frag, out, 1024, 300);
  run_store<1, 2>("16x16x32_f16 -> v_pk_add_f32 ", frag, out, 1024, 300);
  return 0;
}
