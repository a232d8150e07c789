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
  hipMalloc(&out, n * sizeof(float));
  hipMemcpy(frag, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
  run<0, 0>("K=16 x2 per product, MFMA waves only", frag, out, n, blocks, iters, reps);
  run<0, 1>("K=16 x2 per product, beside VALU waves", frag, out, n, blocks, iters, reps);
  run<1, 0>("K=32, MFMA waves only", frag, out, n, blocks, iters, reps);
  run<1, 1>("K=32, beside VALU waves", frag, out, n, blocks, iters, reps);
  run_chain<0>("K=16 chain through different registers", frag, out, 1024, 500);
  run_chain<1>("K=32 chain through different registers", frag, out, 1024, 500);
  return 0;
}
