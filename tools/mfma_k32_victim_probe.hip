// Probe (gfx950): does a wave that issues v_mfma_f32_16x16x32_f16 disturb the co-resident wave of its SIMD?
// (profiles/r03/determinism.txt: the fused EdgeBlock built on that instruction is irreproducible only when
// two workgroups share a CU, and what goes wrong is one float4 component in lanes 48-63 of the OTHER wave.)
// 512-thread workgroups, one per CU: waves 0-3 run split-f16 MFMA chains (K = 32 or K = 16 form), waves 4-7
// -- one beside each MFMA wave -- read a known pattern from LDS (ds_read_b128 / b64 / b32) or from global
// memory (dwordx4) over and over and compare every dword with what it must be.  Mismatches are counted per
// (lane quarter, dword of the access).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_k32_victim_probe.hip -o tools/mfma_k32_victim_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pattern(unsigned word) { return word * 2654435761u + 0x9e3779b9u; }

template <int K32>
__device__ __forceinline__ f32x4 chain(const f16x8 &a, const f16x8 &b, f32x4 acc) {
  if (K32) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(b.lo, a.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(b.hi, a.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, a.lo, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, a.hi, acc, 0, 0, 0);
}

// MODE 0: ds_read_b128, 1: two ds_read_b64, 2: four ds_read_b32, 3: global_load_dwordx4
template <int K32, int MODE>
__global__ __launch_bounds__(512) void probe(const unsigned *gtab, unsigned *hist, float *sink, int iters, int mfma_on) {
  __shared__ __attribute__((aligned(16))) unsigned tab[8192];  // 32 KB of pattern words
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) tab[i] = pattern(i);
  __syncthreads();
  if (wave < 4) {
    if (!mfma_on) return;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (_Float16)(0.01f * ((lane * 7 + j) % 13) - 0.05f);
      b[j] = (_Float16)(0.02f * ((lane * 3 + j) % 11) - 0.1f);
    }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = chain<K32>(a, b, acc[t]);
      if ((it & 7) == 7) {  // a pause now and then, as between the real kernel's MFMA phases
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] *= 0.5f;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    sink[(size_t)blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  // victim: every lane reads 16 bytes at a lane-dependent, iteration-dependent LDS (or global) offset
  unsigned bad[4] = {0, 0, 0, 0};
  const int v = tid - 256;  // 0..255
  for (int it = 0; it < iters * 6; ++it) {
    const unsigned w0 = (unsigned)(((v + it * 37) * 4) & 8188);  // word index of a 16-byte slot
    unsigned x[4];
    if (MODE == 0) {
      const u32x4 q = *reinterpret_cast<const volatile u32x4 *>(tab + w0);
      x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
    } else if (MODE == 1) {
      const u32x2 q0 = *reinterpret_cast<const volatile u32x2 *>(tab + w0);
      const u32x2 q1 = *reinterpret_cast<const volatile u32x2 *>(tab + w0 + 2);
      x[0] = q0.x; x[1] = q0.y; x[2] = q1.x; x[3] = q1.y;
    } else if (MODE == 2) {
      for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const volatile unsigned *>(tab + w0 + j);
    } else {
      const u32x4 q = *reinterpret_cast<const volatile u32x4 *>(gtab + w0);
      x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bad[j] += x[j] != pattern(w0 + j);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (bad[j]) atomicAdd(hist + (lane >> 4) * 4 + j, bad[j]);
}

template <int K32, int MODE>
void run(const char *what, const unsigned *gtab, unsigned *hist, float *sink, int blocks, int iters, int mfma_on) {
  hipMemset(hist, 0, 16 * sizeof(unsigned));
  for (int rep = 0; rep < 4; ++rep) probe<K32, MODE><<<blocks, 512>>>(gtab, hist, sink, iters, mfma_on);
  hipDeviceSynchronize();
  unsigned h[16];
  hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long total = 0;
  for (unsigned x : h) total += x;
  const double reads = 4.0 * blocks * 256.0 * iters * 6;
  printf("%-58s wrong dwords %10llu of %.2e accesses;  by lane quarter x dword:", what, total, reads);
  for (int q = 0; q < 4; ++q) printf("  q%d[%u %u %u %u]", q, h[q * 4], h[q * 4 + 1], h[q * 4 + 2], h[q * 4 + 3]);
  printf("\n");
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int blocks = prop.multiProcessorCount;
  std::vector<unsigned> t(8192);
  for (int i = 0; i < 8192; ++i) t[i] = (unsigned)i * 2654435761u + 0x9e3779b9u;
  unsigned *gtab, *hist;
  float *sink;
  hipMalloc(&gtab, 8192 * 4);
  hipMalloc(&hist, 16 * 4);
  hipMalloc(&sink, (size_t)blocks * 512 * 4);
  hipMemcpy(gtab, t.data(), 8192 * 4, hipMemcpyHostToDevice);
  printf("%s, %d CUs, %d MFMA iterations per launch, 4 launches per line\n", prop.gcnArchName, blocks, iters);
  run<1, 0>("no MFMA wave              | victim ds_read_b128", gtab, hist, sink, blocks, iters, 0);
  run<0, 0>("v_mfma_f32_16x16x16_f16   | victim ds_read_b128", gtab, hist, sink, blocks, iters, 1);
  run<1, 0>("v_mfma_f32_16x16x32_f16   | victim ds_read_b128", gtab, hist, sink, blocks, iters, 1);
  run<1, 1>("v_mfma_f32_16x16x32_f16   | victim ds_read_b64 x2", gtab, hist, sink, blocks, iters, 1);
  run<1, 2>("v_mfma_f32_16x16x32_f16   | victim ds_read_b32 x4", gtab, hist, sink, blocks, iters, 1);
  run<1, 3>("v_mfma_f32_16x16x32_f16   | victim global_load_dwordx4", gtab, hist, sink, blocks, iters, 1);
  run<0, 3>("v_mfma_f32_16x16x16_f16   | victim global_load_dwordx4", gtab, hist, sink, blocks, iters, 1);
  return 0;
}
