#include <hip/hip_runtime.h>
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, const int* idx) {
  __shared__ __attribute__((aligned(16))) float tile[16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = 4 * wave + lane / 16, p = lane % 16;
  const float* g = src + (size_t)idx[row] * 64 + ((p ^ row) & 15) * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
      (__attribute__((address_space(3))) void*)(tile + wave * 256), 16, 0, 0);
  __syncthreads();
  // read back row m piece q
  const int m = threadIdx.x / 16, q = threadIdx.x % 16;
  const float4 v = *reinterpret_cast<const float4*>(tile + m * 64 + ((q ^ m) & 15) * 4);
  *reinterpret_cast<float4*>(dst + (size_t)m * 64 + q * 4) = v;
}
int main() {
  float *s, *d; int *i;
  hipMalloc(&s, 100 * 64 * 4); hipMalloc(&d, 16 * 64 * 4); hipMalloc(&i, 64);
  float hs[100 * 64]; for (int j = 0; j < 6400; ++j) hs[j] = j; int hi[16]; for (int j = 0; j < 16; ++j) hi[j] = (j * 7 + 3) % 100;
  hipMemcpy(s, hs, sizeof(hs), hipMemcpyHostToDevice); hipMemcpy(i, hi, sizeof(hi), hipMemcpyHostToDevice);
  k<<<1, 256>>>(s, d, i);
  float hd[16 * 64]; hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
  int bad = 0; for (int m = 0; m < 16; ++m) for (int c = 0; c < 64; ++c) bad += hd[m * 64 + c] != hs[hi[m] * 64 + c];
  printf("bad=%d\n", bad); return bad != 0;
}
