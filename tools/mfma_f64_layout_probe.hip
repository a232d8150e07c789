// Prints which (row m, column n) of D = A B every (lane, register) of v_mfma_f64_16x16x4_f64 holds, and which (m, k) / (k, n)
// the lane's A / B operands are: D[m][n] = 100 m + n is produced with A[m][k] = (k == 0) * (100 m + 1000 k_tag), etc.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out, int mode) {
  const int lane = threadIdx.x;
  double a, b;
  if (mode == 0) {        // assume A: m = lane % 16, k = lane / 16; B: k = lane / 16, n = lane % 16
    a = (lane / 16 == 0) ? (double)(lane % 16) : 0.0;  // A[m][0] = m
    b = (lane / 16 == 0) ? 1.0 : 0.0;                  // B[0][n] = 1   -> D[m][n] = m
  } else if (mode == 1) {
    a = (lane / 16 == 0) ? 1.0 : 0.0;                  // A[m][0] = 1
    b = (lane / 16 == 0) ? (double)(lane % 16) : 0.0;  // B[0][n] = n   -> D[m][n] = n
  } else {                // which k does lane / 16 select: A[m][k] = 1 for all, B[k][n] = 10^k
    a = 1.0;
    const double p10[4] = {1.0, 10.0, 100.0, 1000.0};
    b = p10[lane / 16];                                // D[m][n] = 1111 if every k pairs with itself
  }
  f64x4 c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = c[j];
}
int main() {
  double *d;
  hipMalloc(&d, 64 * 4 * sizeof(double));
  double h[256];
  for (int mode = 0; mode < 3; ++mode) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (0: value = m, 1: value = n, 2: 1111 expected)\n", mode);
    for (int lane = 0; lane < 64; lane += (mode == 2 ? 21 : 1)) {
      if (mode != 2 && !(lane % 16 == 0 || lane % 16 == 5)) continue;
      printf("  lane %2d: %g %g %g %g\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    }
  }
  return 0;
}
