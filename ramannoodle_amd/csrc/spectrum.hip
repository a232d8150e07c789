// On-device reduction of a polarizability time series to the unpolarised MD Raman spectrum
// (SURVEY.md 8f item 3): MDRamanSpectrum.measure before the laser / Bose-Einstein corrections
// (ramannoodle/spectrum/_raman.py:282-297 with calc_signal_spectrum, spectrum/utils.py:76-124).
//
// The reference forms d(alpha)/dt, seven signals from it, the positive-lag autocorrelation of
// each (scipy.signal.correlate) and the real part of the length-(S-1) FFT of each, and adds
// them as 45 a^2 + 7 g^2.  Everything after the autocorrelation is linear, and by
// Wiener-Khinchin the autocorrelations are inverse transforms of power spectra, so here:
//   7 zero-padded forward FFTs (one batched launch) -> ONE weighted power spectrum
//   5|X_iso|^2 + 3.5(|X_1|^2+|X_2|^2+|X_3|^2) + 21(|X_xy|^2+|X_yz|^2+|X_xz|^2)
//   -> one inverse FFT (the combined autocorrelation) -> one length-(S-1) FFT.
// float64 throughout.  The transforms are hipFFT's, loaded on first use so that the library
// itself does not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdint>
#include <mutex>

#include "../../include/rn_potgnn.h"

namespace {

struct FftApi {
  void *lib = nullptr;
  hipfftResult (*plan_many)(hipfftHandle *, int, int *, int *, int, int, int *, int, int, hipfftType, int) = nullptr;
  hipfftResult (*exec_z2z)(hipfftHandle, hipfftDoubleComplex *, hipfftDoubleComplex *, int) = nullptr;
  hipfftResult (*destroy)(hipfftHandle) = nullptr;
  bool ok = false;
};
FftApi &fft_api() {
  static FftApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"libhipfft.so", "libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    api.plan_many = reinterpret_cast<decltype(api.plan_many)>(dlsym(api.lib, "hipfftPlanMany"));
    api.exec_z2z = reinterpret_cast<decltype(api.exec_z2z)>(dlsym(api.lib, "hipfftExecZ2Z"));
    api.destroy = reinterpret_cast<decltype(api.destroy)>(dlsym(api.lib, "hipfftDestroy"));
    api.ok = api.plan_many && api.exec_z2z && api.destroy;
  });
  return api;
}

// x_j[n] from alpha[n+1] - alpha[n]; signals: trace, xx-yy, yy-zz, zz-xx, xy, yz, xz
__global__ void build_signals_kernel(const double *__restrict__ alpha, int64_t N, int64_t L,
                                     hipfftDoubleComplex *__restrict__ x) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= L) return;
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    const double *a0 = alpha + n * 9, *a1 = a0 + 9;
    const double xx = a1[0] - a0[0], yy = a1[4] - a0[4], zz = a1[8] - a0[8];
    const double xy = a1[1] - a0[1], yz = a1[5] - a0[5], xz = a1[2] - a0[2];
    s[0] = xx + yy + zz;
    s[1] = xx - yy;
    s[2] = yy - zz;
    s[3] = zz - xx;
    s[4] = xy;
    s[5] = yz;
    s[6] = xz;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) x[(int64_t)j * L + n] = make_double2(s[j], 0.0);
}
// 45 (1/9) |iso|^2 + 7 (1/2 (...) + 3 (...))
__global__ void power_kernel(const hipfftDoubleComplex *__restrict__ x, int64_t L,
                             hipfftDoubleComplex *__restrict__ p) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= L) return;
  auto mag2 = [&](int j) {
    const hipfftDoubleComplex v = x[(int64_t)j * L + f];
    return v.x * v.x + v.y * v.y;
  };
  const double v = 5.0 * mag2(0) + 3.5 * (mag2(1) + mag2(2) + mag2(3)) + 21.0 * (mag2(4) + mag2(5) + mag2(6));
  p[f] = make_double2(v, 0.0);
}
__global__ void take_lags_kernel(const hipfftDoubleComplex *__restrict__ r, int64_t N, double scale,
                                 hipfftDoubleComplex *__restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < N) out[k] = make_double2(r[k].x * scale, 0.0);
}
__global__ void real_bins_kernel(const hipfftDoubleComplex *__restrict__ y, int64_t bins, double *__restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m < bins) out[m] = y[m + 1].x;  // the zero-frequency bin is dropped
}

struct Bufs {
  void *alpha = nullptr, *x = nullptr, *p = nullptr, *r = nullptr, *out = nullptr;
  hipfftHandle plan_batch = nullptr, plan_l = nullptr, plan_n = nullptr;
  ~Bufs() {
    FftApi &api = fft_api();
    for (hipfftHandle h : {plan_batch, plan_l, plan_n})
      if (h && api.ok) api.destroy(h);
    for (void *q : {alpha, x, p, r, out})
      if (q) (void)hipFree(q);
  }
};

}  // namespace

extern "C" int rn_md_raman_intensities(const double *alpha, int64_t S, int device, double *intensities,
                                       int64_t num_bins) {
  const int64_t N = S - 1;
  if (!alpha || !intensities || S < 3 || num_bins != (N + 1) / 2 - 1 || N > (int64_t)1 << 28)
    return RN_ERR_INVALID_ARGUMENT;
  FftApi &api = fft_api();
  if (!api.ok) return RN_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RN_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return RN_ERR_HIP;
  if (num_bins == 0) return RN_OK;
  int64_t L = 1;
  while (L < 2 * N - 1) L <<= 1;
  Bufs b;
  const size_t cz = sizeof(hipfftDoubleComplex);
  if (hipMalloc(&b.alpha, (size_t)S * 9 * sizeof(double)) != hipSuccess || hipMalloc(&b.x, (size_t)7 * L * cz) != hipSuccess ||
      hipMalloc(&b.p, (size_t)L * cz) != hipSuccess || hipMalloc(&b.r, (size_t)N * cz) != hipSuccess ||
      hipMalloc(&b.out, (size_t)num_bins * sizeof(double)) != hipSuccess)
    return RN_ERR_OUT_OF_MEMORY;
  if (hipMemcpy(b.alpha, alpha, (size_t)S * 9 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return RN_ERR_HIP;
  int nl = (int)L, nn = (int)N;
  if (api.plan_many(&b.plan_batch, 1, &nl, nullptr, 1, nl, nullptr, 1, nl, HIPFFT_Z2Z, 7) != HIPFFT_SUCCESS ||
      api.plan_many(&b.plan_l, 1, &nl, nullptr, 1, nl, nullptr, 1, nl, HIPFFT_Z2Z, 1) != HIPFFT_SUCCESS ||
      api.plan_many(&b.plan_n, 1, &nn, nullptr, 1, nn, nullptr, 1, nn, HIPFFT_Z2Z, 1) != HIPFFT_SUCCESS)
    return RN_ERR_HIP;
  auto *x = static_cast<hipfftDoubleComplex *>(b.x), *p = static_cast<hipfftDoubleComplex *>(b.p),
       *r = static_cast<hipfftDoubleComplex *>(b.r);
  const unsigned gl = (unsigned)((L + 255) / 256), gn = (unsigned)((N + 255) / 256);
  build_signals_kernel<<<gl, 256>>>(static_cast<const double *>(b.alpha), N, L, x);
  if (api.exec_z2z(b.plan_batch, x, x, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return RN_ERR_HIP;
  power_kernel<<<gl, 256>>>(x, L, p);
  if (api.exec_z2z(b.plan_l, p, p, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) return RN_ERR_HIP;
  take_lags_kernel<<<gn, 256>>>(p, N, 1.0 / (double)L, r);
  if (api.exec_z2z(b.plan_n, r, r, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return RN_ERR_HIP;
  real_bins_kernel<<<(unsigned)((num_bins + 255) / 256), 256>>>(r, num_bins, static_cast<double *>(b.out));
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return RN_ERR_HIP;
  if (hipMemcpy(intensities, b.out, (size_t)num_bins * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return RN_ERR_HIP;
  return RN_OK;
}
