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
// itself does not depend on it.  Plans and work buffers are cached per (device, series length):
// creating three hipFFT plans costs ~50 ms, the transforms of a 10^4-step series ~0.1 ms.
// rn_md_raman_intensities_device takes the time series where the evaluator left it (HBM), so that
// only the intensities leave the GPU.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdint>
#include <list>
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

// plans + work buffers of one (device, series length); kept in a small most-recently-used cache
struct Plans {
  int device = -1;
  int64_t S = 0, L = 0;
  void *x = nullptr, *p = nullptr, *r = nullptr, *out = nullptr, *alpha = nullptr;
  hipfftHandle plan_batch = 0, plan_l = 0, plan_n = 0;
  bool have_batch = false, have_l = false, have_n = false;
  ~Plans() {
    FftApi &api = fft_api();
    if (api.ok) {
      if (have_batch) api.destroy(plan_batch);
      if (have_l) api.destroy(plan_l);
      if (have_n) api.destroy(plan_n);
    }
    for (void *q : {x, p, r, out, alpha})
      if (q) (void)hipFree(q);
  }
};
std::mutex g_cache_mutex;
std::list<Plans> g_cache;  // front = most recently used
constexpr size_t kCacheEntries = 4;

// returns the cache entry for (device, S), creating it if needed; rc != RN_OK on failure
int get_plans(int device, int64_t S, Plans **out) {
  for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
    if (it->device == device && it->S == S) {
      g_cache.splice(g_cache.begin(), g_cache, it);
      *out = &g_cache.front();
      return RN_OK;
    }
  FftApi &api = fft_api();
  const int64_t N = S - 1, bins = (N + 1) / 2 - 1;
  int64_t L = 1;
  while (L < 2 * N - 1) L <<= 1;
  g_cache.emplace_front();
  Plans &b = g_cache.front();
  b.device = device;
  b.S = S;
  b.L = L;
  const size_t cz = sizeof(hipfftDoubleComplex);
  int rc = RN_OK;
  if (hipMalloc(&b.x, (size_t)7 * L * cz) != hipSuccess || hipMalloc(&b.p, (size_t)L * cz) != hipSuccess ||
      hipMalloc(&b.r, (size_t)N * cz) != hipSuccess ||
      hipMalloc(&b.out, (size_t)(bins > 0 ? bins : 1) * sizeof(double)) != hipSuccess)
    rc = RN_ERR_OUT_OF_MEMORY;
  int nl = (int)L, nn = (int)N;
  if (rc == RN_OK) {
    b.have_batch = api.plan_many(&b.plan_batch, 1, &nl, nullptr, 1, nl, nullptr, 1, nl, HIPFFT_Z2Z, 7) == HIPFFT_SUCCESS;
    b.have_l = api.plan_many(&b.plan_l, 1, &nl, nullptr, 1, nl, nullptr, 1, nl, HIPFFT_Z2Z, 1) == HIPFFT_SUCCESS;
    b.have_n = api.plan_many(&b.plan_n, 1, &nn, nullptr, 1, nn, nullptr, 1, nn, HIPFFT_Z2Z, 1) == HIPFFT_SUCCESS;
    if (!(b.have_batch && b.have_l && b.have_n)) rc = RN_ERR_HIP;
  }
  if (rc != RN_OK) {
    g_cache.pop_front();
    return rc;
  }
  while (g_cache.size() > kCacheEntries) g_cache.pop_back();
  *out = &g_cache.front();
  return RN_OK;
}

// d_alpha: device float64[S][3][3]; result in b.out (device float64[bins]); null stream, synchronous
int reduce_on_device(Plans &b, const double *d_alpha) {
  FftApi &api = fft_api();
  const int64_t N = b.S - 1, L = b.L, bins = (N + 1) / 2 - 1;
  auto *x = static_cast<hipfftDoubleComplex *>(b.x), *p = static_cast<hipfftDoubleComplex *>(b.p),
       *r = static_cast<hipfftDoubleComplex *>(b.r);
  const unsigned gl = (unsigned)((L + 255) / 256), gn = (unsigned)((N + 255) / 256);
  build_signals_kernel<<<gl, 256>>>(d_alpha, N, L, x);
  if (api.exec_z2z(b.plan_batch, x, x, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return RN_ERR_HIP;
  power_kernel<<<gl, 256>>>(x, L, p);
  if (api.exec_z2z(b.plan_l, p, p, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) return RN_ERR_HIP;
  take_lags_kernel<<<gn, 256>>>(p, N, 1.0 / (double)L, r);
  if (api.exec_z2z(b.plan_n, r, r, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return RN_ERR_HIP;
  real_bins_kernel<<<(unsigned)((bins + 255) / 256), 256>>>(r, bins, static_cast<double *>(b.out));
  if (hipGetLastError() != hipSuccess) return RN_ERR_HIP;
  return RN_OK;
}

int check_args(const void *alpha, const void *intensities, int64_t S, int64_t num_bins, int device) {
  const int64_t N = S - 1;
  if (!alpha || !intensities || S < 3 || num_bins != (N + 1) / 2 - 1 || N > (int64_t)1 << 28)
    return RN_ERR_INVALID_ARGUMENT;
  if (!fft_api().ok) return RN_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RN_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return RN_ERR_HIP;
  return RN_OK;
}

}  // namespace

extern "C" int rn_md_raman_intensities(const double *alpha, int64_t S, int device, double *intensities,
                                       int64_t num_bins) {
  int rc = check_args(alpha, intensities, S, num_bins, device);
  if (rc != RN_OK || num_bins == 0) return rc;
  std::lock_guard<std::mutex> lock(g_cache_mutex);
  Plans *b = nullptr;
  if ((rc = get_plans(device, S, &b)) != RN_OK) return rc;
  if (!b->alpha && hipMalloc(&b->alpha, (size_t)S * 9 * sizeof(double)) != hipSuccess) return RN_ERR_OUT_OF_MEMORY;
  if (hipMemcpy(b->alpha, alpha, (size_t)S * 9 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return RN_ERR_HIP;
  if ((rc = reduce_on_device(*b, static_cast<const double *>(b->alpha))) != RN_OK) return rc;
  if (hipMemcpy(intensities, b->out, (size_t)num_bins * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return RN_ERR_HIP;
  return RN_OK;
}

extern "C" int rn_md_raman_intensities_device(const double *d_alpha, int64_t S, int device, double *intensities,
                                              int64_t num_bins, void *stream) {
  int rc = check_args(d_alpha, intensities, S, num_bins, device);
  if (rc != RN_OK || num_bins == 0) return rc;
  // the producer of d_alpha (the evaluator) ran on `stream`: the reduction runs on the null stream
  if (stream && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return RN_ERR_HIP;
  std::lock_guard<std::mutex> lock(g_cache_mutex);
  Plans *b = nullptr;
  if ((rc = get_plans(device, S, &b)) != RN_OK) return rc;
  if ((rc = reduce_on_device(*b, d_alpha)) != RN_OK) return rc;
  if (hipMemcpy(intensities, b->out, (size_t)num_bins * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return RN_ERR_HIP;
  return RN_OK;
}
