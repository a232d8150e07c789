// Device-resident optimisation step (_train.py:63-76): the
// parameters, their gradients and the Adam moments stay in HBM in the kernels' own packed layout;
// a step uploads dL/d(vec6) (24 B per structure) and downloads nothing.
//   adam_kernel            torch.optim.Adam (no amsgrad; weight_decay added to the gradient) on the
//                          trainable entries of the packed blob
//   refresh_derived_kernel the entries of the blob that are functions of parameters: transposed
//                          copies for the reverse pass, the duplicate bias of readout Linear 0, the
//                          folded c3_norm_1 constants (node_table / scale0 / shift0: setup_kernel)
//   bn_running_kernel      BatchNorm1d running statistics, torch semantics (momentum 0.1, unbiased var)
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace rn {

__global__ void adam_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, const unsigned char *__restrict__ trainable, size_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                            float bc2_sqrt, const int *__restrict__ poisoned) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (poisoned: the role-specialised EdgeBlock's time-out word -- the gradients of such a step are NaN rows: nothing is applied,
  //  and rn_potgnn_adam_step reports the time-out after its synchronisation)
  if (i >= n || !trainable[i] || (poisoned && *poisoned != 0)) return;
  float grad = g[i];
  if (weight_decay != 0.0f) grad = fmaf(weight_decay, w[i], grad);
  const float mi = beta1 * m[i] + (1.0f - beta1) * grad;  // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = beta2 * v[i] + (1.0f - beta2) * grad * grad;
  m[i] = mi;
  v[i] = vi;
  // torch: denom = sqrt(v) / sqrt(1 - beta2^t) + eps;  w -= lr / (1 - beta1^t) * m / denom
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  w[i] -= (lr / bc1) * (mi / denom);
}

void launch_adam(float *w, const float *g, float *m, float *v, const unsigned char *trainable, size_t n,
                 double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                 hipStream_t st, const int *poisoned) {
  if (n == 0) return;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  adam_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(w, g, m, v, trainable, n, (float)lr, (float)beta1,
                                                          (float)beta2, (float)eps, (float)weight_decay,
                                                          (float)bc1, (float)sqrt(bc2), poisoned);
}

// Pieces of the packed blob the host looks at after a device-resident step (c3_norm_1's folded constants, the prescale /
// finiteness pairs, the readout block of the range guard) -> one contiguous staging buffer: one download instead of 2 P + 1
// small ones, each a round trip of its own.  seg[3 i .. 3 i + 2] = (source offset, staging offset, count).
__global__ void gather_segments_kernel(const float *__restrict__ w, const long long *__restrict__ seg, int nseg,
                                       float *__restrict__ staging) {
  for (int i = blockIdx.x; i < nseg; i += gridDim.x) {
    const long long src = seg[3 * i], dst = seg[3 * i + 1], n = seg[3 * i + 2];
    for (long long k = threadIdx.x; k < n; k += blockDim.x) staging[dst + k] = w[src + k];
  }
}
void launch_gather_segments(const float *w, const long long *seg, int nseg, float *staging, hipStream_t st) {
  if (nseg <= 0) return;
  gather_segments_kernel<<<(unsigned)nseg, 256, 0, st>>>(w, seg, nseg, staging);
}

// one entry per derived range of the packed blob
//   kind 0: dst[n*K + k] = src[k*N + n]   (transposed copy, src is [K][N])
//   kind 1: dst[i] = src[i] * scale       (count = K: copies with scale 1, folded LayerNorm halves)
//   kind 2: dst[0..1] = (s, 1/s), s = mfma_prescale(max |src[k * ld + n]|, k < K, n < N), ld = (int)scale:
//           the power-of-two prescale of a weight block for the split-f16 matrix products
//   kind 3: row-centred copy of a [K][N] matrix of [filter | core] column blocks (kernels.hpp)
__global__ void refresh_derived_kernel(float *__restrict__ w, const DerivedOp *__restrict__ ops, int num_ops) {
  __shared__ float s_max[256];
  for (int o = blockIdx.x; o < num_ops; o += gridDim.x) {
    const DerivedOp op = ops[o];
    if (op.kind == 2) {  // (uniform over the block)
      const int ld = (int)op.scale;
      float mx = 0.0f;
      for (size_t i = threadIdx.x; i < (size_t)op.K * op.N; i += blockDim.x) {
        const float v = fabsf(w[op.src + (i / op.N) * ld + i % op.N]);
        mx = (v <= 3.0e38f) ? fmaxf(mx, v) : __builtin_nanf("");  // a non-finite weight poisons the maximum
      }
      s_max[threadIdx.x] = mx;
      __syncthreads();
      for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
          const float a = s_max[threadIdx.x], b = s_max[threadIdx.x + st];
          s_max[threadIdx.x] = (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b);
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        // (s, 1/s) = (NaN, NaN) marks a block with a non-finite entry: the host's range guard (api.hip:
        // mfma_f16_range_ok) reads the pairs back after a device-resident step and falls back to exact f32
        const float m = s_max[0];
        const float sc = (m != m) ? m : mfma_prescale(m);
        w[op.dst] = sc;
        w[op.dst + 1] = 1.0f / sc;
      }
      __syncthreads();
      continue;
    }
    if (op.kind == 3) {  // one WAVE per (row, [filter | core] block): lanes own columns c, c + 64, ... of both halves
      const int bw = 2 * op.FeP, nblk = op.N / bw, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
      for (int i = wave; i < op.K * nblk; i += nwaves) {
        const size_t base = (size_t)(i / nblk) * op.N + (size_t)(i % nblk) * bw;
        float sum = 0.0f;
        for (int c = lane; c < op.Fe; c += 64) sum += w[op.src + base + c] + w[op.src + base + op.FeP + c];
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum / (float)(2 * op.Fe);
        for (int c = lane; c < op.FeP; c += 64) {
          w[op.dst + base + c] = c < op.Fe ? w[op.src + base + c] - mean : 0.0f;
          w[op.dst + base + op.FeP + c] = c < op.Fe ? w[op.src + base + op.FeP + c] - mean : 0.0f;
        }
      }
      continue;
    }
    const size_t total = (size_t)op.K * (op.kind == 0 ? op.N : 1);
    for (size_t i = threadIdx.x; i < total; i += blockDim.x) {
      if (op.kind == 0) {
        const size_t k = i / op.N, n = i % op.N;
        w[op.dst + n * op.K + k] = w[op.src + i];
      } else {
        w[op.dst + i] = w[op.src + i] * op.scale;
      }
    }
  }
}
void launch_refresh_derived(float *w, const DerivedOp *ops, int num_ops, hipStream_t st) {
  if (num_ops == 0) return;
  refresh_derived_kernel<<<num_ops, 256, 0, st>>>(w, ops, num_ops);
}

__global__ void bn_running_kernel(float *__restrict__ running_mean, float *__restrict__ running_var,
                                  const float *__restrict__ batch_mean, const float *__restrict__ batch_var,
                                  int F, float momentum, float unbias) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= F) return;
  running_mean[k] = (1.0f - momentum) * running_mean[k] + momentum * batch_mean[k];
  running_var[k] = (1.0f - momentum) * running_var[k] + momentum * (batch_var[k] * unbias);
}
void launch_bn_running(float *running_mean, float *running_var, const float *batch_mean, const float *batch_var,
                       int F, double momentum, double unbias, hipStream_t st) {
  bn_running_kernel<<<(F + 63) / 64, 64, 0, st>>>(running_mean, running_var, batch_mean, batch_var, F,
                                                  (float)momentum, (float)unbias);
}

}  // namespace rn
