// Batch normalisation of the optional enable_BN branch.
//
// Reference: tf.compat.v1.layers.batch_normalization(x, momentum=0.95, epsilon=1e-4,
// training=is_train_stage) after every CIN layer's conv1d (xDeepFM.py:234-240, :394-400; x is
// [B, D, L], channels = L) and after every DNN layer's xw_plus_b (xDeepFM.py:497-503; x is
// [B, n]).  Rank-2/3 inputs take TF's non-fused path: batch moments with the biased variance
// (tf.nn.moments), y = (x - mean) * rsqrt(var + eps) * gamma + beta, moving statistics moved by
// variable -= (variable - value) * (1 - momentum) with the same biased variance; inference
// normalises with the moving statistics.
//
// One layout serves both call sites: x[(b*C + c)*I + i], channel c, I = 1 for the DNN ([B, n])
// and I = D for the CIN, whose curr lives as [B, L, D] here.  Not on any BASELINE configuration:
// plain VALU kernels, one block per channel for the reductions (fixed tree: deterministic).
#include <hip/hip_runtime.h>

#include "act.h"
#include "xdfm_common.h"

namespace {

template <int THREADS>
__device__ __forceinline__ float block_sum(float v, float* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ int64_t elem(int64_t j, int c, int C, int I) {
  const int64_t b = j / I;
  return (b * C + c) * I + (j - b * I);
}

// stats[c] = mean, stats[C + c] = rsqrt(var + eps); training also moves the moving statistics
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int64_t B, int C, int I,
                                                       float* __restrict__ moving_mean,
                                                       float* __restrict__ moving_var, float momentum,
                                                       float eps, float* __restrict__ stats) {
  __shared__ float sh[256];
  const int c = blockIdx.x;
  const int64_t n = B * I;
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < n; j += 256) s += x[elem(j, c, C, I)];
  const float mean = block_sum<256>(s, sh) / (float)n;
  float q = 0.f;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    const float d = x[elem(j, c, C, I)] - mean;
    q += d * d;
  }
  const float var = block_sum<256>(q, sh) / (float)n;
  if (threadIdx.x == 0) {
    stats[c] = mean;
    stats[C + c] = 1.f / sqrtf(var + eps);
    const float decay = 1.f - momentum;
    moving_mean[c] -= (moving_mean[c] - mean) * decay;
    moving_var[c] -= (moving_var[c] - var) * decay;
  }
}

__global__ void bn_moving_stats_kernel(const float* __restrict__ moving_mean,
                                       const float* __restrict__ moving_var, int C, float eps,
                                       float* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  stats[c] = moving_mean[c];
  stats[C + c] = 1.f / sqrtf(moving_var[c] + eps);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t total, int C, int I,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ stats, int act,
                                                       float* __restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)((idx / I) % C);
  const float xhat = (x[idx] - stats[c]) * stats[C + c];
  y[idx] = activate(xhat * gamma[c] + beta[c], act);
}

// dgamma[c] = sum g*xhat, dbeta[c] = sum g with g = dy * act'(y)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ dy, int64_t B, int C, int I,
                                                            const float* __restrict__ stats, int act,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  __shared__ float sh[256];
  const int c = blockIdx.x;
  const int64_t n = B * I;
  const float mean = stats[c], inv = stats[C + c];
  float sg = 0.f, sgx = 0.f;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    const int64_t e = elem(j, c, C, I);
    const float g = dy[e] * act_grad(y[e], act);
    sg += g;
    sgx += g * ((x[e] - mean) * inv);
  }
  sg = block_sum<256>(sg, sh);
  sgx = block_sum<256>(sgx, sh);
  if (threadIdx.x == 0) {
    dbeta[c] = sg;
    dgamma[c] = sgx;
  }
}

// dx = gamma * inv * (g - dbeta/n - xhat * dgamma/n); dx may alias dy
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* dy, int64_t total, int64_t n, int C,
                                                           int I, const float* __restrict__ gamma,
                                                           const float* __restrict__ stats, int act,
                                                           const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, float* dx) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)((idx / I) % C);
  const float inv = stats[C + c];
  const float xhat = (x[idx] - stats[c]) * inv;
  const float g = dy[idx] * act_grad(y[idx], act);
  const float rn = 1.f / (float)n;
  dx[idx] = gamma[c] * inv * (g - dbeta[c] * rn - xhat * (dgamma[c] * rn));
}

bool act_ok(int act) { return act >= XDFM_ACT_IDENTITY && act <= XDFM_ACT_ELU; }

}  // namespace

extern "C" int xdfm_bn_fwd(const float* x, int64_t B, int32_t C, int32_t I, const float* gamma,
                           const float* beta, float* moving_mean, float* moving_var, float momentum,
                           float eps, int32_t training, int32_t act, float* stats, float* y,
                           void* stream) {
  XDFM_REQUIRE(x && gamma && beta && moving_mean && moving_var && stats && y, "xdfm_bn_fwd: null argument");
  XDFM_REQUIRE(B >= 0 && C > 0 && I > 0, "xdfm_bn_fwd: bad sizes");
  XDFM_REQUIRE(act_ok(act), "xdfm_bn_fwd: unknown activation %d", act);
  XDFM_REQUIRE(!training || B > 0, "xdfm_bn_fwd: batch statistics of an empty batch");
  if (B == 0) return XDFM_OK;
  hipStream_t st = (hipStream_t)stream;
  if (training)
    XDFM_LAUNCH("bn_stats_kernel", st, bn_stats_kernel, dim3((unsigned)C), dim3(256), 0, x, B, C, I, moving_mean,
                moving_var, momentum, eps, stats);
  else
    XDFM_LAUNCH("bn_moving_stats_kernel", st, bn_moving_stats_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
                moving_mean, moving_var, C, eps, stats);
  const int64_t total = B * C * I;
  XDFM_LAUNCH("bn_apply_kernel", st, bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, x, total, C,
              I, gamma, beta, stats, act, y);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_bn_bwd(const float* x, const float* y, const float* dy, int64_t B, int32_t C, int32_t I,
                           const float* gamma, const float* stats, int32_t act, float* dx, float* dgamma,
                           float* dbeta, void* stream) {
  XDFM_REQUIRE(x && y && dy && gamma && stats && dx && dgamma && dbeta, "xdfm_bn_bwd: null argument");
  XDFM_REQUIRE(B > 0 && C > 0 && I > 0, "xdfm_bn_bwd: bad sizes");
  XDFM_REQUIRE(act_ok(act), "xdfm_bn_bwd: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  XDFM_LAUNCH("bn_bwd_reduce_kernel", st, bn_bwd_reduce_kernel, dim3((unsigned)C), dim3(256), 0, x, y, dy, B, C, I,
              stats, act, dgamma, dbeta);
  const int64_t total = B * C * I;
  XDFM_LAUNCH("bn_bwd_apply_kernel", st, bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, x, y,
              dy, total, B * I, C, I, gamma, stats, act, dgamma, dbeta, dx);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// ------------------------------------------------------- batch split between ranks
// The batch statistics of the reference are those of the WHOLE batch (one process there).  When
// the batch is split between ranks, each rank forms its own moments (xdfm_bn_local_moments), the
// ranks combine them (count-weighted means, M2 = sum M2_r + n_r (mean_r - mean)^2: the host side
// does that on 2C numbers per rank) and xdfm_bn_fwd_moments normalises with the result.  Backward
// likewise: xdfm_bn_bwd_sums gives this rank's sums (= its share of the gamma / beta gradients),
// the ranks add them, xdfm_bn_bwd_apply takes the totals and the total count.
namespace {

// moments[c] = mean of this rank's part, moments[C + c] = sum of squared deviations from it
__global__ __launch_bounds__(256) void bn_local_moments_kernel(const float* __restrict__ x, int64_t B, int C, int I,
                                                               float* __restrict__ moments) {
  __shared__ float sh[256];
  const int c = blockIdx.x;
  const int64_t n = B * I;
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < n; j += 256) s += x[elem(j, c, C, I)];
  const float mean = block_sum<256>(s, sh) / (float)n;
  float q = 0.f;
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    const float d = x[elem(j, c, C, I)] - mean;
    q += d * d;
  }
  q = block_sum<256>(q, sh);
  if (threadIdx.x == 0) {
    moments[c] = mean;
    moments[C + c] = q;
  }
}

// mean_var[c] = mean, mean_var[C + c] = biased variance of the whole batch
__global__ void bn_given_stats_kernel(const float* __restrict__ mean_var, float* __restrict__ moving_mean,
                                      float* __restrict__ moving_var, int C, float momentum, float eps,
                                      float* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = mean_var[c], var = mean_var[C + c];
  stats[c] = mean;
  stats[C + c] = 1.f / sqrtf(var + eps);
  const float decay = 1.f - momentum;
  moving_mean[c] -= (moving_mean[c] - mean) * decay;
  moving_var[c] -= (moving_var[c] - var) * decay;
}

}  // namespace

extern "C" int xdfm_bn_local_moments(const float* x, int64_t B, int32_t C, int32_t I, float* moments,
                                     void* stream) {
  XDFM_REQUIRE(x && moments, "xdfm_bn_local_moments: null argument");
  XDFM_REQUIRE(B > 0 && C > 0 && I > 0, "xdfm_bn_local_moments: bad sizes");
  XDFM_LAUNCH("bn_local_moments_kernel", (hipStream_t)stream, bn_local_moments_kernel, dim3((unsigned)C), dim3(256),
              0, x, B, C, I, moments);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_bn_fwd_moments(const float* x, int64_t B, int32_t C, int32_t I, const float* gamma,
                                   const float* beta, float* moving_mean, float* moving_var, float momentum,
                                   float eps, const float* mean_var, int32_t act, float* stats, float* y,
                                   void* stream) {
  XDFM_REQUIRE(x && gamma && beta && moving_mean && moving_var && mean_var && stats && y,
               "xdfm_bn_fwd_moments: null argument");
  XDFM_REQUIRE(B > 0 && C > 0 && I > 0, "xdfm_bn_fwd_moments: bad sizes");
  XDFM_REQUIRE(act_ok(act), "xdfm_bn_fwd_moments: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  XDFM_LAUNCH("bn_given_stats_kernel", st, bn_given_stats_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0,
              mean_var, moving_mean, moving_var, C, momentum, eps, stats);
  const int64_t total = B * C * I;
  XDFM_LAUNCH("bn_apply_kernel", st, bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, x, total, C,
              I, gamma, beta, stats, act, y);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_bn_bwd_sums(const float* x, const float* y, const float* dy, int64_t B, int32_t C, int32_t I,
                                const float* stats, int32_t act, float* dgamma, float* dbeta, void* stream) {
  XDFM_REQUIRE(x && y && dy && stats && dgamma && dbeta, "xdfm_bn_bwd_sums: null argument");
  XDFM_REQUIRE(B > 0 && C > 0 && I > 0, "xdfm_bn_bwd_sums: bad sizes");
  XDFM_REQUIRE(act_ok(act), "xdfm_bn_bwd_sums: unknown activation %d", act);
  XDFM_LAUNCH("bn_bwd_reduce_kernel", (hipStream_t)stream, bn_bwd_reduce_kernel, dim3((unsigned)C), dim3(256), 0, x, y,
              dy, B, C, I, stats, act, dgamma, dbeta);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_bn_bwd_apply(const float* x, const float* y, const float* dy, int64_t B, int32_t C, int32_t I,
                                 int64_t n_total, const float* gamma, const float* stats, int32_t act,
                                 const float* sum_gamma, const float* sum_beta, float* dx, void* stream) {
  XDFM_REQUIRE(x && y && dy && gamma && stats && sum_gamma && sum_beta && dx, "xdfm_bn_bwd_apply: null argument");
  XDFM_REQUIRE(B > 0 && C > 0 && I > 0 && n_total >= B * I, "xdfm_bn_bwd_apply: bad sizes");
  XDFM_REQUIRE(act_ok(act), "xdfm_bn_bwd_apply: unknown activation %d", act);
  const int64_t total = B * C * I;
  XDFM_LAUNCH("bn_bwd_apply_kernel", (hipStream_t)stream, bn_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)),
              dim3(256), 0, x, y, dy, total, n_total, C, I, gamma, stats, act, sum_gamma, sum_beta, dx);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// ------------------------------------------------------------------------ softmax
// `_activate(..., "softmax")` (base_model.py:343-344): tf.nn.softmax over the last axis -- the
// units of a DNN layer ([B, n]: C = n, I = 1) or the channels of a CIN layer's curr ([B, D, L]
// there, [B, L, D] here: C = L, I = D).  One lane per (b, i).
namespace {

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* x, int64_t rows, int C, int I, float* y) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (b, i)
  if (r >= rows) return;
  const int64_t b = r / I;
  const int64_t base = b * C * I + (r - b * I);
  float m = x[base];
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[base + (int64_t)c * I]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(x[base + (int64_t)c * I] - m);
  const float inv = 1.f / s;
  for (int c = 0; c < C; ++c) y[base + (int64_t)c * I] = expf(x[base + (int64_t)c * I] - m) * inv;
}

// dx = y * (dy - sum_c y*dy)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* dy,
                                                          int64_t rows, int C, int I, float* dx) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const int64_t b = r / I;
  const int64_t base = b * C * I + (r - b * I);
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot = fmaf(y[base + (int64_t)c * I], dy[base + (int64_t)c * I], dot);
  for (int c = 0; c < C; ++c) {
    const int64_t e = base + (int64_t)c * I;
    dx[e] = y[e] * (dy[e] - dot);
  }
}

}  // namespace

extern "C" int xdfm_softmax_fwd(const float* x, int64_t B, int32_t C, int32_t I, float* y, void* stream) {
  XDFM_REQUIRE(x && y && B >= 0 && C > 0 && I > 0, "xdfm_softmax_fwd: bad argument");
  if (B == 0) return XDFM_OK;
  const int64_t rows = B * I;
  XDFM_LAUNCH("softmax_fwd_kernel", (hipStream_t)stream, softmax_fwd_kernel, dim3((unsigned)((rows + 255) / 256)),
              dim3(256), 0, x, rows, C, I, y);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_softmax_bwd(const float* y, const float* dy, int64_t B, int32_t C, int32_t I, float* dx,
                                void* stream) {
  XDFM_REQUIRE(y && dy && dx && B >= 0 && C > 0 && I > 0, "xdfm_softmax_bwd: bad argument");
  if (B == 0) return XDFM_OK;
  const int64_t rows = B * I;
  XDFM_LAUNCH("softmax_bwd_kernel", (hipStream_t)stream, softmax_bwd_kernel, dim3((unsigned)((rows + 255) / 256)),
              dim3(256), 0, y, dy, rows, C, I, dx);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}
