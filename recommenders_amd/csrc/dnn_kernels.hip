// Dense layers of the DNN tower and the two output heads, fp32, gfx950.
//
// Reference: xDeepFM.py:473-533 (`xw_plus_b` + `_active_layer`, base_model.py:324-353) and
// the CIN head xDeepFM.py:279-293.  The notebook/Criteo-shape towers are tiny (390x20,
// 20x20, 20x1): memory/launch bound, so the fp32 form is VALU with the weight row held in
// SGPRs (one lane per sample, x staged through LDS so that the global read is coalesced).
// Weight gradients are reduced over the batch in a fixed split order (no float atomics).
//
// Layouts: x[B][K], w[K][Nout], b[Nout], y[B][Nout].
#include <hip/hip_runtime.h>

#include "xdfm_common.h"

namespace {

constexpr int kT = 128;   // samples per block in the forward
constexpr int kKT = 32;   // k-tile staged per step
constexpr int kMaxSplits = 128;

__device__ __forceinline__ float activate(float v, int act) {
  switch (act) {
    case XDFM_ACT_RELU: return v > 0.f ? v : 0.f;
    case XDFM_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case XDFM_ACT_TANH: return tanhf(v);
    case XDFM_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

// derivative expressed through the saved output y = act(z)
__device__ __forceinline__ float act_grad(float y, int act) {
  switch (act) {
    case XDFM_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case XDFM_ACT_SIGMOID: return y * (1.f - y);
    case XDFM_ACT_TANH: return 1.f - y * y;
    case XDFM_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    default: return 1.f;
  }
}

template <int NCH>
__global__ __launch_bounds__(kT) void dense_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, int64_t B,
                                                        int K, int Nout, int act,
                                                        float* __restrict__ y) {
  __shared__ float xs[kKT][kT + 1];
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kT;
  const int64_t b = b0 + tid;
  const int c0 = blockIdx.y * NCH;
  const bool full = (c0 + NCH <= Nout);
  float acc[NCH];
#pragma unroll
  for (int n = 0; n < NCH; ++n) acc[n] = 0.f;
  for (int k0 = 0; k0 < K; k0 += kKT) {
    __syncthreads();
    for (int idx = tid; idx < kT * kKT; idx += kT) {
      const int bl = idx / kKT;
      const int kk = idx - bl * kKT;
      const int64_t bb = b0 + bl;
      const int k = k0 + kk;
      xs[kk][bl] = (bb < B && k < K) ? x[bb * K + k] : 0.f;
    }
    __syncthreads();
    const int kn = (K - k0 < kKT) ? (K - k0) : kKT;
    for (int kk = 0; kk < kn; ++kk) {
      const float xv = xs[kk][tid];
      const float* __restrict__ wrow = w + (int64_t)(k0 + kk) * Nout + c0;
      if (full) {
#pragma unroll
        for (int n = 0; n < NCH; ++n) acc[n] = fmaf(xv, wrow[n], acc[n]);
      } else {
#pragma unroll
        for (int n = 0; n < NCH; ++n)
          if (c0 + n < Nout) acc[n] = fmaf(xv, wrow[n], acc[n]);
      }
    }
  }
  if (b >= B) return;
#pragma unroll
  for (int n = 0; n < NCH; ++n)
    if (c0 + n < Nout) y[b * Nout + c0 + n] = activate(acc[n] + bias[c0 + n], act);
}

__global__ void dense_dz_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                int64_t total, int act, float* __restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) dz[i] = dy[i] * act_grad(y[i], act);
}

// lane per k (k == K is the bias row, x = 1); registers hold dW[k, c0..c0+NCH)
template <int NCH>
__global__ __launch_bounds__(256) void dense_bwd_dw_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dz, int64_t B,
                                                           int K, int Nout, int samples_per_split,
                                                           float* __restrict__ part) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > K) return;
  const int c0 = blockIdx.z * NCH;
  const bool full = (c0 + NCH <= Nout);
  float acc[NCH];
#pragma unroll
  for (int n = 0; n < NCH; ++n) acc[n] = 0.f;
  const int64_t s_beg = (int64_t)blockIdx.y * samples_per_split;
  int64_t s_end = s_beg + samples_per_split;
  if (s_end > B) s_end = B;
  for (int64_t b = s_beg; b < s_end; ++b) {
    const float xv = (k < K) ? x[b * K + k] : 1.f;
    const float* __restrict__ dzrow = dz + b * Nout + c0;
    if (full) {
#pragma unroll
      for (int n = 0; n < NCH; ++n) acc[n] = fmaf(xv, dzrow[n], acc[n]);
    } else {
#pragma unroll
      for (int n = 0; n < NCH; ++n)
        if (c0 + n < Nout) acc[n] = fmaf(xv, dzrow[n], acc[n]);
    }
  }
  float* dst = part + ((int64_t)blockIdx.y * (K + 1) + k) * Nout + c0;
#pragma unroll
  for (int n = 0; n < NCH; ++n)
    if (c0 + n < Nout) dst[n] = acc[n];
}

__global__ void dense_dw_reduce_kernel(const float* __restrict__ part, int n_splits, int K, int Nout,
                                       float* __restrict__ dw, float* __restrict__ db) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)(K + 1) * Nout;
  if (idx >= total) return;
  float s = 0.f;
  for (int k = 0; k < n_splits; ++k) s += part[(int64_t)k * total + idx];
  if (idx < (int64_t)K * Nout)
    dw[idx] = s;
  else
    db[idx - (int64_t)K * Nout] = s;
}

__global__ void dense_bwd_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                    int64_t B, int K, int Nout, int accumulate,
                                    float* __restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * K) return;
  const int64_t b = idx / K;
  const int k = (int)(idx - b * K);
  const float* __restrict__ dzrow = dz + b * Nout;
  const float* __restrict__ wrow = w + (int64_t)k * Nout;
  float s = 0.f;
  for (int n = 0; n < Nout; ++n) s = fmaf(dzrow[n], wrow[n], s);
  dx[idx] = accumulate ? dx[idx] + s : s;
}

// exFM_out = base_score + xw_plus_b(result)   (xDeepFM.py:272-292)
__global__ void cin_head_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w,
                                    const float* __restrict__ bias, int64_t B, int n,
                                    float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* __restrict__ r = pooled + b * n;
  float base = 0.f, dot = 0.f;
  for (int k = 0; k < n; ++k) {
    base += r[k];
    dot = fmaf(r[k], w[k], dot);
  }
  out[b] = base + (dot + bias[0]);
}

__global__ void cin_head_dpooled_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                        int64_t B, int n, float* __restrict__ d_pooled) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * n) return;
  const int64_t b = idx / n;
  const int k = (int)(idx - b * n);
  d_pooled[idx] = g[b] + g[b] * w[k];
}

int pick_nch(int Nout) {
  if (Nout <= 1) return 1;
  if (Nout <= 2) return 2;
  if (Nout <= 4) return 4;
  return 8;
}

int splits_for(int64_t B) {
  int64_t s = (B + 31) / 32;
  return (int)(s < 1 ? 1 : (s > kMaxSplits ? kMaxSplits : s));
}

bool act_ok(int act) { return act >= XDFM_ACT_IDENTITY && act <= XDFM_ACT_ELU; }

}  // namespace

#define DISPATCH_NCH(nch, CALL) \
  switch (nch) {                \
    case 1: { CALL(1); } break; \
    case 2: { CALL(2); } break; \
    case 4: { CALL(4); } break; \
    default: { CALL(8); } break; \
  }

extern "C" int xdfm_dense_fwd(const float* x, const float* w, const float* b, int64_t B, int32_t K,
                              int32_t Nout, int32_t act, float* y, void* stream) {
  XDFM_REQUIRE(x && w && b && y, "xdfm_dense_fwd: null argument");
  XDFM_REQUIRE(B >= 0 && K > 0 && Nout > 0, "xdfm_dense_fwd: bad sizes");
  XDFM_REQUIRE(act_ok(act), "xdfm_dense_fwd: unknown activation %d", act);
  if (B == 0) return XDFM_OK;
  const int nch = pick_nch(Nout);
  dim3 grid((unsigned)((B + kT - 1) / kT), (unsigned)((Nout + nch - 1) / nch));
#define CALL(N)                                                                                \
  XDFM_LAUNCH("dense_fwd_kernel<" #N ">", stream, dense_fwd_kernel<N>, grid, dim3(kT), 0, x, w, b, B, K, \
                     Nout, act, y)
  DISPATCH_NCH(nch, CALL);
#undef CALL
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_dense_bwd_workspace_floats(int64_t B, int32_t K, int32_t Nout, size_t* floats) {
  XDFM_REQUIRE(floats && B >= 0 && K > 0 && Nout > 0, "xdfm_dense_bwd_workspace_floats: bad argument");
  *floats = (size_t)B * Nout + (size_t)splits_for(B) * (K + 1) * Nout;
  return XDFM_OK;
}

extern "C" int xdfm_dense_bwd(const float* x, const float* w, const float* y, const float* dy,
                              int64_t B, int32_t K, int32_t Nout, int32_t act, float* dx,
                              int32_t accumulate_dx, float* dw, float* db, float* part,
                              void* stream) {
  XDFM_REQUIRE(x && w && y && dy && dw && db && part, "xdfm_dense_bwd: null argument");
  XDFM_REQUIRE(B >= 0 && K > 0 && Nout > 0, "xdfm_dense_bwd: bad sizes");
  XDFM_REQUIRE(act_ok(act), "xdfm_dense_bwd: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    XDFM_HIP(hipMemsetAsync(dw, 0, (size_t)K * Nout * sizeof(float), st));
    XDFM_HIP(hipMemsetAsync(db, 0, (size_t)Nout * sizeof(float), st));
    return XDFM_OK;
  }
  float* dz = part;
  float* psum = part + (size_t)B * Nout;
  const int64_t total = B * Nout;
  XDFM_LAUNCH("dense_dz_kernel", st, dense_dz_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, y, dy,
                     total, act, dz);
  const int n_splits = splits_for(B);
  const int samples_per_split = (int)((B + n_splits - 1) / n_splits);
  const int nch = Nout <= 4 ? pick_nch(Nout) : (Nout <= 8 ? 8 : 8);
  {
    dim3 grid((unsigned)((K + 1 + 255) / 256), (unsigned)n_splits, (unsigned)((Nout + nch - 1) / nch));
#define CALL(N)                                                                              \
  XDFM_LAUNCH("dense_bwd_dw_kernel<" #N ">", st, dense_bwd_dw_kernel<N>, grid, dim3(256), 0, x, dz, B, K, Nout,       \
                     samples_per_split, psum)
    DISPATCH_NCH(nch, CALL);
#undef CALL
  }
  {
    const int64_t tot = (int64_t)(K + 1) * Nout;
    XDFM_LAUNCH("dense_dw_reduce_kernel", st, dense_dw_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       psum, n_splits, K, Nout, dw, db);
  }
  if (dx) {
    const int64_t tot = B * K;
    XDFM_LAUNCH("dense_bwd_dx_kernel", st, dense_bwd_dx_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, dz,
                       w, B, K, Nout, accumulate_dx, dx);
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_head_fwd(const float* pooled, const float* w, const float* bias, int64_t B,
                                 int32_t n, float* out, void* stream) {
  XDFM_REQUIRE(pooled && w && bias && out, "xdfm_cin_head_fwd: null argument");
  XDFM_REQUIRE(B >= 0 && n > 0, "xdfm_cin_head_fwd: bad sizes");
  if (B == 0) return XDFM_OK;
  XDFM_LAUNCH("cin_head_fwd_kernel", stream, cin_head_fwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, pooled, w, bias, B, n, out);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_head_bwd(const float* pooled, const float* w, const float* g, int64_t B,
                                 int32_t n, float* d_pooled, float* dw, float* db, float* part,
                                 void* stream) {
  XDFM_REQUIRE(pooled && w && g && d_pooled && dw && db && part, "xdfm_cin_head_bwd: null argument");
  XDFM_REQUIRE(B >= 0 && n > 0, "xdfm_cin_head_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (B > 0) {
    const int64_t tot = B * n;
    XDFM_LAUNCH("cin_head_dpooled_kernel", st, cin_head_dpooled_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, w,
                       g, B, n, d_pooled);
  }
  // dw, db: the weight-gradient path of a [n] -> [1] identity layer whose dy is g
  return xdfm_dense_bwd(pooled, w, g /* y unused for identity */, g, B, n, 1, XDFM_ACT_IDENTITY,
                        nullptr, 0, dw, db, part, stream);
}
