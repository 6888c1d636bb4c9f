// Dense optimizer step with folded regulariser gradients, logit/pred/data-loss, and the
// fused AUC + logloss evaluation, gfx950.
//
// Reference call sites replaced (recommenders/models/deeprec/):
//   models/base_model.py:121-167, :264-272   l2 / l1 / cross regularisation
//   models/base_model.py:283-302, :305-322   optimizer choice, clip_by_norm, apply_gradients
//   models/xDeepFM.py:46-71 + base_model.py:91-112, :215-262   logit sum, pred, data loss
//   deeprec_utils.py:535-545 (cal_metric auc / logloss) as used by run_eval base_model.py:588
//
// All reductions are two-level with a fixed association (per-block partial, then one
// block adds the partials in index order): no float atomics, run-to-run bit identical.
#include <hip/hip_runtime.h>

#include <cstring>

#include "drift.h"
#include "sort_plan.h"
#include "xdfm_common.h"
#include "stamp.h"

namespace {

constexpr int kMaxSeg = 32;
constexpr int kStatBlocks = 128;  // partial blocks per segment

struct SegTable {
  int n;
  long long off[kMaxSeg + 1];
  float l2[kMaxSeg], l1[kMaxSeg], nl2[kMaxSeg];
};

struct OptDev {
  int kind;
  float lr, alpha, one_minus_b1, one_minus_b2, eps, clip;
  const float* state;  // device {b1^t, b2^t, step size, t} or null
};

// block-wide sum, result valid in thread 0; fixed tree => deterministic
template <typename T, int THREADS>
__device__ __forceinline__ T block_sum(T v, T* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  T r = sh[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ float sgn(float p) { return p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f); }

// total gradient of one element: data gradient + regulariser gradients
__device__ __forceinline__ float total_grad(float g, float p, float l2, float l1, float nl2,
                                            float inv_norm) {
  float t = g;
  if (l2 != 0.f) t += l2 * p;                      // d/dp l2*sum(p^2)/2
  if (l1 != 0.f) t += l1 * sgn(p);                 // d/dp l1*sum|p|
  if (nl2 != 0.f) t += nl2 * (p * inv_norm);       // d/dp nl2*sqrt(sum p^2)
  return t;
}

// pass 0: partial sums of p^2 and |p| ; pass 1: partial sums of (total gradient)^2
__global__ __launch_bounds__(256) void dense_stats_kernel(SegTable tab, const float* __restrict__ param,
                                                          const float* __restrict__ grad,
                                                          const float* __restrict__ stats, int pass,
                                                          float* __restrict__ partial) {
  xdfm::Stamp stamp_(xdfm::kStampDenseStats);
  side_chain_priority();
  __shared__ float sh[256];
  const int s = blockIdx.y;
  const long long beg = tab.off[s], end = tab.off[s + 1];
  float a = 0.f, b = 0.f;
  float inv_norm = 0.f;
  if (pass == 1) {
    const float nrm = sqrtf(stats[s * 4 + 0]);
    inv_norm = nrm > 0.f ? 1.f / nrm : 0.f;
  }
  // eight elements of a thread's strided walk in flight (a large segment gives a thread ~80 of them; one
  // load at a time made this kernel 40 us at configs[4]); they are added in the walk's order as before
  constexpr long long kStride = (long long)kStatBlocks * 256;
  long long i = beg + (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * kStride < end; i += 8 * kStride) {
    float p[8], gq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = param[i + u * kStride];
    if (pass != 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) gq[u] = grad ? grad[i + u * kStride] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (pass == 0) {
        a += p[u] * p[u];
        b += fabsf(p[u]);
      } else {
        const float t = total_grad(gq[u], p[u], tab.l2[s], tab.l1[s], tab.nl2[s], inv_norm);
        a += t * t;
      }
    }
  }
  for (; i < end; i += kStride) {
    const float p = param[i];
    if (pass == 0) {
      a += p * p;
      b += fabsf(p);
    } else {
      const float t = total_grad(grad ? grad[i] : 0.f, p, tab.l2[s], tab.l1[s], tab.nl2[s], inv_norm);
      a += t * t;
    }
  }
  a = block_sum<float, 256>(a, sh);
  b = block_sum<float, 256>(b, sh);
  if (threadIdx.x == 0) {
    partial[((size_t)s * kStatBlocks + blockIdx.x) * 2 + 0] = a;
    partial[((size_t)s * kStatBlocks + blockIdx.x) * 2 + 1] = b;
  }
}

// one block of 16 waves: wave w finishes segments w and w + 16 (n_seg <= 32) with a fixed
// shuffle tree, no barrier per segment; pass 0 also produces the regularisation loss
__global__ __launch_bounds__(1024) void dense_stats_final_kernel(
    SegTable tab, const float* __restrict__ partial, int pass, float* __restrict__ stats,
    float* reg_loss, int accumulate) {
  xdfm::Stamp stamp_(xdfm::kStampDenseStats);
  side_chain_priority();
  static_assert(kStatBlocks == 128, "two partials per lane");
  __shared__ float regs[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = wave; s < tab.n; s += 16) {
    const float* ps = partial + (size_t)s * kStatBlocks * 2;
    float a = ps[lane * 2] + ps[(lane + 64) * 2];
    float b = ps[lane * 2 + 1] + ps[(lane + 64) * 2 + 1];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      a += __shfl_xor(a, m);
      b += __shfl_xor(b, m);
    }
    if (lane == 0) {
      if (pass == 0) {
        stats[s * 4 + 0] = a;  // sum p^2
        stats[s * 4 + 1] = b;  // sum |p|
        // base_model.py:126-131 (l2_loss = sum/2), :141-147 (norm ord=1), :161-165 (ord=1, ord=2)
        regs[s] = tab.l2[s] * (a * 0.5f) + tab.l1[s] * b + tab.nl2[s] * sqrtf(a);
      } else {
        stats[s * 4 + 2] = a;  // sum (total grad)^2
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && pass == 0 && reg_loss) {
    float reg = 0.f;
    for (int s = 0; s < tab.n; ++s) reg += regs[s];
    *reg_loss = accumulate ? *reg_loss + reg : reg;
  }
}

__global__ __launch_bounds__(256) void dense_update_kernel(SegTable tab, OptDev o, float* __restrict__ param,
                                                           const float* __restrict__ grad,
                                                           float* __restrict__ slot_m,
                                                           float* __restrict__ slot_v,
                                                           const float* __restrict__ stats) {
  xdfm::Stamp stamp_(xdfm::kStampDenseUpdate);
  const long long total = tab.off[tab.n];
  const float alpha = o.state ? o.state[2] : o.alpha;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    int s = 0;
    while (s + 1 < tab.n && i >= tab.off[s + 1]) ++s;
    float p = param[i];
    float inv_norm = 0.f;
    if (tab.nl2[s] != 0.f) {
      const float nrm = sqrtf(stats[s * 4 + 0]);
      inv_norm = nrm > 0.f ? 1.f / nrm : 0.f;
    }
    float g = total_grad(grad ? grad[i] : 0.f, p, tab.l2[s], tab.l1[s], tab.nl2[s], inv_norm);
    if (o.clip > 0.f) {
      // tf.clip_by_norm: g * clip / max(||g||, clip), per tensor (base_model.py:315-321)
      const float nrm = sqrtf(stats[s * 4 + 2]);
      g = g * o.clip / (nrm > o.clip ? nrm : o.clip);
    }
    if (o.kind == XDFM_OPT_ADAM) {
      float m = slot_m[i], v = slot_v[i];
      m = m + (g - m) * o.one_minus_b1;
      v = v + (g * g - v) * o.one_minus_b2;
      p = p - (m * alpha) / (sqrtf(v) + o.eps);
      slot_m[i] = m;
      slot_v[i] = v;
    } else if (o.kind == XDFM_OPT_ADAGRAD) {
      float a = slot_m[i];
      a = a + g * g;
      p = p - (g * o.lr) / sqrtf(a);
      slot_m[i] = a;
    } else if (o.kind == XDFM_OPT_RMSPROP) {
      float a = slot_m[i];
      a = a + (g * g - a) * o.one_minus_b2;
      p = p - (g * o.lr) / sqrtf(a + o.eps);
      slot_m[i] = a;
    } else if (o.kind == XDFM_OPT_ADADELTA) {
      float a = slot_m[i], u = slot_v[i];
      a = a * (1.f - o.one_minus_b2) + g * g * o.one_minus_b2;
      const float upd = sqrtf(u + o.eps) / sqrtf(a + o.eps) * g;
      u = u * (1.f - o.one_minus_b2) + upd * upd * o.one_minus_b2;
      p = p - upd * o.lr;
      slot_m[i] = a;
      slot_v[i] = u;
    } else if (o.kind == XDFM_OPT_FTRL) {
      const float a = slot_m[i];
      float l = slot_v[i];
      const float nw = a + g * g;
      l = l + g - (sqrtf(nw) - sqrtf(a)) / o.lr * p;
      p = -l / (sqrtf(nw) / o.lr);
      slot_m[i] = nw;
      slot_v[i] = l;
    } else {
      p = p - g * o.lr;
    }
    param[i] = p;
  }
}

// ------------------------------------------------------------------ drift table (drift.h)
// Partial sums (C0, C1, C2) over the steps i = k0 + 1, k0 + 1 + stride, ... <= n of a row whose stored
// values are current for step s:  c_i = alpha_{s+i} (b1/g)^i,  d_i = eps (g^-i - g^-kDriftC),  C_k = sum c_i d_i^k.
// alpha_j = lr sqrt(1 - b2^j) / (1 - b1^j) from fp32 powers (TF multiplies its beta powers up step by step:
// the same numbers to a few ulp); s >= 2^24 stands for "alpha = lr".
__device__ __forceinline__ float4 drift_partial(float lr, float b1, float b2, float eps, int s, int n, int k0,
                                                int stride) {
  const float lb1 = log2f(b1), lb2 = log2f(b2);
  const float lq = lb1 - 0.5f * lb2;                              // log2(b1 / g)
  const float gc = exp2f(-(float)xdfm::kDriftC * 0.5f * lb2);     // g^-kDriftC
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int i = k0 + 1; i <= n; i += stride) {
    const float j = (float)s + (float)i;
    const float alpha = lr * sqrtf(1.f - exp2f(j * lb2)) / (1.f - exp2f(j * lb1));
    const float c = alpha * exp2f((float)i * lq);
    const float d = eps * (exp2f(-(float)i * 0.5f * lb2) - gc);
    c0 += c;
    c1 += c * d;
    c2 += c * d * d;
  }
  return make_float4(c0, c1, c2, 0.f);
}

// the four lanes of a quad hold the partial sums of one entry: added in a fixed tree
__device__ __forceinline__ float4 quad_sum(float4 c) {
  c.x += __shfl_xor(c.x, 1);
  c.y += __shfl_xor(c.y, 1);
  c.z += __shfl_xor(c.z, 1);
  c.x += __shfl_xor(c.x, 2);
  c.y += __shfl_xor(c.y, 2);
  c.z += __shfl_xor(c.z, 2);
  return c;
}

// entries of slot t & 1 for G = 0 .. kDriftG-1 steps missed as of step t (s = t - G; no row has s < 0)
__device__ __forceinline__ void drift_fill_cur(float4* tab, int t, float lr, float b1, float b2, float eps) {
  const int G = threadIdx.x >> 2, k = threadIdx.x & 3;
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t - G >= 0) c = drift_partial(lr, b1, b2, eps, t - G, G, k, 4);
  c = quad_sum(c);
  if (k == 0) tab[(t & 1) * xdfm::kDriftG + G] = c;
}

// TF keeps beta1_power / beta2_power as fp32 variables multiplied up every step; same here.
// kDriftG threads: the state moves on in thread 0; thread G forms the drift entry of "G steps missed as
// of the new step t" from the entry of "G - 1 steps missed as of t - 1" (the same row one step earlier, in
// the other slot) by adding step t's term:  C_k(t - G, G) = C_k(t - G, G - 1) + alpha_t (b1/g)^G d_G^k.
__global__ __launch_bounds__(xdfm::kDriftG) void opt_advance_kernel(float* st, float lr, float beta1, float beta2,
                                                                    float eps, float4* tab) {
  xdfm::Stamp stamp_(xdfm::kStampOptAdvance);
  const float b1p = st[0] * beta1, b2p = st[1] * beta2, t = st[3] + 1.f;
  const float alpha = lr * sqrtf(1.f - b2p) / (1.f - b1p);
  __syncthreads();  // every thread has read the old state
  if (threadIdx.x == 0) {
    st[0] = b1p;
    st[1] = b2p;
    st[2] = alpha;
    st[3] = t;
  }
  if (tab) {
    const int ti = (int)t, G = threadIdx.x;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (G >= 1 && G <= ti) {
      const float lb1 = log2f(beta1), lb2 = log2f(beta2);
      c = tab[((ti - 1) & 1) * xdfm::kDriftG + G - 1];
      const float w = alpha * exp2f((float)G * (lb1 - 0.5f * lb2));
      const float d = eps * (exp2f(-(float)G * 0.5f * lb2) - exp2f(-(float)xdfm::kDriftC * 0.5f * lb2));
      c.x += w;
      c.y += w * d;
      c.z += w * d * d;
    }
    tab[(ti & 1) * xdfm::kDriftG + G] = c;
  }
}

__global__ __launch_bounds__(4 * xdfm::kDriftG) void drift_init_kernel(const float* st, float lr, float beta1,
                                                                       float beta2, float eps, float4* tab) {
  if (blockIdx.x == 0) {
    drift_fill_cur(tab, (int)st[3], lr, beta1, beta2, eps);
    return;
  }
  // hist[s]: the first kDriftG - 1 missed steps after step s (the rest weighs (b1/g)^255)
  const int s = (int)(blockIdx.x - 1) * xdfm::kDriftG + (threadIdx.x >> 2);
  const int k = threadIdx.x & 3;
  if (s > xdfm::kDriftS) return;  // whole quads leave together
  float4 c = drift_partial(lr, beta1, beta2, eps, s < xdfm::kDriftS ? s : (1 << 24), xdfm::kDriftG - 1, k, 4);
  c = quad_sum(c);
  if (k == 0) tab[2 * xdfm::kDriftG + s] = c;
}

// ------------------------------------------------------------------ loss
struct LossArgs {
  const float *part[4], *bias, *labels;
  long long B;
  long long mean_count;  // samples the mean runs over (B, or the global batch under data parallelism)
  int classification, loss_kind;
  float *logit, *pred, *g_logit;
};

__device__ __forceinline__ float sigmoidf(float z) {
  // same split as Eigen's logistic: no overflow for large |z|
  if (z >= 0.f) return 1.f / (1.f + expf(-z));
  const float e = expf(z);
  return e / (1.f + e);
}

__device__ __forceinline__ float sample_logit(const LossArgs& a, long long i) {
  float z = 0.f;
  bool first = true;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (a.part[k]) {
      float v = a.part[k][i];
      if (k == 0 && a.bias) v += a.bias[0];  // linear part: matmul + b (xDeepFM.py:128)
      z = first ? v : z + v;
      first = false;
    }
  return z;
}

constexpr float kLogEps = 1e-7f;  // tf.compat.v1.losses.log_loss default epsilon

// pass A: logit, pred, per-sample loss term -> per-block partial sums
__global__ __launch_bounds__(256) void loss_fwd_kernel(LossArgs a, float* __restrict__ partial) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.B;
       i += (long long)gridDim.x * 256) {
    const float z = sample_logit(a, i);
    const float p = a.classification ? sigmoidf(z) : z;
    a.logit[i] = z;
    a.pred[i] = p;
    if (a.labels) {
      const float y = a.labels[i];
      float l;
      if (a.loss_kind == XDFM_LOSS_LOG) {
        l = -y * logf(p + kLogEps) - (1.f - y) * logf(1.f - p + kLogEps);
      } else if (a.loss_kind == XDFM_LOSS_CROSS_ENTROPY) {
        // max(z,0) - z*y + log(1+exp(-|z|))
        l = fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
      } else {
        const float d = p - y;
        l = d * d;
      }
      acc += l;
    }
  }
  acc = block_sum<float, 256>(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// pass B: finish the mean (and sqrt for square_loss), then d loss / d logit
__global__ __launch_bounds__(256) void loss_bwd_kernel(LossArgs a, const float* __restrict__ partial,
                                                       int n_partial, float* loss_out) {
  __shared__ float sh[256];
  float s = 0.f;
  for (int k = threadIdx.x; k < n_partial; k += 256) s += partial[k];
  s = block_sum<float, 256>(s, sh);
  __shared__ float loss_sh;
  if (threadIdx.x == 0) {
    float mean = s / (float)a.mean_count;
    loss_sh = (a.loss_kind == XDFM_LOSS_SQUARE) ? sqrtf(mean) : mean;
    if (loss_out && blockIdx.x == 0) *loss_out = loss_sh;
  }
  __syncthreads();
  if (!a.g_logit) return;
  const float loss = loss_sh;
  const float invB = 1.f / (float)a.mean_count;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.B;
       i += (long long)gridDim.x * 256) {
    const float z = a.logit[i], p = a.pred[i], y = a.labels[i];
    const float dp_dz = a.classification ? p * (1.f - p) : 1.f;
    float g;
    if (a.loss_kind == XDFM_LOSS_LOG) {
      g = (-y / (p + kLogEps) + (1.f - y) / (1.f - p + kLogEps)) * dp_dz * invB;
    } else if (a.loss_kind == XDFM_LOSS_CROSS_ENTROPY) {
      g = (sigmoidf(z) - y) * invB;
    } else {
      // d sqrt(mean d^2) = d / (B * rmse)
      g = loss > 0.f ? (p - y) * dp_dz * invB / loss : 0.f;
    }
    a.g_logit[i] = g;
  }
}

// ------------------------------------------------- CIN head + loss + gradients, one launch
// The training step's chain cin_head_fwd -> loss_fwd -> loss_bwd -> cin_head_dpooled in one
// kernel (each link is a few microseconds of work behind a launch on the step's critical path).
// Same expressions in the same order as those kernels: results are bit-identical.  The mean
// is finished by whichever block arrives last, summing the per-block partials in index order.
struct HeadArgs {
  const float *pooled, *w, *b;
  int n;
  float *cin_out, *d_pooled;
};

constexpr int kHeadThreads = 64;  // one wave per block: many short blocks, cheap barriers
__global__ __launch_bounds__(kHeadThreads) void head_loss_kernel(LossArgs a, HeadArgs h, float* __restrict__ partial,
                                                        unsigned* __restrict__ counter,
                                                        float* __restrict__ loss_out) {
  __shared__ float sh[kHeadThreads];
  __shared__ float sh_g[kHeadThreads];
  __shared__ float sh_tile[kHeadThreads][33];
  __shared__ unsigned ticket;
  const float invB = 1.f / (float)a.mean_count;
  float acc = 0.f;
  for (long long i0 = (long long)blockIdx.x * kHeadThreads; i0 < a.B;
       i0 += (long long)gridDim.x * kHeadThreads) {
    const long long i = i0 + threadIdx.x;
    float g = 0.f;
    float cin = 0.f;
    if (h.pooled) {
      // A lane sums its own sample's row in k order (the order of cin_head_fwd: bit-identical), but
      // the rows come through LDS 32 columns at a time: read straight from memory, the 64 lanes of a
      // load touch 64 different lines (126 us for 512 channels at batch 8192; 11 this way).
      const long long left = a.B - i0;
      const int rows = (int)(left < kHeadThreads ? left : kHeadThreads);
      float base = 0.f, dot = 0.f;
      float v[32];
      auto fetch = [&](int k0) {  // lanes 0-31: 128 contiguous bytes of row 2j, lanes 32-63: of row 2j + 1
        const int kw = h.n - k0 < 32 ? h.n - k0 : 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int rr = 2 * j + (threadIdx.x >> 5), kk = threadIdx.x & 31;
          v[j] = (rr < rows && kk < kw) ? h.pooled[(i0 + rr) * h.n + k0 + kk] : 0.f;
        }
      };
      fetch(0);
      for (int k0 = 0; k0 < h.n; k0 += 32) {
        const int kw = h.n - k0 < 32 ? h.n - k0 : 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) sh_tile[2 * j + (threadIdx.x >> 5)][threadIdx.x & 31] = v[j];
        __syncthreads();
        if (k0 + 32 < h.n) fetch(k0 + 32);  // the next columns are on their way during the sums
        if (i < a.B) {
          float x[32];
#pragma unroll
          for (int kk = 0; kk < 32; ++kk) x[kk] = sh_tile[threadIdx.x][kk];
          if (kw == 32) {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
              base += x[kk];
              dot = fmaf(x[kk], h.w[k0 + kk], dot);
            }
          } else {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk)
              if (kk < kw) {
                base += x[kk];
                dot = fmaf(x[kk], h.w[k0 + kk], dot);
              }
          }
        }
        __syncthreads();
      }
      cin = base + (dot + h.b[0]);
      if (i < a.B) h.cin_out[i] = cin;
    }
    if (i < a.B) {
      float z = 0.f;
      bool first = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool is_cin = (k == 2 && h.pooled);
        if (a.part[k] || is_cin) {
          float v = is_cin ? cin : a.part[k][i];
          if (k == 0 && a.bias) v += a.bias[0];
          z = first ? v : z + v;
          first = false;
        }
      }
      const float p = a.classification ? sigmoidf(z) : z;
      a.logit[i] = z;
      a.pred[i] = p;
      const float y = a.labels[i];
      const float dp_dz = a.classification ? p * (1.f - p) : 1.f;
      if (a.loss_kind == XDFM_LOSS_LOG) {
        acc += -y * logf(p + kLogEps) - (1.f - y) * logf(1.f - p + kLogEps);
        g = (-y / (p + kLogEps) + (1.f - y) / (1.f - p + kLogEps)) * dp_dz * invB;
      } else {  // XDFM_LOSS_CROSS_ENTROPY (square_loss needs the finished mean first: not served here)
        acc += fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
        g = (sigmoidf(z) - y) * invB;
      }
      a.g_logit[i] = g;
    }
    if (h.d_pooled) {
      sh_g[threadIdx.x] = g;
      __syncthreads();
      const long long left = a.B - i0;
      const int rows = (int)(left < kHeadThreads ? left : kHeadThreads);
      // a lane keeps its column's weight and walks down the rows: no division per element, the 64 lanes
      // of a store write 256 contiguous bytes (the same expression per element as cin_head_bwd)
      float* __restrict__ dp = h.d_pooled + i0 * h.n;
      for (int k = threadIdx.x; k < h.n; k += kHeadThreads) {
        const float wk = h.w[k];
        int b = 0;
        for (; b + 8 <= rows; b += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float gb = sh_g[b + u];
            dp[(long long)(b + u) * h.n + k] = gb + gb * wk;
          }
        }
        for (; b < rows; ++b) {
          const float gb = sh_g[b];
          dp[(long long)b * h.n + k] = gb + gb * wk;
        }
      }
      __syncthreads();
    }
  }
  acc = block_sum<float, kHeadThreads>(acc, sh);
  if (!counter) {  // the caller finishes the mean elsewhere (xdfm_loss_finish)
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
    return;
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = acc;
    __threadfence();
    ticket = atomicAdd(counter, 1u);
  }
  __syncthreads();
  if (ticket != gridDim.x - 1) return;
  __threadfence();
  const volatile float* vp = partial;
  float s = 0.f;
  for (int k = threadIdx.x; k < (int)gridDim.x; k += kHeadThreads) s += vp[k];
  s = block_sum<float, kHeadThreads>(s, sh);
  if (threadIdx.x == 0) {
    if (loss_out) *loss_out = s / (float)a.mean_count;
    *counter = 0u;  // ready for the next launch
  }
}

// the mean of a deferred xdfm_head_loss: partials in index order, as its last block would add them
__global__ __launch_bounds__(kHeadThreads) void loss_finish_kernel(const float* __restrict__ partial, int n,
                                                                   float count, int root,
                                                                   float* __restrict__ loss_out) {
  xdfm::Stamp stamp_(xdfm::kStampLossFinish);
  side_chain_priority();
  __shared__ float sh[kHeadThreads];
  float s = 0.f;
  for (int k = threadIdx.x; k < n; k += kHeadThreads) s += partial[k];
  s = block_sum<float, kHeadThreads>(s, sh);
  if (threadIdx.x == 0) {
    const float mean = s / count;
    *loss_out = root ? sqrtf(mean) : mean;
  }
}

// ------------------------------------------------------------------ eval
__device__ __forceinline__ uint32_t float_key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order-preserving
}

__device__ __forceinline__ float key_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void eval_keys_kernel(const float* __restrict__ preds, const float* __restrict__ labels,
                                 long long n, uint32_t* __restrict__ keys, uint32_t* __restrict__ pos) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = float_key(preds[i] + 0.f);  // (-0.0 + 0.0 = +0.0: the two zeros are one score, a tie)
  pos[i] = labels[i] > 0.5f ? 1u : 0u;
}

// One pass over the score-sorted (key, label) pairs (ascending), per block in a fixed order:
//  * AUC as sklearn's roc_auc_score forms it (trapezoids over tied scores = Mann-Whitney U with average
//    ranks; deeprec_utils.py:535-537), in integers: a positive in the tie group [s, e) has twice the
//    average rank s + e + 1 (1-based), so 2 U = sum over positives of (s + e + 1) - n_pos (n_pos + 1).
//    A sample whose neighbours hold other scores is its own group; only ties search for their bounds.
//  * logloss after the reference's clip to [1e-11, 1 - 1e-11] (deeprec_utils.py:541-545), in float64
//    (the score is recovered from its key bit for bit, the label from the payload's truth bit; the
//    float label itself decides acc-style thresholds nowhere here: labels are 0 / 1).
__global__ __launch_bounds__(256) void eval_reduce_kernel(const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ pos, long long n, unsigned long long* __restrict__ part_rank,
                                                          unsigned long long* __restrict__ part_pos,
                                                          double* __restrict__ part_ll) {
  __shared__ unsigned long long shu[256];
  __shared__ double shd[256];
  unsigned long long rank2 = 0, npos = 0;
  double ll = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const uint32_t k = keys[i];
    const bool y = pos[i] != 0u;
    double p = (double)key_float(k);
    p = fmin(fmax(p, 10e-12), 1.0 - 10e-12);  // deeprec_utils.py:543
    ll += y ? -log(p) : -log(1.0 - p);
    if (y) {
      long long s = i, e = i + 1;
      if (i > 0 && keys[i - 1] == k) {  // first position holding k
        long long lo = 0, hi = i - 1;   // keys[hi] == k
        while (lo < hi) {
          const long long mid = (lo + hi) >> 1;
          if (keys[mid] == k) hi = mid; else lo = mid + 1;
        }
        s = lo;
      }
      if (i + 1 < n && keys[i + 1] == k) {  // first position past the run of k
        long long lo = i + 2, hi = n;
        while (lo < hi) {
          const long long mid = (lo + hi) >> 1;
          if (keys[mid] == k) lo = mid + 1; else hi = mid;
        }
        e = lo;
      }
      rank2 += (unsigned long long)(s + e + 1);
      npos += 1;
    }
  }
  rank2 = block_sum<unsigned long long, 256>(rank2, shu);
  npos = block_sum<unsigned long long, 256>(npos, shu);
  ll = block_sum<double, 256>(ll, shd);
  if (threadIdx.x == 0) {
    part_rank[blockIdx.x] = rank2;
    part_pos[blockIdx.x] = npos;
    part_ll[blockIdx.x] = ll;
  }
}

__global__ __launch_bounds__(256) void eval_final_kernel(const unsigned long long* __restrict__ part_rank,
                                                         const unsigned long long* __restrict__ part_pos,
                                                         const double* __restrict__ part_ll, int n_partial,
                                                         long long n, double* out) {
  __shared__ unsigned long long shu[256];
  __shared__ double shd[256];
  unsigned long long r = 0, c = 0;
  double s = 0.0;
  for (int k = threadIdx.x; k < n_partial; k += 256) {
    r += part_rank[k];
    c += part_pos[k];
    s += part_ll[k];
  }
  r = block_sum<unsigned long long, 256>(r, shu);
  c = block_sum<unsigned long long, 256>(c, shu);
  s = block_sum<double, 256>(s, shd);
  if (threadIdx.x == 0) {
    const double n_pos = (double)c, n_neg = (double)((unsigned long long)n - c);
    const unsigned long long u2 = r - c * (c + 1ull);  // 2 U, exact
    out[0] = (c > 0 && (unsigned long long)n > c) ? (double)u2 / (2.0 * n_pos * n_neg) : nan("");
    out[1] = n > 0 ? s / (double)n : nan("");
    out[2] = n_pos;
    out[3] = (double)n;
  }
}

// [data loss, regularisation loss] -> [data, data + reg]: the two numbers a training step reports
__global__ void loss_pair_kernel(const float* __restrict__ loss2, float* __restrict__ out2) {
  xdfm::Stamp stamp_(xdfm::kStampLossPair);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float d = loss2[0];
    out2[0] = d;
    out2[1] = d + loss2[1];
  }
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int kEvalBlocks = 1024;

struct EvalLayout {
  size_t keys_in, pos_in, keys_out, pos_out, part_rank, part_pos, part_ll, sort_ws, sort_ws_bytes, total;
};

int eval_layout(long long n, EvalLayout* L) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align_up(bytes, 256);
    return o;
  };
  const size_t nn = (size_t)n;
  L->keys_in = take(nn * 4);
  L->pos_in = take(nn * 4);
  L->keys_out = take(nn * 4);
  L->pos_out = take(nn * 4);
  L->part_rank = take(kEvalBlocks * sizeof(unsigned long long));
  L->part_pos = take(kEvalBlocks * sizeof(unsigned long long));
  L->part_ll = take(kEvalBlocks * sizeof(double));
  L->sort_ws_bytes = xdfm::plan_sort_workspace_bytes(n);
  L->sort_ws = take(L->sort_ws_bytes);
  L->total = off + 256;
  return XDFM_OK;
}

}  // namespace

extern "C" int xdfm_dense_update(float* param, const float* grad, float* slot_m, float* slot_v,
                                 int32_t n_seg, const int64_t* seg_off, const float* seg_l2,
                                 const float* seg_l1, const float* seg_norm_l2,
                                 const xdfm_opt_args* opt, float* stats, float* reg_loss,
                                 int32_t accumulate_reg_loss, void* stream) {
  XDFM_REQUIRE(param && seg_off && opt && stats, "xdfm_dense_update: null argument");
  XDFM_REQUIRE(n_seg > 0 && n_seg <= kMaxSeg, "xdfm_dense_update: n_seg %d outside [1,%d]", n_seg, kMaxSeg);
  XDFM_REQUIRE(opt->kind >= XDFM_OPT_SGD && opt->kind <= XDFM_OPT_FTRL,
               "xdfm_dense_update: unknown optimizer kind %d", opt->kind);
  XDFM_REQUIRE(opt->kind == XDFM_OPT_SGD || slot_m, "xdfm_dense_update: slot_m missing");
  XDFM_REQUIRE(!xdfm::opt_needs_v(opt->kind) || slot_v, "xdfm_dense_update: slot_v missing");
  SegTable tab;
  memset(&tab, 0, sizeof(tab));
  tab.n = n_seg;
  const bool precomputed = accumulate_reg_loss == XDFM_REG_PRECOMPUTED;
  XDFM_REQUIRE(!precomputed || opt->clip_norm == 0.f,
               "xdfm_dense_update: clipping needs the gradient statistics, which cannot be precomputed");
  bool need_stats = (reg_loss != nullptr) || opt->clip_norm > 0.f;
  for (int s = 0; s <= n_seg; ++s) {
    tab.off[s] = seg_off[s];
    XDFM_REQUIRE(s == 0 || seg_off[s] >= seg_off[s - 1], "xdfm_dense_update: seg_off not ascending");
  }
  XDFM_REQUIRE(seg_off[0] == 0, "xdfm_dense_update: seg_off[0] must be 0");
  for (int s = 0; s < n_seg; ++s) {
    tab.l2[s] = seg_l2 ? seg_l2[s] : 0.f;
    tab.l1[s] = seg_l1 ? seg_l1[s] : 0.f;
    tab.nl2[s] = seg_norm_l2 ? seg_norm_l2[s] : 0.f;
    if (tab.nl2[s] != 0.f) need_stats = true;
  }
  OptDev o;
  o.kind = opt->kind;
  o.lr = opt->lr;
  o.eps = opt->eps;
  o.one_minus_b1 = 1.f - opt->beta1;
  o.one_minus_b2 = 1.f - opt->beta2;
  o.clip = opt->clip_norm;
  o.state = opt->opt_state;
  o.alpha = (opt->kind == XDFM_OPT_ADAM)
                ? opt->lr * sqrtf(1.f - opt->beta2_power) / (1.f - opt->beta1_power)
                : 0.f;
  hipStream_t st = (hipStream_t)stream;
  const long long total = tab.off[n_seg];
  if (total == 0) {
    if (reg_loss && !accumulate_reg_loss) XDFM_HIP(hipMemsetAsync(reg_loss, 0, sizeof(float), st));
    return XDFM_OK;
  }
  float* partial = stats + (size_t)n_seg * 4;
  if (need_stats && !precomputed) {
    XDFM_LAUNCH("dense_stats_kernel", st, dense_stats_kernel, dim3(kStatBlocks, n_seg), dim3(256), 0, tab, param, grad,
                       stats, 0, partial);
    XDFM_LAUNCH("dense_stats_final_kernel", st, dense_stats_final_kernel, dim3(1), dim3(1024), 0, tab, partial, 0,
                       stats, reg_loss, accumulate_reg_loss);
    if (opt->clip_norm > 0.f) {
      XDFM_LAUNCH("dense_stats_kernel", st, dense_stats_kernel, dim3(kStatBlocks, n_seg), dim3(256), 0, tab, param,
                         grad, stats, 1, partial);
      XDFM_LAUNCH("dense_stats_final_kernel", st, dense_stats_final_kernel, dim3(1), dim3(1024), 0, tab, partial, 1,
                         stats, nullptr, 0);
    }
  }
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  XDFM_LAUNCH("dense_update_kernel", st, dense_update_kernel, dim3((unsigned)blocks), dim3(256), 0, tab, o, param, grad,
                     slot_m, slot_v, stats);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_dense_reg_loss(const float* param, int32_t n_seg, const int64_t* seg_off, const float* seg_l2,
                                   const float* seg_l1, const float* seg_norm_l2, float* stats, float* reg_loss,
                                   int32_t accumulate_reg_loss, void* stream) {
  XDFM_REQUIRE(param && seg_off && stats && reg_loss, "xdfm_dense_reg_loss: null argument");
  XDFM_REQUIRE(n_seg > 0 && n_seg <= kMaxSeg, "xdfm_dense_reg_loss: n_seg %d outside [1,%d]", n_seg, kMaxSeg);
  XDFM_REQUIRE(seg_off[0] == 0, "xdfm_dense_reg_loss: seg_off[0] must be 0");
  SegTable tab;
  memset(&tab, 0, sizeof(tab));
  tab.n = n_seg;
  for (int s = 0; s <= n_seg; ++s) {
    tab.off[s] = seg_off[s];
    XDFM_REQUIRE(s == 0 || seg_off[s] >= seg_off[s - 1], "xdfm_dense_reg_loss: seg_off not ascending");
  }
  for (int s = 0; s < n_seg; ++s) {
    tab.l2[s] = seg_l2 ? seg_l2[s] : 0.f;
    tab.l1[s] = seg_l1 ? seg_l1[s] : 0.f;
    tab.nl2[s] = seg_norm_l2 ? seg_norm_l2[s] : 0.f;
  }
  hipStream_t st = (hipStream_t)stream;
  if (tab.off[n_seg] == 0) {
    if (!accumulate_reg_loss) XDFM_HIP(hipMemsetAsync(reg_loss, 0, sizeof(float), st));
    return XDFM_OK;
  }
  float* partial = stats + (size_t)n_seg * 4;
  XDFM_LAUNCH("dense_stats_kernel", st, dense_stats_kernel, dim3(kStatBlocks, n_seg), dim3(256), 0, tab, param, nullptr,
              stats, 0, partial);
  XDFM_LAUNCH("dense_stats_final_kernel", st, dense_stats_final_kernel, dim3(1), dim3(1024), 0, tab, partial, 0, stats,
              reg_loss, accumulate_reg_loss);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_opt_advance(float* opt_state, float lr, float beta1, float beta2, float eps, float* drift_tab,
                                void* stream) {
  XDFM_REQUIRE(opt_state, "xdfm_opt_advance: null state");
  XDFM_REQUIRE(!drift_tab || (beta1 > 0.f && beta1 < 1.f && beta2 > 0.f && beta2 < 1.f),
               "xdfm_opt_advance: the drift table is Adam's (betas in (0, 1))");
  XDFM_LAUNCH("opt_advance_kernel", stream, opt_advance_kernel, dim3(1), dim3(xdfm::kDriftG), 0, opt_state, lr,
              beta1, beta2, eps, reinterpret_cast<float4*>(drift_tab));
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_drift_tab_floats(size_t* floats) {
  XDFM_REQUIRE(floats, "xdfm_drift_tab_floats: null argument");
  *floats = (size_t)xdfm::kDriftFloat4 * 4;
  return XDFM_OK;
}

extern "C" int xdfm_drift_tab_init(float* drift_tab, const float* opt_state, float lr, float beta1, float beta2,
                                   float eps, void* stream) {
  XDFM_REQUIRE(drift_tab && opt_state, "xdfm_drift_tab_init: null argument");
  XDFM_REQUIRE((reinterpret_cast<uintptr_t>(drift_tab) % 16) == 0, "xdfm_drift_tab_init: the table must be 16-byte aligned");
  XDFM_REQUIRE(beta1 > 0.f && beta1 < 1.f && beta2 > 0.f && beta2 < 1.f, "xdfm_drift_tab_init: betas in (0, 1)");
  hipStream_t st = (hipStream_t)stream;
  XDFM_HIP(hipMemsetAsync(drift_tab, 0, (size_t)2 * xdfm::kDriftG * sizeof(float4), st));
  const unsigned blocks = 1 + (xdfm::kDriftS + 1 + xdfm::kDriftG - 1) / xdfm::kDriftG;
  XDFM_LAUNCH("drift_init_kernel", st, drift_init_kernel, dim3(blocks), dim3(4 * xdfm::kDriftG), 0, opt_state, lr,
              beta1, beta2, eps, reinterpret_cast<float4*>(drift_tab));
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_logit_loss(const float* part0, const float* part1, const float* part2,
                               const float* part3, const float* bias, const float* labels, int64_t B,
                               int64_t mean_count, int32_t classification, int32_t loss_kind,
                               float* logit, float* pred, float* g_logit, float* loss, float* scratch,
                               void* stream) {
  XDFM_REQUIRE(logit && pred && scratch, "xdfm_logit_loss: null output");
  XDFM_REQUIRE(part0 || part1 || part2 || part3, "xdfm_logit_loss: no logit part given");
  XDFM_REQUIRE(B > 0, "xdfm_logit_loss: empty batch");
  XDFM_REQUIRE(mean_count == 0 || mean_count >= B, "xdfm_logit_loss: mean_count must be 0 or >= B");
  XDFM_REQUIRE(mean_count == 0 || mean_count == B || loss_kind != XDFM_LOSS_SQUARE,
               "xdfm_logit_loss: square_loss (a root of a mean) cannot be split over ranks");
  XDFM_REQUIRE(loss_kind >= XDFM_LOSS_LOG && loss_kind <= XDFM_LOSS_SQUARE,
               "xdfm_logit_loss: unknown loss %d", loss_kind);
  XDFM_REQUIRE(labels || (!g_logit && !loss), "xdfm_logit_loss: loss/gradient need labels");
  XDFM_REQUIRE(!bias || part0, "xdfm_logit_loss: bias belongs to part0 (the linear part)");
  LossArgs a;
  a.part[0] = part0;
  a.part[1] = part1;
  a.part[2] = part2;
  a.part[3] = part3;
  a.bias = bias;
  a.labels = labels;
  a.B = B;
  a.mean_count = mean_count > 0 ? mean_count : B;
  a.classification = classification;
  a.loss_kind = loss_kind;
  a.logit = logit;
  a.pred = pred;
  a.g_logit = g_logit;
  hipStream_t st = (hipStream_t)stream;
  long long blocks = (B + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  XDFM_LAUNCH("loss_fwd_kernel", st, loss_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, a, scratch);
  if (labels && (g_logit || loss))
    XDFM_LAUNCH("loss_bwd_kernel", st, loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, a, scratch,
                       (int)blocks, loss);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_head_loss(const float* part0, const float* part1, const float* part3, const float* bias,
                              const float* pooled, const float* head_w, const float* head_b, int32_t n,
                              const float* labels, int64_t B, int64_t mean_count, int32_t classification,
                              int32_t loss_kind, float* logit, float* pred, float* cin_out, float* g_logit,
                              float* d_pooled, float* loss, float* scratch, void* stream) {
  XDFM_REQUIRE(pooled && head_w && head_b && cin_out && d_pooled && n > 0, "xdfm_head_loss: null head argument");
  XDFM_REQUIRE(labels && logit && pred && g_logit && scratch, "xdfm_head_loss: null argument");
  XDFM_REQUIRE(B > 0, "xdfm_head_loss: empty batch");
  XDFM_REQUIRE(mean_count == 0 || mean_count >= B, "xdfm_head_loss: mean_count must be 0 or >= B");
  XDFM_REQUIRE(loss_kind >= XDFM_LOSS_LOG && loss_kind <= XDFM_LOSS_SQUARE, "xdfm_head_loss: unknown loss %d",
               loss_kind);
  XDFM_REQUIRE(!bias || part0, "xdfm_head_loss: bias belongs to part0 (the linear part)");
  if (loss_kind == XDFM_LOSS_SQUARE) {
    // the gradient of a root of a mean needs the finished mean: the separate kernels serve it
    int rc = xdfm_cin_head_fwd(pooled, head_w, head_b, B, n, cin_out, stream);
    if (rc) return rc;
    rc = xdfm_logit_loss(part0, part1, cin_out, part3, bias, labels, B, mean_count, classification, loss_kind,
                         logit, pred, g_logit, loss, scratch, stream);
    if (rc) return rc;
    return xdfm_cin_head_bwd(pooled, head_w, g_logit, B, n, d_pooled, nullptr, nullptr, nullptr, stream);
  }
  LossArgs a;
  a.part[0] = part0;
  a.part[1] = part1;
  a.part[2] = nullptr;
  a.part[3] = part3;
  a.bias = bias;
  a.labels = labels;
  a.B = B;
  a.mean_count = mean_count > 0 ? mean_count : B;
  a.classification = classification;
  a.loss_kind = loss_kind;
  a.logit = logit;
  a.pred = pred;
  a.g_logit = g_logit;
  const HeadArgs h{pooled, head_w, head_b, n, cin_out, d_pooled};
  long long blocks = (B + kHeadThreads - 1) / kHeadThreads;
  if (blocks > 1024) blocks = 1024;
  XDFM_LAUNCH("head_loss_kernel", (hipStream_t)stream, head_loss_kernel, dim3((unsigned)blocks), dim3(kHeadThreads), 0,
              a, h,
              scratch, loss ? reinterpret_cast<unsigned*>(scratch + 1024) : nullptr, loss);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_loss_finish(const float* scratch, int64_t B, int64_t mean_count, int32_t loss_kind,
                                float* loss, void* stream) {
  XDFM_REQUIRE(scratch && loss && B > 0, "xdfm_loss_finish: bad argument");
  XDFM_REQUIRE(loss_kind >= XDFM_LOSS_LOG && loss_kind <= XDFM_LOSS_SQUARE, "xdfm_loss_finish: unknown loss %d",
               loss_kind);
  // as many partial sums as the call being finished wrote: one per block of its first pass
  const int per = loss_kind == XDFM_LOSS_SQUARE ? 256 : kHeadThreads;
  long long blocks = (B + per - 1) / per;
  if (blocks > 1024) blocks = 1024;
  XDFM_LAUNCH("loss_finish_kernel", (hipStream_t)stream, loss_finish_kernel, dim3(1), dim3(kHeadThreads), 0, scratch,
              (int)blocks, (float)(mean_count > 0 ? mean_count : B), loss_kind == XDFM_LOSS_SQUARE ? 1 : 0, loss);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_loss_finish_partials(const float* scratch, int32_t n_partial, int64_t count, float* loss,
                                         void* stream) {
  XDFM_REQUIRE(scratch && loss && n_partial > 0 && count > 0, "xdfm_loss_finish_partials: bad argument");
  XDFM_LAUNCH("loss_finish_kernel", (hipStream_t)stream, loss_finish_kernel, dim3(1), dim3(kHeadThreads), 0, scratch,
              (int)n_partial, (float)count, 0, loss);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_eval_workspace_bytes(int64_t n, size_t* bytes) {
  XDFM_REQUIRE(bytes && n > 0 && n < ((int64_t)1 << 31), "xdfm_eval_workspace_bytes: bad argument");
  EvalLayout L;
  int rc = eval_layout(n, &L);
  if (rc) return rc;
  *bytes = L.total;
  return XDFM_OK;
}

extern "C" int xdfm_eval_auc_logloss(const float* preds, const float* labels, int64_t n, double* out,
                                     void* ws, size_t ws_bytes, void* stream) {
  XDFM_REQUIRE(preds && labels && out && ws, "xdfm_eval_auc_logloss: null argument");
  XDFM_REQUIRE(n > 0 && n < ((int64_t)1 << 31), "xdfm_eval_auc_logloss: n out of range");
  EvalLayout L;
  int rc = eval_layout(n, &L);
  if (rc) return rc;
  if (ws_bytes < L.total)
    return xdfm::fail(XDFM_E_WORKSPACE, "xdfm_eval_auc_logloss: workspace %zu < %zu", ws_bytes, L.total);
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);  // (L.total has the slack)
  uint32_t* keys_in = (uint32_t*)(base + L.keys_in);
  uint32_t* pos_in = (uint32_t*)(base + L.pos_in);
  uint32_t* keys_out = (uint32_t*)(base + L.keys_out);
  uint32_t* pos_out = (uint32_t*)(base + L.pos_out);
  unsigned long long* part_rank = (unsigned long long*)(base + L.part_rank);
  unsigned long long* part_pos = (unsigned long long*)(base + L.part_pos);
  double* part_ll = (double*)(base + L.part_ll);
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  XDFM_LAUNCH("eval_keys_kernel", st, eval_keys_kernel, dim3(blocks), dim3(256), 0, preds, labels, (long long)n,
              keys_in, pos_in);
  // the in-tree stable LSD radix sort (csrc/radix_sort.hip) over the 32 key bits, label as payload
  rc = xdfm::sort_pairs_u32(keys_in, pos_in, n, 32, keys_out, pos_out, base + L.sort_ws, L.sort_ws_bytes, st);
  if (rc) return rc;
  const unsigned rblocks = blocks > (unsigned)kEvalBlocks ? (unsigned)kEvalBlocks : blocks;
  XDFM_LAUNCH("eval_reduce_kernel", st, eval_reduce_kernel, dim3(rblocks), dim3(256), 0, keys_out, pos_out,
              (long long)n, part_rank, part_pos, part_ll);
  XDFM_LAUNCH("eval_final_kernel", st, eval_final_kernel, dim3(1), dim3(256), 0, part_rank, part_pos, part_ll,
              (int)rblocks, (long long)n, out);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_loss_total(const float* loss2, float* out2, void* stream) {
  XDFM_REQUIRE(loss2 && out2, "xdfm_loss_total: null argument");
  XDFM_LAUNCH("loss_pair_kernel", stream, loss_pair_kernel, dim3(1), dim3(64), 0, loss2, out2);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

XDFM_STAMP_UNIT(stamp_unit_optim)
