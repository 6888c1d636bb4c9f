// Fast ("low-rank") Compressed Interaction Network, one layer.
//
// Reference: recommenders/models/deeprec/models/xDeepFM.py:295-451 (`_build_fast_CIN`, taken when
// hparams.fast_CIN_d > 0).  With m = (sample, d), C = fast_CIN_d * L, c = l * fast_CIN_d + dd:
//
//   layer 0 (:330-351)   curr[m,l] = sum_dd 0.5 * (P[m,c] - Q[m,c]),   P = X0 . W,  Q = X0^2 . W^2
//                        (the reference does NOT square P; this follows it)
//   layer k (:352-382)   curr[m,l] = sum_dd P1[m,c] * P2[m,c],          P1 = X0 . W,  P2 = Xk . V
//
// W is [F][C], V is [H][C] (the reference's [1, F, C] / [1, H, C] conv1d filters).  The split of
// curr into the next layer's input (first n_hidden channels) and the pooled output is the CIN's
// (xDeepFM.py:419-436).  This branch is not on the benchmark path: the kernels are plain VALU
// code, one lane per row m, X tiles in LDS ([feature][row]), filter values through scalar
// loads (wave-uniform addresses), CG channels of P1/P2 per pass in registers.  Sums run in a
// fixed order; the filter gradients are reduced over row blocks in block order (no atomics).
#include <hip/hip_runtime.h>

#include "cin_stage.h"
#include "xdfm_common.h"

namespace xdfm_cin {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FcDims {
  int64_t B;
  int F, H, D, L, d, n_hidden, first;
};

constexpr int CG = 8;  // channels of P1 / P2 held in registers per pass

// P1[j] = sum_f x0[f] * W[f][c0+j], P2[j] = sum_h xk[h] * V[h][c0+j]   (layer 0: P2 = Q)
__device__ __forceinline__ void fc_products(const FcDims& c, int C, int c0, const float* as,
                                            const float* xs, int Tp, int row,
                                            const float* __restrict__ w, const float* __restrict__ v,
                                            float (&p1)[CG], float (&p2)[CG]) {
#pragma unroll
  for (int j = 0; j < CG; ++j) p1[j] = p2[j] = 0.f;
  const int last = C - 1 - c0;  // channels past C repeat the last one; their products are ignored
  for (int f = 0; f < c.F; ++f) {
    const float xv = as[f * Tp + row];
    const float* wr = w + f * C + c0;
    if (c.first) {
      const float xv2 = xv * xv;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        const float wv = wr[j < last ? j : last];
        p1[j] = fmaf(xv, wv, p1[j]);
        p2[j] = fmaf(xv2, wv * wv, p2[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CG; ++j) p1[j] = fmaf(xv, wr[j < last ? j : last], p1[j]);
    }
  }
  if (!c.first) {
    for (int h = 0; h < c.H; ++h) {
      const float xv = xs[h * Tp + row];
      const float* vr = v + h * C + c0;
#pragma unroll
      for (int j = 0; j < CG; ++j) p2[j] = fmaf(xv, vr[j < last ? j : last], p2[j]);
    }
  }
}

template <int ROWS>
__global__ __launch_bounds__(ROWS) void fast_cin_fwd_kernel(
    FcDims c, const float* __restrict__ x0, const float* __restrict__ xk,
    const float* __restrict__ w, const float* __restrict__ v, float* __restrict__ xnext,
    float* __restrict__ pooled, int pool_stride, int pool_off, int SPB) {
  extern __shared__ float lds[];
  constexpr int Tp = ROWS + kPad;
  const int C = c.d * c.L;
  float* as = lds;                                  // [F][Tp]
  float* xs = c.first ? as : as + c.F * Tp;         // [H][Tp]
  float* outs = xs + (c.first ? c.F : c.H) * Tp;    // [L][Tp]
  const int64_t b0 = (int64_t)blockIdx.x * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  stage_cols(x0, b0, nb, c.F, c.F, c.D, as, Tp);
  if (!c.first) stage_cols(xk, b0, nb, c.H, c.H, c.D, xs, Tp);
  const int row = threadIdx.x;
  for (int l = 0; l < c.L; ++l) outs[l * Tp + row] = 0.f;
  __syncthreads();

  for (int c0 = 0; c0 < C; c0 += CG) {
    float p1[CG], p2[CG];
    fc_products(c, C, c0, as, xs, Tp, row, w, v, p1, p2);
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      const int ch = c0 + j;  // wave-uniform
      if (ch < C) {
        const int l = ch / c.d;
        const float val = c.first ? 0.5f * (p1[j] - p2[j]) : p1[j] * p2[j];
        outs[l * Tp + row] += val;  // sum over dd in channel order
      }
    }
  }
  __syncthreads();
  const int cols = nb * c.D;
  for (int idx = threadIdx.x; idx < c.n_hidden * cols; idx += blockDim.x) {
    const int ch = idx / cols;
    const int rl = idx - ch * cols;
    const int bl = rl / c.D, d = rl - bl * c.D;
    xnext[((b0 + bl) * c.n_hidden + ch) * c.D + d] = outs[ch * Tp + rl];
  }
  const int npool = c.L - c.n_hidden;
  for (int idx = threadIdx.x; idx < npool * nb; idx += blockDim.x) {
    const int p = idx / nb;
    const int bl = idx - p * nb;
    float s = 0.f;
    for (int d = 0; d < c.D; ++d) s += outs[(c.n_hidden + p) * Tp + bl * c.D + d];
    pooled[(b0 + bl) * pool_stride + pool_off + p] = s;
  }
}

// Backward of one layer for a block of ROWS rows.  P1 / P2 are recomputed per pass.
//   g[m,l] = upstream gradient of curr;  layer k:  dP1 = g*P2, dP2 = g*P1;  layer 0: dP = g/2, dQ = -g/2
//   dX0[m,f] += sum_c dP1*W[f,c] (+ layer 0: dQ * 2*x*W^2),   dXk[m,h] = sum_c dP2*V[h,c]
//   dW[f,c]  = sum_m x0[m,f]*dP1[m,c] (+ layer 0: 2*W[f,c] * sum_m x0^2*dQ),   dV[h,c] = sum_m xk[m,h]*dP2[m,c]
// part[block][(f | F+h) * C + c] receives this block's share of dW / dV.
template <int ROWS>
__global__ __launch_bounds__(ROWS) void fast_cin_bwd_kernel(
    FcDims c, const float* __restrict__ x0, const float* __restrict__ xk,
    const float* __restrict__ w, const float* __restrict__ v, const float* __restrict__ d_xnext,
    const float* __restrict__ d_pooled, int pool_stride, int pool_off, float* d_xk, float* d_x0,
    float* __restrict__ part, int SPB) {
  extern __shared__ float lds[];
  constexpr int Tp = ROWS + kPad;
  const int C = c.d * c.L;
  const int nfeat = c.F + (c.first ? 0 : c.H);
  float* as = lds;                                    // [F][Tp]   x0
  float* xs = c.first ? as : as + c.F * Tp;           // [H][Tp]   xk
  float* das = xs + (c.first ? c.F : c.H) * Tp;       // [F][Tp]   dX0
  float* dxs = das + c.F * Tp;                        // [H][Tp]   dXk (layer k only)
  float* gs = dxs + (c.first ? 0 : c.H) * Tp;         // [L][Tp]   upstream gradient
  float* dps = gs + c.L * Tp;                         // [2][CG][Tp]  dP1 / dP2 of the pass
  const int64_t b0 = (int64_t)blockIdx.x * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  const int cols = nb * c.D;
  stage_cols(x0, b0, nb, c.F, c.F, c.D, as, Tp);
  if (!c.first) stage_cols(xk, b0, nb, c.H, c.H, c.D, xs, Tp);
  const int row = threadIdx.x;
  {
    const bool valid = row < cols;
    const int rc = valid ? row : 0;
    const int bl = rc / c.D, d = rc - bl * c.D;
    const int64_t b = b0 + bl;
    for (int l = 0; l < c.L; ++l) {
      float g = 0.f;
      if (valid)
        g = l < c.n_hidden ? d_xnext[(b * c.n_hidden + l) * c.D + d]
                           : d_pooled[b * pool_stride + pool_off + l - c.n_hidden];
      gs[l * Tp + row] = g;
    }
    for (int f = 0; f < c.F; ++f) das[f * Tp + row] = 0.f;
    if (!c.first)
      for (int h = 0; h < c.H; ++h) dxs[h * Tp + row] = 0.f;
  }
  __syncthreads();

  float* mypart = part + (size_t)blockIdx.x * nfeat * C;
  for (int c0 = 0; c0 < C; c0 += CG) {
    float p1[CG], p2[CG], dp1[CG], dp2[CG];
    fc_products(c, C, c0, as, xs, Tp, row, w, v, p1, p2);
    const int last = C - 1 - c0;
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      const int ch = c0 + j;
      float g = 0.f;
      if (ch < C) g = gs[(ch / c.d) * Tp + row];
      dp1[j] = c.first ? 0.5f * g : g * p2[j];
      dp2[j] = c.first ? -0.5f * g : g * p1[j];
      dps[j * Tp + row] = dp1[j];
      dps[(CG + j) * Tp + row] = dp2[j];
    }
    // input gradients of this lane's row
    for (int f = 0; f < c.F; ++f) {
      const float* wr = w + f * C + c0;
      float s = 0.f;
      if (c.first) {
        const float xv2 = 2.f * as[f * Tp + row];
#pragma unroll
        for (int j = 0; j < CG; ++j) {
          const float wv = wr[j < last ? j : last];
          s = fmaf(dp1[j], wv, s);
          s = fmaf(dp2[j] * xv2, wv * wv, s);
        }
      } else {
#pragma unroll
        for (int j = 0; j < CG; ++j) s = fmaf(dp1[j], wr[j < last ? j : last], s);
      }
      das[f * Tp + row] += s;
    }
    if (!c.first) {
      for (int h = 0; h < c.H; ++h) {
        const float* vr = v + h * C + c0;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CG; ++j) s = fmaf(dp2[j], vr[j < last ? j : last], s);
        dxs[h * Tp + row] += s;
      }
    }
    __syncthreads();  // dps of all rows are in LDS
    // filter gradients of the pass: one (feature, channel) pair per thread, summed over the rows
    for (int idx = threadIdx.x; idx < nfeat * CG; idx += blockDim.x) {
      const int feat = idx / CG;
      const int j = idx - feat * CG;
      const int ch = c0 + j;
      if (ch >= C) continue;
      const bool is_w = feat < c.F;
      const float* xrow = is_w ? as + feat * Tp : xs + (feat - c.F) * Tp;
      const float* drow = dps + ((is_w ? 0 : CG) + j) * Tp;
      float s = 0.f, s2 = 0.f;
      for (int r = 0; r < ROWS; r += 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + r);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(drow + r);
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fmaf(xv[q], dv[q], s);
        if (c.first) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(dps + (CG + j) * Tp + r);
#pragma unroll
          for (int q = 0; q < 4; ++q) s2 = fmaf(xv[q] * xv[q], qv[q], s2);
        }
      }
      if (c.first) s = fmaf(2.f * w[feat * C + ch], s2, s);
      mypart[feat * C + ch] = s;
    }
    __syncthreads();  // before the next pass overwrites dps
  }
  // d_x0[b][f][d] += dX0, d_xk[b][h][d] = dXk
  {
    const int total = nb * c.F * c.D;
    float* base = d_x0 + b0 * c.F * c.D;
    const float invFD = 1.0f / (float)(c.F * c.D), invD = 1.0f / (float)c.D;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int bl = fdiv(idx, invFD);
      const int rem = idx - bl * c.F * c.D;
      const int f = fdiv(rem, invD);
      const int d = rem - f * c.D;
      base[idx] += das[f * Tp + bl * c.D + d];
    }
  }
  if (!c.first) {
    const int total = nb * c.H * c.D;
    float* base = d_xk + b0 * c.H * c.D;
    const float invHD = 1.0f / (float)(c.H * c.D), invD = 1.0f / (float)c.D;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int bl = fdiv(idx, invHD);
      const int rem = idx - bl * c.H * c.D;
      const int h = fdiv(rem, invD);
      const int d = rem - h * c.D;
      base[idx] = dxs[h * Tp + bl * c.D + d];
    }
  }
}

// out[i] = sum over blocks of part[blk][i], in block order, 8 loads in flight
__global__ __launch_bounds__(256) void fast_cin_reduce_kernel(const float* __restrict__ part, int n_blocks,
                                                              int FC, int total, float* __restrict__ d_w,
                                                              float* __restrict__ d_v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float s = 0.f;
  for (int k0 = 0; k0 < n_blocks; k0 += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      t[u] = part[(size_t)(k < n_blocks ? k : k0) * total + i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (k0 + u < n_blocks) ? t[u] : 0.f;
  }
  if (i < FC)
    d_w[i] = s;
  else
    d_v[i - FC] = s;
}

constexpr int kFwdRows = 256, kBwdRows = 128;
constexpr size_t kLds = 160 * 1024;

size_t fwd_lds(int F, int H, int L, int first) {
  return (size_t)(F + (first ? 0 : H) + L) * (kFwdRows + kPad) * sizeof(float);
}
size_t bwd_lds(int F, int H, int L, int first) {
  return (size_t)(2 * F + (first ? 0 : 2 * H) + L + 2 * CG) * (kBwdRows + kPad) * sizeof(float);
}

int check(const char* who, int64_t B, int F, int H, int D, int L, int d, int n_hidden, int first) {
  XDFM_REQUIRE(B >= 0 && F > 0 && H > 0 && D > 0 && L > 0 && d > 0, "%s: bad sizes", who);
  XDFM_REQUIRE(n_hidden >= 0 && n_hidden <= L, "%s: n_hidden %d outside [0,%d]", who, n_hidden, L);
  XDFM_REQUIRE(D <= kBwdRows, "%s: dim %d not supported (max %d)", who, D, kBwdRows);
  XDFM_REQUIRE((int64_t)(F + H) * d * L < ((int64_t)1 << 31), "%s: filter too large", who);
  XDFM_REQUIRE(!first || H == F, "%s: layer 0 has H == F", who);
  return XDFM_OK;
}

}  // namespace
}  // namespace xdfm_cin

using namespace xdfm_cin;

extern "C" int xdfm_fast_cin_fwd(const float* x0, const float* xk, const float* w, const float* v,
                                 int64_t B, int32_t F, int32_t H, int32_t D, int32_t L, int32_t d,
                                 int32_t n_hidden, int32_t first, float* xnext, float* pooled,
                                 int32_t pool_stride, int32_t pool_off, void* stream) {
  int rc = check("xdfm_fast_cin_fwd", B, F, H, D, L, d, n_hidden, first);
  if (rc) return rc;
  XDFM_REQUIRE(x0 && w && (first || (xk && v)), "xdfm_fast_cin_fwd: null input");
  XDFM_REQUIRE(n_hidden == 0 || xnext, "xdfm_fast_cin_fwd: xnext is null");
  XDFM_REQUIRE(n_hidden == L || pooled, "xdfm_fast_cin_fwd: pooled is null");
  if (B == 0) return XDFM_OK;
  const size_t lds_bytes = fwd_lds(F, H, L, first);
  XDFM_REQUIRE(lds_bytes <= kLds, "xdfm_fast_cin_fwd: tile does not fit the LDS (F=%d H=%d L=%d)", F, H, L);
  if (lds_bytes > 64 * 1024)
    XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fast_cin_fwd_kernel<kFwdRows>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const int SPB = kFwdRows / D;
  FcDims c{B, F, H, D, L, d, n_hidden, first};
  XDFM_LAUNCH("fast_cin_fwd_kernel", stream, fast_cin_fwd_kernel<kFwdRows>, dim3((unsigned)((B + SPB - 1) / SPB)),
              dim3(kFwdRows), lds_bytes, c, x0, first ? x0 : xk, w, v, xnext, pooled, pool_stride, pool_off, SPB);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_fast_cin_bwd_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                                  int32_t d, size_t* floats) {
  XDFM_REQUIRE(floats && B >= 0 && F > 0 && H > 0 && D > 0 && D <= kBwdRows && L > 0 && d > 0,
               "xdfm_fast_cin_bwd_workspace_floats: bad argument");
  const int SPB = kBwdRows / D;
  *floats = (size_t)((B + SPB - 1) / SPB) * (F + H) * d * L;
  return XDFM_OK;
}

extern "C" int xdfm_fast_cin_bwd(const float* x0, const float* xk, const float* w, const float* v,
                                 int64_t B, int32_t F, int32_t H, int32_t D, int32_t L, int32_t d,
                                 int32_t n_hidden, int32_t first, const float* d_xnext,
                                 const float* d_pooled, int32_t pool_stride, int32_t pool_off,
                                 float* d_xk, float* d_x0, float* d_w, float* d_v, float* workspace,
                                 void* stream) {
  int rc = check("xdfm_fast_cin_bwd", B, F, H, D, L, d, n_hidden, first);
  if (rc) return rc;
  XDFM_REQUIRE(x0 && w && d_x0 && d_w && workspace, "xdfm_fast_cin_bwd: null argument");
  XDFM_REQUIRE(first || (xk && v && d_xk && d_v), "xdfm_fast_cin_bwd: layer k needs xk, v, d_xk, d_v");
  XDFM_REQUIRE(n_hidden == 0 || d_xnext, "xdfm_fast_cin_bwd: d_xnext is null");
  XDFM_REQUIRE(n_hidden == L || d_pooled, "xdfm_fast_cin_bwd: d_pooled is null");
  hipStream_t st = (hipStream_t)stream;
  const int C = d * L;
  if (B == 0) {
    XDFM_HIP(hipMemsetAsync(d_w, 0, (size_t)F * C * sizeof(float), st));
    if (!first) XDFM_HIP(hipMemsetAsync(d_v, 0, (size_t)H * C * sizeof(float), st));
    return XDFM_OK;
  }
  const size_t lds_bytes = bwd_lds(F, H, L, first);
  XDFM_REQUIRE(lds_bytes <= kLds, "xdfm_fast_cin_bwd: tile does not fit the LDS (F=%d H=%d L=%d)", F, H, L);
  if (lds_bytes > 64 * 1024)
    XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fast_cin_bwd_kernel<kBwdRows>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const int SPB = kBwdRows / D;
  const int n_blocks = (int)((B + SPB - 1) / SPB);
  FcDims c{B, F, H, D, L, d, n_hidden, first};
  XDFM_LAUNCH("fast_cin_bwd_kernel", st, fast_cin_bwd_kernel<kBwdRows>, dim3((unsigned)n_blocks), dim3(kBwdRows),
              lds_bytes, c, x0, first ? x0 : xk, w, v, d_xnext, d_pooled, pool_stride, pool_off, d_xk, d_x0,
              workspace, SPB);
  const int total = (F + (first ? 0 : H)) * C;
  XDFM_LAUNCH("fast_cin_reduce_kernel", st, fast_cin_reduce_kernel, dim3((unsigned)((total + 255) / 256)),
              dim3(256), 0, workspace, n_blocks, F * C, total, d_w, d_v);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}
