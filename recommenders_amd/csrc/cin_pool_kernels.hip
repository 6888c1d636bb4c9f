// The pooled ("direct") channels of a CIN layer without the factor D of work.
//
// xDeepFM.py:253-270: the channels of a layer that do not feed the next layer are summed over the
// embedding dimension, pooled[b,j] = sum_d curr[b,d,j].  Since curr[b,d,j] = sum_{f,h} s M[f,h]
// x0[b,f,d] xk[b,h,d] W[(f,h),j], the sum over d can be taken FIRST:
//     S[b,(f,h)] = sum_d x0[b,f,d] xk[b,h,d]                 (one small product per sample)
//     pooled[b,j] = sum_{(f,h)} S[b,(f,h)] Wp[(f,h),j]        (a plain [B,P] x [P,Lp] product)
// which is the same number (another association of the same fp32 sums) for 1/D of the multiply-adds
// of the per-(b,d) contraction: half of every hidden layer's channels and all of the last layer's.
// Backward: dS = d_pooled Wp^T, dWp = S^T d_pooled (the dense-layer kernels), and
//     d_x0[b,f,:] += sum_h dS[b,(f,h)] xk[b,h,:],   d_xk[b,h,:] += sum_f dS[b,(f,h)] x0[b,f,:].
// Pairs are enumerated p = f*H + h, or, for the masked first layer (f < h, times 2:
// xDeepFM.py:208-217), p = index of (f,h) in the row-major strict upper triangle; the factor 2 is
// folded into the packed filter slice.
//
// HBM-bound kernels (S is B x P floats: 24 MB at the bench shape); the per-sample products run on the
// fp32 matrix cores, one wave per sample, a group of samples staged per block.
#include <hip/hip_runtime.h>

#include "cin_common.h"
#include "xdfm_common.h"

namespace {

__host__ __device__ inline int n_pairs(int F, int H, int masked) { return masked ? F * (F - 1) / 2 : F * H; }
// first pair index of row f in the strict upper triangle of an F x F matrix
__host__ __device__ inline int tri_row_start(int f, int F) { return f * (2 * F - f - 1) / 2; }

// LDS floats of a sample's dS row in the backward kernel: compact on the masked layer (triangle),
// [Fp][Hp + 1] zero-padded otherwise (rows one bank apart: 16 lanes read 16 different f at one h;
// no bounds to test in the product loop)
__host__ __device__ inline int g_floats(int F, int H, int masked) {
  return ((masked ? n_pairs(F, H, 1) : ((F + 15) & ~15) * (((H + 15) & ~15) + 1)) + 3) & ~3;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PairDims {
  int64_t B;
  int F, H, D, masked;
  int Fp, Hp;   // rows padded to the 16-row MFMA tiles
  int Dp;       // LDS pitch of a feature row: the d axis padded to a multiple of 4, + 1 (odd: the
                // fragment reads of 16 different rows fall on 16 different banks)
  int NS;       // samples per block
  int shared_x; // xk == x0
};

// `ns` samples' [R][D] blocks (contiguous in global memory, `sample_stride` floats apart) -> LDS
// [s][Rp][Dp] (`per` floats apart), everything outside [R) x [D) zero.  Eight loads in flight per lane.
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int64_t sample_stride, int ns, int R,
                                           int Rp, int D, int Dp, float* __restrict__ dst, int per) {
  const int tile = Rp * Dp, total = ns * tile;
  const float inv_tile = 1.0f / (float)tile, inv_dp = 1.0f / (float)Dp;
  constexpr int U = 8;
  if ((D & 3) == 0 && (sample_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    // rows of whole float4s: 16-byte loads, eight in flight; the padding is cleared first
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      int s = (int)((float)i * inv_tile);
      s += ((s + 1) * tile <= i) - (s * tile > i);
      const int j = i - s * tile;
      int r = (int)((float)j * inv_dp);
      r += ((r + 1) * Dp <= j) - (r * Dp > j);
      if (r >= R || j - r * Dp >= D) dst[s * per + j] = 0.f;
    }
    const int q = D >> 2, per_s = R * q, n4 = ns * per_s;
    const float inv_ps = 1.0f / (float)per_s, inv_q = 1.0f / (float)q;
    for (int i0 = threadIdx.x; i0 < n4; i0 += blockDim.x * U) {
      float4 v[U];
      int at[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * blockDim.x;
        int s = (int)((float)i * inv_ps);
        s += ((s + 1) * per_s <= i) - (s * per_s > i);
        const int j = i - s * per_s;
        int r = (int)((float)j * inv_q);
        r += ((r + 1) * q <= j) - (r * q > j);
        const int d4 = j - r * q;
        at[u] = i < n4 ? s * per + r * Dp + 4 * d4 : -1;
        v[u] = i < n4 ? *reinterpret_cast<const float4*>(src + s * sample_stride + r * D + 4 * d4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (at[u] >= 0) {
          dst[at[u]] = v[u].x;
          dst[at[u] + 1] = v[u].y;
          dst[at[u] + 2] = v[u].z;
          dst[at[u] + 3] = v[u].w;
        }
    }
    return;
  }
  for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * U) {
    float v[U];
    int at[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * blockDim.x;
      int s = (int)((float)i * inv_tile);
      s += ((s + 1) * tile <= i) - (s * tile > i);  // exact floor
      const int j = i - s * tile;
      int r = (int)((float)j * inv_dp);
      r += ((r + 1) * Dp <= j) - (r * Dp > j);
      const int d = j - r * Dp;
      at[u] = i < total ? s * per + j : -1;
      v[u] = (i < total && r < R && d < D) ? src[s * sample_stride + r * D + d] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (at[u] >= 0) dst[at[u]] = v[u];
  }
}

// ---------------------------------------------------------------- S = sum_d x0 * xk
// S_b = X0_b Xk_b^T on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32), one wave per
// sample, NS = 4 samples per block.  Lane L of an accumulator holds
// S[f = 16 ft + 4 (L >> 4) + r][h = 16 ht + (L & 15)]: 16 consecutive h, i.e. 16 consecutive pairs.
__global__ __launch_bounds__(256) void cin_pair_sum_kernel(PairDims c, const float* __restrict__ x0,
                                                           const float* __restrict__ xk, float* __restrict__ S) {
  extern __shared__ float lds[];
  const int per_a = c.Fp * c.Dp, per_b = c.shared_x ? 0 : c.Hp * c.Dp;
  const int per = per_a + per_b + 16;
  const int64_t b0 = (int64_t)blockIdx.x * c.NS;
  const int ns = (int)(c.B - b0 < c.NS ? c.B - b0 : c.NS);
  stage_rows(x0 + b0 * c.F * c.D, (int64_t)c.F * c.D, ns, c.F, c.Fp, c.D, c.Dp, lds, per);
  if (!c.shared_x) stage_rows(xk + b0 * c.H * c.D, (int64_t)c.H * c.D, ns, c.H, c.Hp, c.D, c.Dp, lds + per_a, per);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int nft = c.Fp >> 4, nht = c.Hp >> 4, nks = (c.D + 3) >> 2;
  const int P = n_pairs(c.F, c.H, c.masked);
  // (sample, f tile, h tile) items dealt to the block's waves in turn
  for (int it = wave; it < ns * nft * nht; it += 4) {
    const int s = it / (nft * nht), t = it - s * (nft * nht);
    const int ft = t / nht, ht = t - ft * nht;
    if (c.masked && ht < ft) continue;
    const float* a = lds + s * per;
    const float* bq = c.shared_x ? a : a + per_a;
    float* out = S + (b0 + s) * P;
    {
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ap = a + (16 * ft + li) * c.Dp + lk;
        const float* bp = bq + (16 * ht + li) * c.Dp + lk;
        for (int ks = 0; ks < nks; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks], bp[4 * ks], acc, 0, 0, 0);
        const int h = 16 * ht + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * ft + 4 * lk + r;
          if (f < c.F && h < c.H && (!c.masked || h > f))
            out[c.masked ? tri_row_start(f, c.F) + h - f - 1 : f * c.H + h] = acc[r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------- backward of the pair sums
// With G[f][h] = dS[b,(f,h)] (masked layer: the symmetric G + G^T, both roles are x0):
//   d_x0_b = G xk_b   (M = f, N = d, K = h),   d_xk_b = G^T x0_b   (M = h, N = d, K = f)
// on the fp32 matrix cores, one wave per sample; the G fragments are read straight out of the
// sample's compact dS row in LDS.  x0_mode / xk_mode: 0 = write, 1 = add.
__device__ __forceinline__ float g_at(const float* __restrict__ g, const PairDims& c, int f, int h) {
  if (f >= c.F || h >= c.H) return 0.f;
  if (!c.masked) return g[f * (c.Hp + 1) + h];
  if (f == h) return 0.f;
  const int lo = f < h ? f : h, hi = f < h ? h : f;
  return g[tri_row_start(lo, c.F) + hi - lo - 1];
}

__global__ __launch_bounds__(256) void cin_pair_bwd_kernel(PairDims c, const float* __restrict__ x0,
                                                           const float* __restrict__ xk, const float* __restrict__ dS,
                                                           float* __restrict__ d_x0, int x0_mode,
                                                           float* __restrict__ d_xk, int xk_mode) {
  extern __shared__ float lds[];
  const int per_a = c.Fp * c.Dp, per_b = c.shared_x ? 0 : c.Hp * c.Dp;
  const int P = n_pairs(c.F, c.H, c.masked);
  const int Pp = g_floats(c.F, c.H, c.masked);
  const int per = per_a + per_b + 16 + Pp;
  const int64_t b0 = (int64_t)blockIdx.x * c.NS;
  const int ns = (int)(c.B - b0 < c.NS ? c.B - b0 : c.NS);
  stage_rows(x0 + b0 * c.F * c.D, (int64_t)c.F * c.D, ns, c.F, c.Fp, c.D, c.Dp, lds, per);
  if (!c.shared_x) stage_rows(xk + b0 * c.H * c.D, (int64_t)c.H * c.D, ns, c.H, c.Hp, c.D, c.Dp, lds + per_a, per);
  if (c.masked)
    stage_rows(dS + b0 * P, (int64_t)P, ns, 1, 1, P, Pp, lds + per_a + per_b + 16, per);
  else
    stage_rows(dS + b0 * P, (int64_t)P, ns, c.F, c.Fp, c.H, c.Hp + 1, lds + per_a + per_b + 16, per);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int nft = c.Fp >> 4, nht = c.Hp >> 4, ndt = (c.D + 15) >> 4;
  // (sample, output tile) items dealt to the block's waves in turn: first the d_x0 = G * xk tiles
  // (f tile x d tile), then the d_xk = G^T * x0 tiles (h tile x d tile)
  const int n0 = nft * ndt, n1 = c.shared_x ? 0 : nht * ndt;
  for (int it = wave; it < ns * (n0 + n1); it += 4) {
    const int s = it / (n0 + n1), t = it - s * (n0 + n1);
    const float* a = lds + s * per;
    const float* bq = c.shared_x ? a : a + per_a;
    const float* g = a + per_a + per_b + 16;
    if (t < n0) {
      const int ft = t / ndt, dt = t - ft * ndt;
      const int d = 16 * dt + li;
      // the values to add to are requested before the products, not after them
      float old[4] = {0.f, 0.f, 0.f, 0.f};
      if (x0_mode) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * ft + 4 * lk + r;
          if (f < c.F && d < c.D) old[r] = d_x0[((b0 + s) * c.F + f) * c.D + d];
        }
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* xp = bq + lk * c.Dp + 16 * dt + li;
      if (!c.masked) {
        // zero-padded [Fp][Hp + 1] image: plain strided reads, two independent chains (Hp / 4 is even)
        const int Gp = c.Hp + 1;
        const float* gp = g + (16 * ft + li) * Gp + lk;
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < (c.Hp >> 2); ks += 2) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[4 * ks], xp[4 * ks * c.Dp], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[4 * ks + 4], xp[(4 * ks + 4) * c.Dp], acc1, 0, 0, 0);
        }
        acc += acc1;
      } else {
        for (int ks = 0; ks < (c.Hp >> 2); ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(g_at(g, c, 16 * ft + li, 4 * ks + lk), xp[4 * ks * c.Dp], acc, 0,
                                                     0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * ft + 4 * lk + r;
        if (f < c.F && d < c.D) d_x0[((b0 + s) * c.F + f) * c.D + d] = old[r] + acc[r];
      }
    } else {
      const int u = t - n0;
      const int ht = u / ndt, dt = u - ht * ndt;
      const int d = 16 * dt + li;
      float old[4] = {0.f, 0.f, 0.f, 0.f};
      if (xk_mode) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int h = 16 * ht + 4 * lk + r;
          if (h < c.H && d < c.D) old[r] = d_xk[((b0 + s) * c.H + h) * c.D + d];
        }
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* xp = a + lk * c.Dp + 16 * dt + li;
      if (!c.masked) {
        const int Gp = c.Hp + 1;
        const float* gp = g + lk * Gp + 16 * ht + li;
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < (c.Fp >> 2); ks += 2) {  // Fp / 4 is even as well
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[4 * ks * Gp], xp[4 * ks * c.Dp], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[(4 * ks + 4) * Gp], xp[(4 * ks + 4) * c.Dp], acc1, 0, 0, 0);
        }
        acc += acc1;
      } else {
        for (int ks = 0; ks < (c.Fp >> 2); ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(g_at(g, c, 4 * ks + lk, 16 * ht + li), xp[4 * ks * c.Dp], acc, 0,
                                                     0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = 16 * ht + 4 * lk + r;
        if (h < c.H && d < c.D) d_xk[((b0 + s) * c.H + h) * c.D + d] = old[r] + acc[r];
      }
    }
  }
}

// ---------------------------------------------------------------- filter slice in pair order
// wp[p][j] = s * W[(f,h), n_hidden + j]   (s = 2 on the masked layer);  and back:
// d_w[(f,h), n_hidden + j] = s * dwp[p][j], masked-out rows get 0.
__global__ void cin_pool_pack_kernel(const float* __restrict__ w, int F, int H, int L, int nh, int masked,
                                     float* __restrict__ wp) {
  const int Lp = L - nh;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)F * H * Lp) return;
  const int r = (int)(idx / Lp), j = (int)(idx - (int64_t)r * Lp);
  const int f = r / H, h = r - f * H;
  if (masked && h <= f) return;
  const int p = masked ? tri_row_start(f, F) + h - f - 1 : r;
  wp[(int64_t)p * Lp + j] = w[(int64_t)r * L + nh + j] * (masked ? 2.f : 1.f);
}

__global__ void cin_pool_unpack_kernel(const float* __restrict__ dwp, int F, int H, int L, int nh, int masked,
                                       float* __restrict__ d_w) {
  const int Lp = L - nh;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)F * H * Lp) return;
  const int r = (int)(idx / Lp), j = (int)(idx - (int64_t)r * Lp);
  const int f = r / H, h = r - f * H;
  float v = 0.f;
  if (!(masked && h <= f)) {
    const int p = masked ? tri_row_start(f, F) + h - f - 1 : r;
    v = dwp[(int64_t)p * Lp + j] * (masked ? 2.f : 1.f);
  }
  d_w[(int64_t)r * L + nh + j] = v;
}

PairDims pair_dims(int64_t B, int F, int H, int D, int masked, bool shared_x, bool bwd, size_t* lds_bytes) {
  PairDims c;
  c.B = B;
  c.F = F;
  c.H = H;
  c.D = D;
  c.masked = masked ? 1 : 0;
  c.Fp = (F + 15) & ~15;
  c.Hp = (H + 15) & ~15;
  c.Dp = ((D + 3) & ~3) + 1;
  c.shared_x = shared_x ? 1 : 0;
  size_t per = (size_t)c.Fp * c.Dp + (shared_x ? 0 : (size_t)c.Hp * c.Dp) + 16;
  if (bwd) per += (size_t)g_floats(F, H, masked);
  // one wave per sample; fewer samples per block when four do not fit 64 KB
  c.NS = 4;
  while (c.NS > 1 && per * c.NS * sizeof(float) > 64 * 1024) c.NS >>= 1;
  *lds_bytes = per * c.NS * sizeof(float);
  return c;
}

int check(const char* who, int64_t B, int F, int H, int D, int masked) {
  XDFM_REQUIRE(B >= 0 && F > 0 && H > 0 && D > 0, "%s: bad sizes", who);
  XDFM_REQUIRE(!masked || F == H, "%s: the masked layer pairs the fields with themselves (H == F)", who);
  return XDFM_OK;
}

}  // namespace

extern "C" int xdfm_cin_pair_count(int32_t F, int32_t H, int32_t masked, int64_t* pairs) {
  XDFM_REQUIRE(pairs && F > 0 && H > 0, "xdfm_cin_pair_count: bad argument");
  *pairs = n_pairs(F, H, masked);
  return XDFM_OK;
}

extern "C" int xdfm_cin_pair_sum(const float* x0, const float* xk, int64_t B, int32_t F, int32_t H, int32_t D,
                                 int32_t masked, float* S, void* stream) {
  int rc = check("xdfm_cin_pair_sum", B, F, H, D, masked);
  if (rc) return rc;
  if (B == 0) return XDFM_OK;
  XDFM_REQUIRE(x0 && xk && S, "xdfm_cin_pair_sum: null argument");
  XDFM_REQUIRE(!masked || xk == x0, "xdfm_cin_pair_sum: the masked layer has xk == x0");
  size_t lds = 0;
  const PairDims c = pair_dims(B, F, H, D, masked, xk == x0, false, &lds);
  XDFM_REQUIRE(lds <= 160 * 1024, "xdfm_cin_pair_sum: a sample's rows do not fit the LDS");
  XDFM_HIP(hipFuncSetAttribute((const void*)cin_pair_sum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  XDFM_LAUNCH("cin_pair_sum_kernel", stream, cin_pair_sum_kernel, dim3((unsigned)((B + c.NS - 1) / c.NS)), dim3(256),
              lds, c, x0, xk, S);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_pair_bwd(const float* x0, const float* xk, const float* dS, int64_t B, int32_t F, int32_t H,
                                 int32_t D, int32_t masked, float* d_x0, int32_t accumulate_x0, float* d_xk,
                                 int32_t accumulate_xk, void* stream) {
  int rc = check("xdfm_cin_pair_bwd", B, F, H, D, masked);
  if (rc) return rc;
  if (B == 0) return XDFM_OK;
  XDFM_REQUIRE(x0 && xk && dS && d_x0, "xdfm_cin_pair_bwd: null argument");
  XDFM_REQUIRE((xk == x0) == (d_xk == nullptr), "xdfm_cin_pair_bwd: d_xk must be NULL exactly when xk == x0");
  XDFM_REQUIRE(!masked || xk == x0, "xdfm_cin_pair_bwd: the masked layer has xk == x0");
  size_t lds = 0;
  const PairDims c = pair_dims(B, F, H, D, masked, xk == x0, true, &lds);
  XDFM_REQUIRE(lds <= 160 * 1024, "xdfm_cin_pair_bwd: a sample's rows do not fit the LDS");
  XDFM_HIP(hipFuncSetAttribute((const void*)cin_pair_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  XDFM_LAUNCH("cin_pair_bwd_kernel", stream, cin_pair_bwd_kernel, dim3((unsigned)((B + c.NS - 1) / c.NS)), dim3(256),
              lds, c, x0, xk, dS, d_x0, accumulate_x0 ? 1 : 0, d_xk, accumulate_xk ? 1 : 0);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_pool_pack(const float* w, int32_t F, int32_t H, int32_t L, int32_t n_hidden, int32_t masked,
                                  float* wp, void* stream) {
  XDFM_REQUIRE(w && wp && F > 0 && H > 0 && L > 0 && n_hidden >= 0 && n_hidden < L, "xdfm_cin_pool_pack: bad argument");
  const int64_t total = (int64_t)F * H * (L - n_hidden);
  XDFM_LAUNCH("cin_pool_pack_kernel", stream, cin_pool_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
              w, F, H, L, n_hidden, masked, wp);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_pool_unpack(const float* dwp, int32_t F, int32_t H, int32_t L, int32_t n_hidden,
                                    int32_t masked, float* d_w, void* stream) {
  XDFM_REQUIRE(dwp && d_w && F > 0 && H > 0 && L > 0 && n_hidden >= 0 && n_hidden < L,
               "xdfm_cin_pool_unpack: bad argument");
  const int64_t total = (int64_t)F * H * (L - n_hidden);
  XDFM_LAUNCH("cin_pool_unpack_kernel", stream, cin_pool_unpack_kernel, dim3((unsigned)((total + 255) / 256)),
              dim3(256), 0, dwp, F, H, L, n_hidden, masked, d_w);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}
