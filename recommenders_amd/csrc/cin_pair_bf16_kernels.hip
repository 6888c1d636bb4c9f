// The pair sums of the pooled channels (xdfm_cin_pair_sum / xdfm_cin_pair_bwd, cin_pool_kernels.hip) on the
// bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulation), for the wide layers of the bf16
// path (BASELINE configs[4]: F = 39, H = 128, D = 64): per sample
//     S      = X0 Xk^T            [F x H],  contraction over d
//     d_X0   = G Xk               [F x D],  contraction over h       (G = dS of the sample, [F x H])
//     d_Xk   = G^T X0             [H x D],  contraction over f
// (xDeepFM.py:253-270: the channels summed over the embedding dimension, with that sum taken first).
// One wave per sample, no LDS and no barrier: a lane's operand fragment is 8 consecutive values of the
// contraction axis; where that axis is the fastest one in memory (d for S, h for d_X0's G) the fragment is
// two 16-byte loads, elsewhere eight 4-byte loads whose 32 lanes read 128 consecutive bytes.  The fp32 values
// are rounded to bf16 (round to nearest even) as they are packed; sums are fp32.  These kernels move
// 0.5 - 1.2 GB per launch at configs[4]'s sizes and the fp32 forms reached 0.26 - 0.34 of the HBM rate (one
// sample's rows staged through 46 - 79 KB of LDS per block, a barrier, 16 times the matrix instructions).
//
// The masked first layer (xk == x0, pairs f < h in the row-major strict upper triangle): S is the full square's
// product of which the triangle is stored; its backward is ONE product, d_X0 = (G + G^T) X0, with the symmetric
// matrix read element by element out of the compact triangle.
//
// Served: D a multiple of 16 (sums) / 32 (backward) up to 64 and F <= 64; the unmasked backward also wants H a
// multiple of 8.  The entry points hand every other shape to the exact fp32 kernels.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cin_bf16.h"
#include "stamp.h"
#include "xdfm_common.h"

using namespace xdfm_bf16;

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// first pair index of row f in the strict upper triangle of an F x F matrix (cin_pool_kernels.hip)
__host__ __device__ inline int tri_start(int f, int F) { return f * (2 * F - f - 1) / 2; }

__device__ __forceinline__ bf16x8 frag_of(const float (&v)[8]) {
  uint4 p;
  p.x = pack_bf16(v[0], v[1]);
  p.y = pack_bf16(v[2], v[3]);
  p.z = pack_bf16(v[4], v[5]);
  p.w = pack_bf16(v[6], v[7]);
  return as_frag(p);
}

// 8 consecutive floats at p (16-byte aligned), or zeros
__device__ __forceinline__ bf16x8 frag_rows(const float* __restrict__ p, bool on) {
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (on) {
    const f32x4v a = *reinterpret_cast<const f32x4v*>(p);
    const f32x4v b = *reinterpret_cast<const f32x4v*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
  return frag_of(v);
}

// 8 floats `stride` apart starting at p; element i is taken when i < n (n <= 0: none), zeros elsewhere
__device__ __forceinline__ bf16x8 frag_strided(const float* __restrict__ p, int64_t stride, int n, bool on) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (on && i < n) ? p[(int64_t)i * stride] : 0.f;
  return frag_of(v);
}

__device__ __forceinline__ f32x16 zero16() {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return z;
}

// ---------------------------------------------------------------- S = X0 Xk^T
// NFT 32-row tiles of f, NKS = D / 16 steps over d.  The X0 fragments stay in registers for the sample; per
// 32 rows of Xk: NKS fragment loads, NFT * NKS matrix instructions, the NFT result tiles stored (a register
// holds 32 consecutive h of one f: 128-byte stores).
template <int NFT, int NKS, bool MASKED>
__global__ __launch_bounds__(256) void cin_pair_sum_bf16_kernel(int64_t B, int F, int H, const float* __restrict__ x0,
                                                                const float* __restrict__ xk, float* __restrict__ S) {
  xdfm::Stamp stamp_(xdfm::kStampCinFwd);
  constexpr int D = 16 * NKS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, half = lane >> 5;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;  // (wave-uniform; no barrier in this kernel)
  const float* xa = x0 + b * F * D;
  const float* xb = xk + b * H * D;
  float* out = S + b * (MASKED ? F * (F - 1) / 2 : F * H);
  bf16x8 A[NFT][NKS];
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) {
    const int f = 32 * ft + lr;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) A[ft][ks] = frag_rows(xa + f * D + 16 * ks + 8 * half, f < F);
  }
  for (int h0 = 0; h0 < H; h0 += 32) {
    const int h = h0 + lr;
    bf16x8 Bf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) Bf[ks] = frag_rows(xb + h * D + 16 * ks + 8 * half, h < H);
    f32x16 acc[NFT];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) acc[ft] = zero16();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) acc[ft] = mfma(A[ft][ks], Bf[ks], acc[ft]);
    if (h < H) {
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int f = 32 * ft + tile_row(i, half);
          if (MASKED) {
            if (f < F && h > f) out[tri_start(f, F) + h - f - 1] = acc[ft][i];
          } else if (f < F) {
            out[f * H + h] = acc[ft][i];
          }
        }
    }
  }
}

// ---------------------------------------------------------------- d_X0 = G Xk,  d_Xk = G^T X0
// NDT = D / 32 column tiles of d.  First d_X0: per 16 values of h the G fragments of the NFT row tiles (two
// 16-byte loads each) and the Xk fragments of the NDT column tiles (eight 4-byte loads each), NFT * NDT
// matrix instructions; the NFT * NDT result tiles leave at the end.  Then d_Xk, 32 rows of h at a time: the X0
// fragments (K = f, ceil(F / 16) steps) stay in registers, the G^T fragments are eight 4-byte loads.
template <int NFT, int NDT, bool MASKED>
__global__ __launch_bounds__(256) void cin_pair_bwd_bf16_kernel(int64_t B, int F, int H, const float* __restrict__ x0,
                                                                const float* __restrict__ xk,
                                                                const float* __restrict__ dS, float* __restrict__ d_x0,
                                                                int x0_mode, float* __restrict__ d_xk, int xk_mode) {
  xdfm::Stamp stamp_(xdfm::kStampCinDx);
  constexpr int D = 32 * NDT;
  constexpr int NKF = 2 * NFT;  // 16-value steps over f (F <= 32 NFT)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, half = lane >> 5;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float* xa = x0 + b * F * D;
  const float* xb = xk + b * H * D;
  const float* g = dS + b * (MASKED ? F * (F - 1) / 2 : F * H);
  {
    f32x16 acc[NFT][NDT];
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) acc[ft][dt] = zero16();
    for (int h0 = 0; h0 < H; h0 += 16) {
      const int hk = h0 + 8 * half;  // this lane's 8 values of h
      bf16x8 Ga[NFT], Xb[NDT];
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) {
        const int f = 32 * ft + lr;
        if (MASKED) {
          // G + G^T out of the triangle: (f, h) and (h, f) are one pair, the diagonal is empty
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int h = hk + i;
            const int lo = f < h ? f : h, hi = f < h ? h : f;
            v[i] = (f < F && h < H && f != h) ? g[tri_start(lo, F) + hi - lo - 1] : 0.f;
          }
          Ga[ft] = frag_of(v);
        } else {
          // (H is a multiple of 8 here: a fragment lies inside [0, H) or outside it)
          Ga[ft] = frag_rows(g + f * H + hk, f < F && hk < H);
        }
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) Xb[dt] = frag_strided(xb + (int64_t)hk * D + 32 * dt + lr, D, H - hk, true);
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) acc[ft][dt] = mfma(Ga[ft], Xb[dt], acc[ft][dt]);
    }
    float* out = d_x0 + b * F * D;
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int f = 32 * ft + tile_row(i, half);
          if (f < F) {
            float* p = out + f * D + 32 * dt + lr;
            *p = x0_mode ? *p + acc[ft][dt][i] : acc[ft][dt][i];
          }
        }
  }
  if (MASKED || d_xk == nullptr) return;
  bf16x8 Xa[NKF][NDT];
#pragma unroll
  for (int ks = 0; ks < NKF; ++ks) {
    const int fk = 16 * ks + 8 * half;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      Xa[ks][dt] = frag_strided(xa + (int64_t)fk * D + 32 * dt + lr, D, F - fk, 16 * ks < F);
  }
  float* out = d_xk + b * H * D;
  for (int h0 = 0; h0 < H; h0 += 32) {
    const int h = h0 + lr;
    f32x16 acc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) acc[dt] = zero16();
#pragma unroll
    for (int ks = 0; ks < NKF; ++ks) {
      if (16 * ks >= F) break;  // (uniform: steps past the last field hold zeros only)
      const int fk = 16 * ks + 8 * half;
      const bf16x8 Gt = frag_strided(g + (int64_t)fk * H + h, H, F - fk, h < H);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) acc[dt] = mfma(Gt, Xa[ks][dt], acc[dt]);
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int hr = h0 + tile_row(i, half);
        if (hr < H) {
          float* p = out + hr * D + 32 * dt + lr;
          *p = xk_mode ? *p + acc[dt][i] : acc[dt][i];
        }
      }
  }
}

bool served(int64_t B, int F, int H, int D, int masked, const float* x0, const float* xk, bool bwd) {
  if (B <= 0 || (masked ? (xk != x0 || F != H) : xk == x0)) return false;
  if (F > 64 || D > 64 || D % (bwd ? 32 : 16) != 0) return false;
  if (bwd && !masked && H % 8 != 0) return false;  // a G fragment is 8 consecutive h of one row, 16-byte aligned
  return true;
}

}  // namespace

extern "C" int xdfm_cin_pair_sum_bf16(const float* x0, const float* xk, int64_t B, int32_t F, int32_t H, int32_t D,
                                      int32_t masked, float* S, void* stream) {
  if (!served(B, F, H, D, masked, x0, xk, false)) return xdfm_cin_pair_sum(x0, xk, B, F, H, D, masked, S, stream);
  XDFM_REQUIRE(x0 && xk && S && F > 0 && H > 0, "xdfm_cin_pair_sum_bf16: bad argument");
  XDFM_REQUIRE(((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(xk)) & 15) == 0,
               "xdfm_cin_pair_sum_bf16: x0 and xk must be 16-byte aligned");
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  const int nft = (F + 31) / 32, nks = D / 16;
#define SUM_CASE(NFT_, NKS_)                                                                                  \
  if (nft == NFT_ && nks == NKS_) {                                                                           \
    if (masked)                                                                                               \
      XDFM_LAUNCH("cin_pair_sum_bf16_kernel<" #NFT_ ", " #NKS_ ", masked>", stream,                           \
                  (cin_pair_sum_bf16_kernel<NFT_, NKS_, true>), grid, block, 0, B, F, H, x0, xk, S);          \
    else                                                                                                      \
      XDFM_LAUNCH("cin_pair_sum_bf16_kernel<" #NFT_ ", " #NKS_ ">", stream,                                   \
                  (cin_pair_sum_bf16_kernel<NFT_, NKS_, false>), grid, block, 0, B, F, H, x0, xk, S);         \
  }
  SUM_CASE(1, 1);
  SUM_CASE(1, 2);
  SUM_CASE(1, 3);
  SUM_CASE(1, 4);
  SUM_CASE(2, 1);
  SUM_CASE(2, 2);
  SUM_CASE(2, 3);
  SUM_CASE(2, 4);
#undef SUM_CASE
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_pair_bwd_bf16(const float* x0, const float* xk, const float* dS, int64_t B, int32_t F,
                                      int32_t H, int32_t D, int32_t masked, float* d_x0, int32_t accumulate_x0,
                                      float* d_xk, int32_t accumulate_xk, void* stream) {
  if (!served(B, F, H, D, masked, x0, xk, true))
    return xdfm_cin_pair_bwd(x0, xk, dS, B, F, H, D, masked, d_x0, accumulate_x0, d_xk, accumulate_xk, stream);
  XDFM_REQUIRE(x0 && xk && dS && d_x0 && F > 0 && H > 0, "xdfm_cin_pair_bwd_bf16: bad argument");
  XDFM_REQUIRE((xk == x0) == (d_xk == nullptr), "xdfm_cin_pair_bwd_bf16: d_xk must be NULL exactly when xk == x0");
  XDFM_REQUIRE((reinterpret_cast<uintptr_t>(dS) & 15) == 0, "xdfm_cin_pair_bwd_bf16: dS must be 16-byte aligned");
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  const int nft = (F + 31) / 32, ndt = D / 32;
#define BWD_CASE(NFT_, NDT_)                                                                                  \
  if (nft == NFT_ && ndt == NDT_) {                                                                           \
    if (masked)                                                                                               \
      XDFM_LAUNCH("cin_pair_bwd_bf16_kernel<" #NFT_ ", " #NDT_ ", masked>", stream,                           \
                  (cin_pair_bwd_bf16_kernel<NFT_, NDT_, true>), grid, block, 0, B, F, H, x0, xk, dS, d_x0,    \
                  accumulate_x0 ? 1 : 0, d_xk, accumulate_xk ? 1 : 0);                                        \
    else                                                                                                      \
      XDFM_LAUNCH("cin_pair_bwd_bf16_kernel<" #NFT_ ", " #NDT_ ">", stream,                                   \
                  (cin_pair_bwd_bf16_kernel<NFT_, NDT_, false>), grid, block, 0, B, F, H, x0, xk, dS, d_x0,   \
                  accumulate_x0 ? 1 : 0, d_xk, accumulate_xk ? 1 : 0);                                        \
  }
  BWD_CASE(1, 1);
  BWD_CASE(1, 2);
  BWD_CASE(2, 1);
  BWD_CASE(2, 2);
#undef BWD_CASE
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}
