// Compressed Interaction Network on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16,
// fp32 accumulation): the wide shapes of BASELINE configs[4] (embedding dim 64, layers of 256
// channels, 128 of them hidden).  Reference: recommenders/models/deeprec/models/xDeepFM.py:191-267.
//
// These kernels serve a layer's HIDDEN channels only (curr[:, :n_hidden] -> the next layer's
// input); the pooled channels go through the pair sums of cin_pool_kernels.hip, which cost 1/D of
// the multiply-adds.  With m = (sample, d) and W[(f,h), n] the filter:
//
//   forward   curr[n,m] = sum_f x0[f,m] * U_f[n,m],   U_f[n,m] = sum_h W[(f,h),n] xk[h,m]
//   d-input   T_f[h,m]  = sum_n W[(f,h),n] dcurr[n,m]
//             d_xk[h,m] = sum_f x0[f,m] T_f[h,m],     d_x0[f,m] = sum_h xk[h,m] T_f[h,m]
//   d-filter  dW[(f,h),n] = sum_m x0[f,m] xk[h,m] dcurr[n,m]
//
// Forward and d-input share one shape: the matrix product runs with the filter as the A operand
// (streamed through LDS from an image packed in operand order) and a register-resident B operand
// whose COLUMN is m, so a lane owns one m: the result tile has its 16 values of one m in the lane's
// registers, and the sum over f is a per-lane multiply-add with x0[f,m] -- no cross-lane traffic,
// no materialised outer product (the reference's Z, :192-200).  The masked layer 0 (:208-217: pairs
// f < h, times 2) runs as the full square with the symmetrised filter W[(min,max)] and a zero
// diagonal, which is the same sum.
//
// Operands are rounded to bf16 (x, W, dcurr); sums and the x0 factors stay fp32.  Parity bar: 2e-2
// relative (BASELINE.json north_star, bf16).  All sums have a fixed order: bit-identical reruns.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cin_bf16.h"
#include "xdfm_common.h"

using namespace xdfm_bf16;

namespace {

// the filter as the kernels read it: symmetrised with a zero diagonal on the masked layer
__device__ __forceinline__ float filter_at(const float* __restrict__ w, int F, int H, int L, int ldw, int masked,
                                           int f, int h, int n) {
  if (f >= F || h >= H || n >= L) return 0.f;
  if (masked) {
    if (f == h) return 0.f;
    const int lo = f < h ? f : h, hi = f < h ? h : f;
    return w[(int64_t)(lo * H + hi) * ldw + n];
  }
  return w[(int64_t)(f * H + h) * ldw + n];
}

// forward image  [nt][f][ks][lane][8]: A[row n = 32nt + (lane&31)][k: h = 16ks + 8(lane>>5) + j]
// d-input image  [f][ht][ks][lane][8]: A[row h = 32ht + (lane&31)][k: n = 16ks + 8(lane>>5) + j]
__global__ void cin_bf16_pack_kernel(const float* __restrict__ w, int F, int H, int L, int ldw, int masked, int Fp,
                                     int nks_h, int nt_n, int nks_n, int nt_h, uint32_t* __restrict__ fwd_img,
                                     uint32_t* __restrict__ dx_img) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one packed pair each
  const int64_t n_fwd = (int64_t)nt_n * Fp * nks_h * 64 * 4;
  const int64_t n_dx = (int64_t)Fp * nt_h * nks_n * 64 * 4;
  if (idx < n_fwd && fwd_img) {
    const int jp = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    int64_t t = idx >> 8;
    const int ks = (int)(t % nks_h);
    t /= nks_h;
    const int f = (int)(t % Fp), nt = (int)(t / Fp);
    const int n = 32 * nt + (lane & 31), h = 16 * ks + 8 * (lane >> 5) + 2 * jp;
    fwd_img[idx] = pack_bf16(filter_at(w, F, H, L, ldw, masked, f, h, n), filter_at(w, F, H, L, ldw, masked, f, h + 1, n));
  }
  if (idx < n_dx && dx_img) {
    const int jp = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    int64_t t = idx >> 8;
    const int ks = (int)(t % nks_n);
    t /= nks_n;
    const int ht = (int)(t % nt_h), f = (int)(t / nt_h);
    const int h = 32 * ht + (lane & 31), n = 16 * ks + 8 * (lane >> 5) + 2 * jp;
    dx_img[idx] = pack_bf16(filter_at(w, F, H, L, ldw, masked, f, h, n), filter_at(w, F, H, L, ldw, masked, f, h, n + 1));
  }
}

// B-operand fragments of src[b][R][D] for this lane's column m: element j of step ks is row
// 16ks + 8(lane>>5) + j.  Lanes of a half read consecutive d: coalesced.
template <int NKS>
__device__ __forceinline__ void load_frags(const float* __restrict__ src, int R, int D, int64_t b, int d, bool live,
                                           int half, bf16x8 (&frag)[NKS]) {
  // (b, d) of a lane without a column are (0, 0): every address below is valid, so the loads carry no
  // branch and no select until a batch of them (4 k steps = 32 loads) is in flight
  const float* base = src + b * R * D + d;
#pragma unroll
  for (int ks0 = 0; ks0 < NKS; ks0 += 4) {
    float v[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ks0 + q < NKS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 16 * (ks0 + q) + 8 * half + j;
          v[q][j] = base[(r < R ? r : R - 1) * D];
        }
      }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ks0 + q < NKS) {
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (live && 16 * (ks0 + q) + 8 * half + j < R) ? v[q][j] : 0.f;
        uint4 p;
        p.x = pack_bf16(w[0], w[1]);
        p.y = pack_bf16(w[2], w[3]);
        p.z = pack_bf16(w[4], w[5]);
        p.w = pack_bf16(w[6], w[7]);
        frag[ks0 + q] = as_frag(p);
      }
  }
}

// x0[f][m] of the block's m range as fp32 in LDS: the factors of the sum over f.  Eight loads in flight
// per thread (clamped addresses, masks applied when stored).
__device__ __forceinline__ void stage_x0(const Dims& c, const float* __restrict__ x0, int64_t m0, float* __restrict__ x0s) {
  const int ml = threadIdx.x & (kMB - 1);  // kMB == blockDim.x: a thread keeps its column, f advances
  const int64_t m = m0 + ml;
  const bool ok = m < c.cols;
  const int64_t b = ok ? m / c.D : 0;
  const float* base = x0 + b * c.F * c.D + (ok ? (int)(m - b * c.D) : 0);
  for (int f0 = 0; f0 < c.Fp; f0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f0 + u;
      v[u] = base[(f < c.F ? f : c.F - 1) * c.D];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f0 + u;
      if (f < c.Fp) x0s[f * kMB + ml] = (ok && f < c.F) ? v[u] : 0.f;
    }
  }
}

// ---------------------------------------------------------------- forward
// Chunk = the images of G consecutive f for one channel tile (G * NKS KB): staged global ->
// registers -> LDS one chunk ahead of its use, one barrier per chunk.
template <int NKS>
__global__ __launch_bounds__(256, 2) void cin_bf16_fwd_kernel(Dims c, const float* __restrict__ x0,
                                                               const float* __restrict__ xk,
                                                               const uint4* __restrict__ img, float* __restrict__ xnext) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* x0s = reinterpret_cast<float*>(smem);                                // [Fp][kMB]
  uint4* wbuf = reinterpret_cast<uint4*>(smem + (size_t)c.Fp * kMB * 4);     // [2][chunk]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int chunk = c.G * NKS * 64;  // uint4 per chunk
  const int64_t m0 = (int64_t)blockIdx.x * kMB;
  const int NT = (c.L + 31) >> 5, OG = c.Fp / c.G;
  const int nchunks = NT * OG;

  constexpr int kFetch = NKS == 8 ? 4 : 2;  // uint4 per thread and chunk (make_dims: 16 KB chunks for NKS 8, else 8 KB)
  auto fetch = [&](int ci, uint4 (&r)[kFetch]) {
    const uint4* src = img + (size_t)(ci < nchunks ? ci : nchunks - 1) * chunk;
#pragma unroll
    for (int u = 0; u < kFetch; ++u) {
      const int i = threadIdx.x + u * 256;
      r[u] = i < chunk ? src[i] : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int buf, const uint4 (&r)[kFetch]) {
#pragma unroll
    for (int u = 0; u < kFetch; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < chunk) wbuf[buf * chunk + i] = r[u];
    }
  };

  uint4 pre[kFetch];
  fetch(0, pre);
  stage_x0(c, x0, m0, x0s);
  int64_t mb[kMT];
  int md[kMT], mloc[kMT];
  bool live[kMT];
  bf16x8 xf[kMT][NKS];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt) {
    mloc[mt] = (wave * kMT + mt) * 32 + (lane & 31);
    const int64_t m = m0 + mloc[mt];
    live[mt] = m < c.cols;
    mb[mt] = live[mt] ? m / c.D : 0;
    md[mt] = live[mt] ? (int)(m - mb[mt] * c.D) : 0;
    load_frags<NKS>(xk, c.H, c.D, mb[mt], md[mt], live[mt], half, xf[mt]);
  }
  stash(0, pre);
  __syncthreads();

  f32x16 acc[kMT];
  for (int ci = 0; ci < nchunks; ++ci) {
    const int nt = ci / OG, og = ci - nt * OG;
    fetch(ci + 1, pre);
    if (og == 0) {
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    }
    const uint4* wb = wbuf + (ci & 1) * chunk + lane;
    for (int g = 0; g < c.G; ++g) {
      const int f = og * c.G + g;
      f32x16 U[kMT];
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) U[mt][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8 a = as_frag(wb[(g * NKS + ks) * 64]);
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) U[mt] = mfma(a, xf[mt][ks], U[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) {
        const float xo = x0s[f * kMB + mloc[mt]];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = fmaf(xo, U[mt][i], acc[mt][i]);
      }
    }
    if (og == OG - 1) {
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) {
        if (!live[mt]) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = 32 * nt + tile_row(i, half);
          if (n < c.L) xnext[(mb[mt] * c.L + n) * c.D + md[mt]] = acc[mt][i];
        }
      }
    }
    stash((ci + 1) & 1, pre);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- input gradients
// B operand: the upstream gradient dcurr[n][m] (d_xnext[b][n][d]), K = the channels.  One pass per
// 32-row h tile; inside a pass one chunk per f (the image rows of (f, h tile), NKS KB):
//   T[h,m] = sum_n W[(f,h),n] dcurr[n,m]          matrix cores
//   d_xk[h,m] += x0[f,m] T[h,m]                   per lane, 16 rows of the tile in registers
//   d_x0[f,m] += sum_h xk[h,m] T[h,m]             per lane, then the two lane halves are added
// The masked layer (xk == x0, symmetrised filter) needs the second sum only, times 2.
template <int NKS>
__global__ __launch_bounds__(256, 2) void cin_bf16_dx_kernel(Dims c, const float* __restrict__ x0,
                                                              const float* __restrict__ xk,
                                                              const uint4* __restrict__ img,
                                                              const float* __restrict__ d_xnext, float* __restrict__ d_xk,
                                                              float* __restrict__ d_x0, int accumulate_x0) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* x0s = reinterpret_cast<float*>(smem);                             // [Fp][kMB]
  uint4* wbuf = reinterpret_cast<uint4*>(smem + (size_t)c.Fp * kMB * 4);  // [2][chunk]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  constexpr int chunk = NKS * 64;  // uint4 per chunk: one (f, h tile)
  const int64_t m0 = (int64_t)blockIdx.x * kMB;
  const int nht = (c.H + 31) >> 5;
  const int nchunks = nht * c.F;  // pass-major: chunk ci = (ht = ci / F, f = ci % F)

  auto fetch = [&](int ci, uint4 (&r)[(NKS + 3) / 4]) {
    if (ci >= nchunks) ci = nchunks - 1;
    const int ht = ci / c.F, f = ci - ht * c.F;
    const uint4* src = img + (size_t)(f * nht + ht) * chunk;
#pragma unroll
    for (int u = 0; u < (NKS + 3) / 4; ++u) {
      const int i = threadIdx.x + u * 256;
      r[u] = (NKS % 4 == 0 || i < chunk) ? src[i] : make_uint4(0, 0, 0, 0);  // whole rounds: no mask
    }
  };
  auto stash = [&](int buf, const uint4 (&r)[(NKS + 3) / 4]) {
#pragma unroll
    for (int u = 0; u < (NKS + 3) / 4; ++u) {
      const int i = threadIdx.x + u * 256;
      if (NKS % 4 == 0 || i < chunk) wbuf[buf * chunk + i] = r[u];
    }
  };

  uint4 pre[(NKS + 3) / 4];
  fetch(0, pre);
  stage_x0(c, x0, m0, x0s);
  int64_t mb[kMT];
  int md[kMT], mloc[kMT];
  bool live[kMT];
  bf16x8 df[kMT][NKS];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt) {
    mloc[mt] = (wave * kMT + mt) * 32 + (lane & 31);
    const int64_t m = m0 + mloc[mt];
    live[mt] = m < c.cols;
    mb[mt] = live[mt] ? m / c.D : 0;
    md[mt] = live[mt] ? (int)(m - mb[mt] * c.D) : 0;
    load_frags<NKS>(d_xnext, c.L, c.D, mb[mt], md[mt], live[mt], half, df[mt]);
  }
  stash(0, pre);
  __syncthreads();

  float* col_x0[kMT];  // d_x0[b][0][d] of this lane's column: row f is f * D further on
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt) col_x0[mt] = d_x0 + mb[mt] * c.F * c.D + md[mt];
  const bool want_xk = d_xk != nullptr;
  const float pscale = c.masked ? 2.f : 1.f;
  f32x16 dxk[kMT], xkv[kMT];
  for (int ci = 0; ci < nchunks; ++ci) {
    const int ht = ci / c.F, f = ci - ht * c.F;
    fetch(ci + 1, pre);
    if (f == 0) {  // a new pass: this tile's xk rows, cleared sums
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int h = 32 * ht + tile_row(i, half);
          xkv[mt][i] = xk[(mb[mt] * c.H + (h < c.H ? h : c.H - 1)) * c.D + md[mt]];  // masked below
          dxk[mt][i] = 0.f;
        }
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (!(live[mt] && 32 * ht + tile_row(i, half) < c.H)) xkv[mt][i] = 0.f;
    }
    const uint4* wb = wbuf + (ci & 1) * chunk + lane;
    // the d_x0 values this chunk adds to are requested now and used after the products
    float prev[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
      prev[mt] = (half == 0 && live[mt] && (accumulate_x0 || ht > 0)) ? col_x0[mt][(int64_t)f * c.D] : 0.f;
    f32x16 T[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) T[mt][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const bf16x8 a = as_frag(wb[ks * 64]);
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) T[mt] = mfma(a, df[mt][ks], T[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      // two running sums (packed multiply-adds), added at the end: a fixed order
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 pp = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 16; i += 2)
        pp = __builtin_elementwise_fma(f32x2{xkv[mt][i], xkv[mt][i + 1]}, f32x2{T[mt][i], T[mt][i + 1]}, pp);
      float p = pp.x + pp.y;
      if (want_xk) {
        const float xo = x0s[f * kMB + mloc[mt]];
#pragma unroll
        for (int i = 0; i < 16; ++i) dxk[mt][i] = fmaf(xo, T[mt][i], dxk[mt][i]);
      }
      // the other half's sum by a register swap between the halves (no LDS round trip)
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
      p += __uint_as_float(half == 0 ? sw[1] : sw[0]);
      if (half == 0 && live[mt]) col_x0[mt][(int64_t)f * c.D] = prev[mt] + pscale * p;
    }
    if (f == c.F - 1 && want_xk) {
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) {
        if (!live[mt]) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int h = 32 * ht + tile_row(i, half);
          if (h < c.H) d_xk[(mb[mt] * c.H + h) * c.D + md[mt]] = dxk[mt][i];
        }
      }
    }
    stash((ci + 1) & 1, pre);
    __syncthreads();
  }
}

int nks_for(int K) {  // k steps (of 16) the kernels are instantiated for
  const int n = (K + 15) / 16;
  return n <= 2 ? 2 : (n <= 3 ? 3 : (n <= 4 ? 4 : (n <= 6 ? 6 : (n <= 8 ? 8 : 0))));
}

}  // namespace

extern "C" int xdfm_cin_bf16_image_bytes(int32_t F, int32_t H, int32_t L, size_t* fwd_bytes, size_t* dx_bytes) {
  XDFM_REQUIRE(fwd_bytes && dx_bytes && F > 0 && H > 0 && L > 0, "xdfm_cin_bf16_image_bytes: bad argument");
  const int nks_h = nks_for(H), nks_n = nks_for(L);
  XDFM_REQUIRE(nks_h && nks_n, "xdfm_cin_bf16: at most 128 inputs per side and 128 channels per call (H=%d, L=%d)", H, L);
  Dims cf, cd;
  int rc = make_dims(1, F, H, 1, L, 0, nks_h, &cf);
  if (rc) return rc;
  rc = make_dims(1, F, H, 1, L, 0, nks_n, &cd);
  if (rc) return rc;
  *fwd_bytes = (size_t)((L + 31) / 32) * cf.Fp * nks_h * 1024;
  *dx_bytes = (size_t)cd.Fp * ((H + 31) / 32) * nks_n * 1024;
  return XDFM_OK;
}

extern "C" int xdfm_cin_bf16_pack(const float* w, int32_t F, int32_t H, int32_t L, int32_t ldw, int32_t masked,
                                  void* fwd_image, void* dx_image, void* stream) {
  XDFM_REQUIRE(w && (fwd_image || dx_image), "xdfm_cin_bf16_pack: null argument");
  const int nks_h = nks_for(H), nks_n = nks_for(L);
  XDFM_REQUIRE(nks_h && nks_n, "xdfm_cin_bf16: at most 128 inputs per side and 128 channels per call (H=%d, L=%d)", H, L);
  Dims cf, cd;
  int rc = make_dims(1, F, H, 1, L, masked, nks_h, &cf);
  if (rc) return rc;
  rc = make_dims(1, F, H, 1, L, masked, nks_n, &cd);
  if (rc) return rc;
  const int nt_n = (L + 31) / 32, nt_h = (H + 31) / 32;
  // one launch per image: their f padding may differ
  if (fwd_image) {
    const int64_t n = (int64_t)nt_n * cf.Fp * nks_h * 256;
    XDFM_LAUNCH("cin_bf16_pack_kernel", stream, cin_bf16_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w,
                F, H, L, ldw ? ldw : L, masked ? 1 : 0, cf.Fp, nks_h, nt_n, nks_n, nt_h, (uint32_t*)fwd_image,
                (uint32_t*)nullptr);
  }
  if (dx_image) {
    const int64_t n = (int64_t)cd.Fp * nt_h * nks_n * 256;
    XDFM_LAUNCH("cin_bf16_pack_kernel", stream, cin_bf16_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w,
                F, H, L, ldw ? ldw : L, masked ? 1 : 0, cd.Fp, nks_h, nt_n, nks_n, nt_h, (uint32_t*)nullptr,
                (uint32_t*)dx_image);
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_bf16_fwd(const float* x0, const float* xk, const void* fwd_image, int64_t B, int32_t F,
                                 int32_t H, int32_t D, int32_t L, int32_t masked, float* xnext, void* stream) {
  const int nks = nks_for(H);
  XDFM_REQUIRE(nks && nks_for(L), "xdfm_cin_bf16: at most 128 inputs per side and 128 channels per call (H=%d, L=%d)", H, L);
  Dims c;
  int rc = make_dims(B, F, H, D, L, masked, nks, &c);
  if (rc) return rc;
  if (B == 0) return XDFM_OK;
  XDFM_REQUIRE(x0 && xk && fwd_image && xnext, "xdfm_cin_bf16_fwd: null argument");
  const size_t lds = (size_t)c.Fp * kMB * 4 + (size_t)2 * c.G * nks * 1024;
  XDFM_REQUIRE(lds <= 160 * 1024, "xdfm_cin_bf16_fwd: %d fields do not fit the LDS", F);
  const dim3 grid((unsigned)((c.cols + kMB - 1) / kMB));
#define FWD(N)                                                                                                       \
  do {                                                                                                               \
    if (lds > 64 * 1024)                                                                                             \
      XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bf16_fwd_kernel<N>),                            \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
    XDFM_LAUNCH("cin_bf16_fwd_kernel<" #N ">", stream, cin_bf16_fwd_kernel<N>, grid, dim3(256), lds, c, x0, xk,      \
                (const uint4*)fwd_image, xnext);                                                                     \
  } while (0)
  switch (nks) {
    case 2: FWD(2); break;
    case 3: FWD(3); break;
    case 4: FWD(4); break;
    case 6: FWD(6); break;
    default: FWD(8); break;
  }
#undef FWD
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_bf16_bwd_dx(const float* x0, const float* xk, const void* dx_image, const float* d_xnext,
                                    int64_t B, int32_t F, int32_t H, int32_t D, int32_t L, int32_t masked,
                                    float* d_xk, float* d_x0, int32_t accumulate_x0, void* stream) {
  const int nks = nks_for(L);
  XDFM_REQUIRE(nks && nks_for(H), "xdfm_cin_bf16: at most 128 inputs per side and 128 channels per call (H=%d, L=%d)", H, L);
  Dims c;
  int rc = make_dims(B, F, H, D, L, masked, nks, &c);
  if (rc) return rc;
  if (B == 0) return XDFM_OK;
  XDFM_REQUIRE(x0 && xk && dx_image && d_xnext && d_x0, "xdfm_cin_bf16_bwd_dx: null argument");
  XDFM_REQUIRE(masked ? d_xk == nullptr : d_xk != nullptr,
               "xdfm_cin_bf16_bwd_dx: d_xk is for the layers after the masked one only");
  const size_t lds = (size_t)c.Fp * kMB * 4 + (size_t)2 * nks * 1024;
  XDFM_REQUIRE(lds <= 160 * 1024, "xdfm_cin_bf16_bwd_dx: %d fields do not fit the LDS", F);
  const dim3 grid((unsigned)((c.cols + kMB - 1) / kMB));
#define DX(N)                                                                                                      \
  do {                                                                                                             \
    if (lds > 64 * 1024)                                                                                           \
      XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bf16_dx_kernel<N>),                           \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
    XDFM_LAUNCH("cin_bf16_dx_kernel<" #N ">", stream, cin_bf16_dx_kernel<N>, grid, dim3(256), lds, c, x0, xk,      \
                (const uint4*)dx_image, d_xnext, d_xk, d_x0, accumulate_x0);                                       \
  } while (0)
  switch (nks) {
    case 2: DX(2); break;
    case 3: DX(3); break;
    case 4: DX(4); break;
    case 6: DX(6); break;
    default: DX(8); break;
  }
#undef DX
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

