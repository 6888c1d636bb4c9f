// Compressed Interaction Network on the fp32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32: exact fp32, bitwise a k-ordered fmaf chain).
//
// Reference: recommenders/models/deeprec/models/xDeepFM.py:191-267.  The reference
// materialises Z[b,d,f*H+h] = X0[b,f,d]*Xk[b,h,d] (:192-200), masks layer 0 (:208-217) and
// runs a width-1 conv1d = GEMM over the pair index (:220-222).  Z is never stored here.
// With m = (sample, d) the three contractions and the form each kernel uses:
//
//   forward   curr[m,n] = sum_o Xo[m,o] * (sum_i Xi[m,i] W[(o,i),n])   inner sum on the matrix
//             cores (K = the longer feature axis), outer sum as 4 FMAs per 4x4 (o, n) tile
//   d-input   T[(f,h),m] = sum_n W[(f,h),n] dcurr[m,n]                 matrix cores, K = channels
//             dXk[m,h] = sum_f X0[m,f] T ,  dX0[m,f] = sum_h Xk[m,h] T   2 FMAs per product
//   d-filter  dW[p,n] = sum_m Z[m,p] dcurr[m,n]     matrix cores, K = rows, Z = one v_mul per
//             operand from LDS-resident X tiles
//
// Pairs of the masked layer 0 (f >= h) are skipped at tile granularity.  All sums have a
// fixed association (no float atomics): results are bit-identical from run to run.
// On gfx950 fp32 MFMAs and fp32 VALU ops share the SIMD's lanes (tools/ub_mfma_mix.hip), so
// every kernel is laid out to issue as little as possible besides its MFMAs: operands come
// from LDS with wide reads at loop-invariant addresses, filters from images packed in MFMA
// operand order.
//
// Layouts: x0[B][F][D], xk[B][H][D], w[F*H][L], xnext[B][n_hidden][D], pooled[B][stride].
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "cin_common.h"
#include "cin_stage.h"
#include "xdfm_common.h"
#include "stamp.h"

namespace xdfm_cin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {


__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------ forward, factorised
//   curr[m,n] = scale * sum_o xo[m,o] * U[m,(o,n)],    U[m,(o,n)] = sum_i xi[m,i] * W'[i,(o,n)]
// (xi, xo) = (xk, x0) or, when F > H, (x0, xk): the longer feature axis is the K of the matrix
// product, the shorter one is folded in with 4 FMAs per tile.  On gfx950 fp32 MFMAs and VALU
// ops share the SIMD's fp32 lanes (tools/ub_mfma_mix.hip: every extra VALU op per MFMA costs
// ~3 cycles on top of the MFMA's 32), so the loop is built to issue almost nothing else:
//   A[i][k] = W'[4ks+k][(4oq + i>>2, 4nq + i&3)]   packed image wq, one float4 per 4 steps
//   B[k][j] = xi[row][4ks + k]                     LDS [feature][row]; lane j owns rows 4j..4j+3
//                                                  of its wave (4 row tiles), one ds_read_b128
//   D[i = 4*oi + r][j]: lane (oi = lane>>4, j) holds channels 4nq..4nq+3 of its 4 rows for
//   outer feature 4oq+oi and folds them into its accumulator with xo[row][4oq+oi]; the 4 oi
//   groups are added once per nq.
// Steps (oq, ks) of one nq form a flat list (descriptors in global memory, read through the
// scalar cache); the masked layer skips ks < oq (h <= f there).
__host__ __device__ inline int fwd2_steps(int noq, int nks, int masked) {
  return masked ? noq * nks - noq * (noq - 1) / 2 : noq * nks;
}
constexpr int kFwdLast = 1 << 16, kFwdFirst = 1 << 17, kFwdValid = 1 << 24;
// step s -> ks | oq << 8 | flags (0 for a padding step)
__device__ inline int fwd2_step_desc(int s, int noq, int nks, int masked) {
  int oq = 0, rem = s;
  while (oq < noq) {
    const int cnt = masked ? nks - oq : nks;
    if (rem < cnt) break;
    rem -= cnt;
    ++oq;
  }
  if (oq >= noq) return 0;
  const int ks0 = masked ? oq : 0;
  const int ks = ks0 + rem;
  return ks | (oq << 8) | ((ks == nks - 1) ? kFwdLast : 0) | ((ks == ks0) ? kFwdFirst : 0) | kFwdValid;
}

// image wq[nq][group][lane][4 steps] followed by the step descriptors stab[nsteps_img]
__device__ __forceinline__ void pack_fwd_body(const CinDims& c, const float* __restrict__ w, int noq, int nks,
                                              int nsteps_img, int NQ, int swap, float* __restrict__ wq,
                                              int64_t idx) {
  const int64_t total = (int64_t)NQ * nsteps_img * 64;
  if (idx < nsteps_img)
    reinterpret_cast<int*>(wq + total)[idx] = fwd2_step_desc((int)idx, noq, nks, c.masked);
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int s = (int)((idx >> 6) % nsteps_img);
  const int nq = (int)((idx >> 6) / nsteps_img);
  const int desc = fwd2_step_desc(s, noq, nks, c.masked);
  const int ks = desc & 0xff, oq = (desc >> 8) & 0xff;
  const int i = lane & 15, k = lane >> 4;
  const int o = 4 * oq + (i >> 2), n = 4 * nq + (i & 3), in = 4 * ks + k;
  const int f = swap ? in : o, h = swap ? o : in;
  const bool ok = (desc & kFwdValid) && f < c.F && h < c.H && n < c.L && (!c.masked || h > f);
  const float v = ok ? w[(int64_t)(f * c.H + h) * cin_ldw(c) + n] : 0.f;
  wq[(((int64_t)nq * (nsteps_img >> 2) + (s >> 2)) * 64 + lane) * 4 + (s & 3)] = v;
}

__global__ void cin_pack_fwd_kernel(CinDims c, const float* __restrict__ w, int noq, int nks,
                                    int nsteps_img, int NQ, int swap, float* __restrict__ wq) {
  pack_fwd_body(c, w, noq, nks, nsteps_img, NQ, swap, wq, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// xi: K-side input [B][Ki][D], xo: folded input [B][Ko][D] (xi == xo for the masked layer)
__global__ __launch_bounds__(256) void cin_fwd2_mfma_kernel(
    CinDims c, const float* __restrict__ xi, const float* __restrict__ xo, int Ki, int Ko,
    const float* __restrict__ wq, const int* __restrict__ stab, float* __restrict__ xnext,
    float* __restrict__ pooled, int pool_stride, int pool_off, int SPB, int ngrp, int NQ) {
  xdfm::Stamp stamp_(xdfm::kStampCinFwd);
  extern __shared__ float lds[];
  constexpr int RT = 4;
  constexpr int ROWS = 64 * RT;
  constexpr int Tp = ROWS + kPad;
  const bool shared_x = (xi == xo);
  const int Kip = (Ki + 3) & ~3, Kop = (Ko + 3) & ~3;
  const int npool = c.L - c.n_hidden;
  float* xs = lds;                             // [Kip][Tp]
  float* as = shared_x ? xs : xs + Kip * Tp;   // [Kop][Tp]
  float* outs = as + Kop * Tp;                 // [npool][Tp]  pooled channels only

  const int64_t b0 = (int64_t)blockIdx.x * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  stage_cols(xi, b0, nb, Ki, Kip, c.D, xs, Tp);
  if (!shared_x) stage_cols(xo, b0, nb, Ko, Kop, c.D, as, Tp);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int cols = nb * c.D;
  const int m0 = wave * 64 + 4 * j;  // this lane's 4 rows: m0 .. m0+3
  const float* xlane = xs + g * Tp + m0;
  const float* alane = as + g * Tp + m0;
  const float scale = c.masked ? 2.f : 1.f;  // exact, so applying it last changes no bit
  const int4* stab4 = reinterpret_cast<const int4*>(stab);
  const bool vec_store = (c.D & 3) == 0 && (reinterpret_cast<uintptr_t>(xnext) & 15) == 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  struct Grp {
    f32x4 b[4];
    f32x4 xv[4];  // xo values of the steps that close a tile (stale otherwise, never used)
    int d[4];
  };
  auto issue = [&](Grp& G, int4 dd) {
    G.d[0] = dd.x;
    G.d[1] = dd.y;
    G.d[2] = dd.z;
    G.d[3] = dd.w;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      G.b[s] = *reinterpret_cast<const f32x4*>(xlane + (G.d[s] & 0xff) * (4 * Tp));
      if (G.d[s] & kFwdLast)
        G.xv[s] = *reinterpret_cast<const f32x4*>(alane + ((G.d[s] >> 8) & 0xff) * (4 * Tp));
    }
  };

  float* xn = xnext + b0 * c.n_hidden * c.D;
  // The filter image is one stream over (nq, group): a ring of four float4 registers keeps
  // the loads three groups (48 MFMAs) ahead of their use, across nq boundaries as well.
  const int ngrp4 = (ngrp + 3) & ~3;  // groups per nq in the image (padding groups are zero)
  const f32x4* img = reinterpret_cast<const f32x4*>(wq) + lane;
  const int glast = NQ * ngrp4 - 1;
  auto aload = [&](int gidx) { return img[(size_t)(gidx < glast ? gidx : glast) * 64]; };
  f32x4 A0 = aload(0), A1 = aload(1), A2 = aload(2), A3;
  for (int nq = 0; nq < NQ; ++nq) {
    f32x4 oacc[RT], U[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      oacc[rt] = zero4;
      U[rt] = zero4;
    }
    auto compute = [&](const Grp& G, const f32x4& A) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int d = G.d[s];  // wave-uniform (scalar loads)
        if (d & kFwdFirst) {  // a new (oq, nq) tile: start from zero without clearing registers
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) U[rt] = mfma(A[s], G.b[s][rt], zero4);
        } else {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) U[rt] = mfma(A[s], G.b[s][rt], U[rt]);
        }
        if (d & kFwdLast) {  // the tile is complete: fold it in with this lane's xo values
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[rt][r] = fmaf(G.xv[s][rt], U[rt][r], oacc[rt][r]);
        }
      }
    };
    const int gbase = nq * ngrp4;
    auto desc = [&](int q) { return stab4[q < ngrp ? q : ngrp - 1]; };
    Grp g0, g1;
#pragma unroll
    for (int s = 0; s < 4; ++s) g0.xv[s] = g1.xv[s] = zero4;
    issue(g0, desc(0));
    int4 d_next = desc(1);
    for (int q = 0; q < ngrp4; q += 4) {
      // groups past ngrp exist only in the image (zeros); they are loaded, never multiplied
      const int4 d_a = desc(q + 2);
      A3 = aload(gbase + q + 3);
      issue(g1, d_next);
      __builtin_amdgcn_sched_barrier(0);
      compute(g0, A0);
      __builtin_amdgcn_sched_barrier(0);
      const int4 d_b = desc(q + 3);
      A0 = aload(gbase + q + 4);
      issue(g0, d_a);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 1 < ngrp) compute(g1, A1);
      __builtin_amdgcn_sched_barrier(0);
      const int4 d_c = desc(q + 4);
      A1 = aload(gbase + q + 5);
      issue(g1, d_b);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 2 < ngrp) compute(g0, A2);
      __builtin_amdgcn_sched_barrier(0);
      const int4 d_d = desc(q + 5);
      A2 = aload(gbase + q + 6);
      issue(g0, d_c);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 3 < ngrp) compute(g1, A3);
      __builtin_amdgcn_sched_barrier(0);
      d_next = d_d;
    }
    // add the 4 oi groups (fixed order), then group g owns channel 4*nq + g of rows m0..m0+3:
    // hidden channels go straight to xnext[b][ch][d], pooled ones wait in LDS for the sum over d
    const int ch = 4 * nq + g;
    f32x4 o;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 v = oacc[rt];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = v[r];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        v[r] = t;
      }
      o[rt] = (g == 0 ? v[0] : (g == 1 ? v[1] : (g == 2 ? v[2] : v[3]))) * scale;
    }
    if (ch < c.n_hidden) {
      if (vec_store) {  // 4 | D: the 4 rows are 4 consecutive d of one sample
        if (m0 < cols) {
          const int bl = m0 / c.D, dd = m0 - bl * c.D;
          *reinterpret_cast<f32x4*>(xn + (bl * c.n_hidden + ch) * c.D + dd) = o;
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int m = m0 + rt;
          if (m < cols) {
            const int bl = m / c.D, dd = m - bl * c.D;
            xn[(bl * c.n_hidden + ch) * c.D + dd] = o[rt];
          }
        }
      }
    } else if (ch < c.L) {
      *reinterpret_cast<f32x4*>(outs + (ch - c.n_hidden) * Tp + m0) = o;
    }
  }
  if (npool > 0) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < npool * nb; idx += blockDim.x) {
      const int p = idx / nb;
      const int bl = idx - p * nb;
      float s = 0.f;
      for (int d = 0; d < c.D; ++d) s += outs[p * Tp + bl * c.D + d];  // reduce_sum over D (:270)
      pooled[(b0 + bl) * pool_stride + pool_off + p] = s;
    }
  }
}

// ------------------------------------------------------------ forward, masked layer, static form
// The same contraction, image and operand layout as cin_fwd2_mfma_kernel for the layer whose two inputs are
// one tensor (x0 (x) x0, pairs h > f), with the step list known at compile time (NKS = ceil(F / 4) steps of
// 4 features; tile oq runs ks = oq .. NKS-1):
//   * a lane's B operands are the same 4 NKS values for every tile and channel group (x[4ks + g][m0 .. m0+3]),
//     and the value that folds tile oq in is the one of step ks = oq: they are read ONCE, straight from
//     the tensor into registers (one 16-byte load per step: D % 4 == 0), and the loop touches no LDS,
//     reads no descriptors and computes no addresses
//   * the filter image of a channel group (NKS (NKS+1) / 2 steps, one float4 per 4 steps) is requested a
//     whole group ahead: each register is refilled for the next group behind the last step that reads it
//   * a tile's fold (16 FMAs) is issued behind the first matrix instructions of the next tile
// What the generic kernel spends between its matrix instructions (branches on step flags, scalar descriptor
// loads and the full lgkmcnt wait each phase ends with, LDS reads at computed addresses: 43 % matrix-pipe
// busy) is gone; sums keep their order, so the two kernels agree bit for bit.
template <int NKS>
__global__ __launch_bounds__(256, 2) void cin_fwd_static_kernel(
    CinDims c, const float* __restrict__ x, const float* __restrict__ wq, float* __restrict__ xnext,
    float* __restrict__ pooled, int pool_stride, int pool_off, int SPB, int ngrp4, int NQ) {
  xdfm::Stamp stamp_(xdfm::kStampCinFwd);
  extern __shared__ float lds[];
  constexpr int RT = 4;
  constexpr int ROWS = 64 * RT;
  constexpr int Tp = ROWS + kPad;
  constexpr int NS = NKS * (NKS + 1) / 2;
  constexpr int NG = (NS + 3) / 4;
  float* outs = lds;  // [npool][Tp]  pooled channels only
  const int npool = c.L - c.n_hidden;
  const int64_t b0 = (int64_t)blockIdx.x * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int cols = nb * c.D;
  const int m0 = wave * 64 + 4 * j;  // this lane's 4 rows: m0 .. m0+3, 4 consecutive d of one sample
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool on = m0 < cols;
  const int bl = on ? m0 / c.D : 0, dd = on ? m0 - bl * c.D : 0;

  f32x4 bx[NKS];
  {
    const float* xrow = x + ((b0 + bl) * c.F) * c.D + dd;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int f = 4 * ks + g;
      bx[ks] = (on && f < c.F) ? *reinterpret_cast<const f32x4*>(xrow + (int64_t)f * c.D) : zero4;
    }
  }
  const f32x4* img = reinterpret_cast<const f32x4*>(wq) + lane;
  float* xn = xnext + b0 * c.n_hidden * c.D;

  f32x4 A[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) A[gi] = img[gi * 64];
  for (int nq = 0; nq < NQ; ++nq) {
    // the next group's image (the last group re-reads its own: no branch in the loop)
    const f32x4* nxt = img + (size_t)(nq + 1 < NQ ? nq + 1 : nq) * ngrp4 * 64;
    f32x4 U[2][RT];
    f32x2 oacc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) oacc[rt][0] = oacc[rt][1] = f32x2{0.f, 0.f};
    // two packed FMAs per row tile (the pairs follow the accumulator's registers: no moves)
    auto fold = [&](int oq) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x2 xx = {bx[oq][rt], bx[oq][rt]};
        const f32x4 u = U[oq & 1][rt];
        oacc[rt][0] = __builtin_elementwise_fma(xx, f32x2{u[0], u[1]}, oacc[rt][0]);
        oacc[rt][1] = __builtin_elementwise_fma(xx, f32x2{u[2], u[3]}, oacc[rt][1]);
      }
    };
#pragma unroll
    for (int oq = 0; oq < NKS; ++oq) {
#pragma unroll
      for (int ks = oq; ks < NKS; ++ks) {
        const int s = oq * NKS - oq * (oq - 1) / 2 + (ks - oq);
        const float a = A[s >> 2][s & 3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          U[oq & 1][rt] = mfma(a, bx[ks][rt], ks == oq ? zero4 : U[oq & 1][rt]);
        if ((s & 3) == 3 || s == NS - 1) A[s >> 2] = nxt[(s >> 2) * 64];  // its four steps are done
        // the previous tile is folded in behind the second step of this one (8 matrix instructions after
        // its last: the results are there, nothing waits)
        if (oq > 0 && ks == (oq + 1 < NKS ? oq + 1 : oq)) {
          __builtin_amdgcn_sched_barrier(0);
          fold(oq - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    fold(NKS - 1);
    // add the 4 oi groups (fixed order), then group g owns channel 4*nq + g of rows m0..m0+3
    const int ch = 4 * nq + g;
    f32x4 o;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 v = {oacc[rt][0][0], oacc[rt][0][1], oacc[rt][1][0], oacc[rt][1][1]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = v[r];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        v[r] = t;
      }
      o[rt] = (g == 0 ? v[0] : (g == 1 ? v[1] : (g == 2 ? v[2] : v[3]))) * 2.f;  // the masked layer's factor
    }
    if (ch < c.n_hidden) {
      if (on) *reinterpret_cast<f32x4*>(xn + (bl * c.n_hidden + ch) * c.D + dd) = o;
    } else if (ch < c.L) {
      *reinterpret_cast<f32x4*>(outs + (ch - c.n_hidden) * Tp + m0) = o;
    }
  }
  if (npool > 0) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < npool * nb; idx += blockDim.x) {
      const int p = idx / nb;
      const int bq = idx - p * nb;
      float s = 0.f;
      for (int d = 0; d < c.D; ++d) s += outs[p * Tp + bq * c.D + d];  // reduce_sum over D (:270)
      pooled[(b0 + bq) * pool_stride + pool_off + p] = s;
    }
  }
}

// ---------------------------------------------------------------- input gradients
// LDS image of the filter for one f tile, built once per call: wpk[ft][fi][Hp][Lp] with
// zeros where the pair (4ft+fi, h) is not part of the layer (padding, or h <= f when masked)
__global__ void cin_pack_w_kernel(CinDims c, const float* __restrict__ w, int Hp, int Lp, int nft,
                                  float* __restrict__ wpk) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)nft * 4 * Hp * Lp;
  if (idx >= total) return;
  const int n = (int)(idx % Lp);
  const int64_t t = idx / Lp;
  const int h = (int)(t % Hp);
  const int f = (int)(t / Hp);  // = 4*ft + fi
  const bool ok = f < c.F && h < c.H && n < c.L && (!c.masked || h > f);
  wpk[idx] = ok ? w[(int64_t)(f * c.H + h) * cin_ldw(c) + n] : 0.f;
}

// Block = 4 waves x 16 rows.  Pair tile = 4 f x 4 h (lane&15 = fi*4 + hi).
// A[i=row][k=ch] = dcurr[row][4s+k]   (registers, SMAX steps)
// B[k=ch][j=pair] = w[(f,h)][4s+k]    (LDS copy of wpk[ft])
// D[row=(lane>>4)*4+r][pair=lane&15] = T
template <int NHT, int SMAX, bool PREFETCH>
__global__ __launch_bounds__(256) void cin_bwd_dx_mfma_kernel(
    CinDims c, const float* __restrict__ x0, const float* __restrict__ xk,
    const float* __restrict__ wpk, const float* __restrict__ dc_mn, float* d_xk, float* d_x0,
    int SPB, int Lp) {
  extern __shared__ float lds[];
  constexpr int ROWS = 64;
  constexpr int Tp = ROWS + kPad;
  const bool shared_x = (xk == x0);
  const int Hp = (c.H + 3) & ~3, Fp = (c.F + 3) & ~3;
  const int nht = Hp >> 2, nft = Fp >> 2;
  float* xs = lds;                                   // [Hp][Tp]
  float* as = shared_x ? xs : xs + Hp * Tp;          // [Fp][Tp]
  float* dxk_s = as + Fp * Tp;                       // [Hp][Tp]
  float* dx0_s = dxk_s + Hp * Tp;                    // [Fp][Tp]
  float* wst = dx0_s + Fp * Tp;                      // [4][Hp][Lp]

  const int64_t b0 = (int64_t)blockIdx.x * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  const int cols = nb * c.D;
  stage_cols(xk, b0, nb, c.H, Hp, c.D, xs, Tp);
  if (!shared_x) stage_cols(x0, b0, nb, c.F, Fp, c.D, as, Tp);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int fi = j >> 2, hi = j & 3;
  const int S = (c.L + 3) >> 2;
  // A operands: this lane's row (for the A role: row = lane&15) and channels 4s + kk
  float dcA[SMAX];
  {
    const int rl = wave * 16 + j;
    const int64_t m = b0 * c.D + rl;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const int ch = 4 * s + kk;
      dcA[s] = (s < S && rl < cols && ch < c.L) ? dc_mn[m * c.L + ch] : 0.f;
    }
  }
  const int rrow = wave * 16 + kk * 4;  // first of this lane's 4 result rows
  f32x4 accK[NHT];
#pragma unroll
  for (int ht = 0; ht < NHT; ++ht) accK[ht] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int per_ft4 = Hp * Lp;  // float4 count of one wpk[ft] image (4*Hp*Lp floats)
  // When the image fits 4 float4 per thread, the next f tile's image is fetched into registers
  // while this one multiplies, so the global latency is off the critical path.
  constexpr bool prefetch = PREFETCH;  // host guarantees per_ft4 <= 4*256 when set
  // four named registers (an indexed array ends up in scratch here)
  float4 wn0 = {0, 0, 0, 0}, wn1 = wn0, wn2 = wn0, wn3 = wn0;
  const int pq0 = threadIdx.x, pq1 = threadIdx.x + 256, pq2 = threadIdx.x + 512, pq3 = threadIdx.x + 768;
  const int last4 = per_ft4 - 1;
#define XDFM_FETCH_W(FT)                                                                     \
  do {                                                                                       \
    const float4* src_ = reinterpret_cast<const float4*>(wpk) + (size_t)(FT) * per_ft4;      \
    wn0 = src_[pq0 < per_ft4 ? pq0 : last4];                                                 \
    wn1 = src_[pq1 < per_ft4 ? pq1 : last4];                                                 \
    wn2 = src_[pq2 < per_ft4 ? pq2 : last4];                                                 \
    wn3 = src_[pq3 < per_ft4 ? pq3 : last4];                                                 \
  } while (0)
  if constexpr (prefetch) XDFM_FETCH_W(0);
  for (int ft = 0; ft < nft; ++ft) {
    __syncthreads();  // previous tile row is done with wst (and the X tiles are staged)
    {
      const float4* src = reinterpret_cast<const float4*>(wpk) + (size_t)ft * per_ft4;
      float4* dst = reinterpret_cast<float4*>(wst);
      if constexpr (prefetch) {
        if (pq0 < per_ft4) dst[pq0] = wn0;
        if (pq1 < per_ft4) dst[pq1] = wn1;
        if (pq2 < per_ft4) dst[pq2] = wn2;
        if (pq3 < per_ft4) dst[pq3] = wn3;
      } else {
        for (int q0 = threadIdx.x; q0 < per_ft4; q0 += 4 * 256) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * 256;
            v[u] = src[q < per_ft4 ? q : per_ft4 - 1];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q0 + u * 256 < per_ft4) dst[q0 + u * 256] = v[u];
        }
      }
    }
    __syncthreads();
    if constexpr (prefetch) XDFM_FETCH_W(ft + 1 < nft ? ft + 1 : ft);
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(as + (4 * ft + fi) * Tp + rrow);
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ht_begin = c.masked ? ft : 0;
    const float* wl = wst + (fi * Hp + hi) * Lp + kk;
    const float* xl = xs + hi * Tp + rrow;
    // two h tiles per trip: two independent accumulate chains keep the matrix pipe busy
#pragma unroll
    for (int hp = 0; hp < (NHT + 1) / 2; ++hp) {
      const int h0 = 2 * hp, h1 = 2 * hp + 1;
      const bool on0 = h0 >= ht_begin && h0 < nht;
      const bool on1 = h1 < NHT && h1 >= ht_begin && h1 < nht;
      if (on0 && on1) {
        f32x4 T0 = f32x4{0.f, 0.f, 0.f, 0.f}, T1 = T0;
        float w0[SMAX], w1[SMAX];
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {  // Lp = 4*SMAX + 1: steps beyond L read staged zeros
          w0[s] = wl[4 * h0 * Lp + 4 * s];
          w1[s] = wl[4 * h1 * Lp + 4 * s];
        }
        const f32x4 x0v = *reinterpret_cast<const f32x4*>(xl + 4 * h0 * Tp);
        const f32x4 x1v = *reinterpret_cast<const f32x4*>(xl + 4 * h1 * Tp);
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
          T0 = mfma(dcA[s], w0[s], T0);
          T1 = mfma(dcA[s], w1[s], T1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc0[r] = fmaf(x0v[r], T0[r], acc0[r]);
          accK[h0][r] = fmaf(a0[r], T0[r], accK[h0][r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc0[r] = fmaf(x1v[r], T1[r], acc0[r]);
          accK[h1 < NHT ? h1 : 0][r] = fmaf(a0[r], T1[r], accK[h1 < NHT ? h1 : 0][r]);
        }
      } else if (on0 || on1) {
        const int ht = on0 ? h0 : (h1 < NHT ? h1 : 0);
        f32x4 T = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wp = wl + 4 * ht * Lp;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) T = mfma(dcA[s], wp[4 * s], T);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xl + 4 * ht * Tp);
        if (on0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc0[r] = fmaf(xv[r], T[r], acc0[r]);
            accK[h0][r] = fmaf(a0[r], T[r], accK[h0][r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc0[r] = fmaf(xv[r], T[r], acc0[r]);
            accK[h1 < NHT ? h1 : 0][r] = fmaf(a0[r], T[r], accK[h1 < NHT ? h1 : 0][r]);
          }
        }
      }
    }
    // dX0[rows][4ft+fi] = sum over hi (the 4 lanes of a quad), fixed order
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc0[r];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      acc0[r] = v;
    }
    if (hi == 0) *reinterpret_cast<f32x4*>(dx0_s + (4 * ft + fi) * Tp + rrow) = acc0;
  }
#undef XDFM_FETCH_W
  // dXk[rows][4ht+hi] = sum over fi (lanes j, j+4, j+8, j+12), fixed order
#pragma unroll
  for (int ht = 0; ht < NHT; ++ht) {
    if (ht < nht) {
      f32x4 v = accK[ht];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = v[r];
        t += __shfl_down(t, 8, 16);
        t += __shfl_down(t, 4, 16);
        v[r] = t;
      }
      if (fi == 0) *reinterpret_cast<f32x4*>(dxk_s + (4 * ht + hi) * Tp + rrow) = v;
    }
  }
  __syncthreads();
  const float scale = c.masked ? 2.f : 1.f;
  // d_x0[b][f][d] += ... ; layer 0 (xk == x0): both roles go to d_x0
  {
    const int total = nb * c.F * c.D;
    float* base = d_x0 + b0 * c.F * c.D;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int bl = idx / (c.F * c.D);
      const int rem = idx - bl * c.F * c.D;
      const int f = rem / c.D;
      const int d = rem - f * c.D;
      float v = dx0_s[f * Tp + bl * c.D + d];
      if (shared_x) v += dxk_s[f * Tp + bl * c.D + d];
      base[idx] += v * scale;
    }
  }
  if (!shared_x) {
    const int total = nb * c.H * c.D;
    float* base = d_xk + b0 * c.H * c.D;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int bl = idx / (c.H * c.D);
      const int rem = idx - bl * c.H * c.D;
      const int h = rem / c.D;
      const int d = rem - h * c.D;
      base[idx] = dxk_s[h * Tp + bl * c.D + d] * scale;
    }
  }
}

// ------------------------------------------------- input gradients, second form
// Same contraction as above with the MFMA roles transposed so that no sum crosses lanes
// inside the loop, and with everything that is not an MFMA or one of the 2 FMAs per product
// kept out of it (fp32 MFMAs and VALU ops share the SIMD lanes):
//   A[i][k] = w[(4ft + i>>2, 4ht + i&3)][4s + k]     packed image, streamed in tile order
//   B[k][j] = dcurr[row][4s + k]                     registers, read once from d_xnext/d_pooled
//   D[i = 4*fi + r][j]: lane (fi = lane>>4, j) holds T for f = 4ft+fi, h = 4ht+r, and the
//   lane's 4 rows (m0..m0+3, one per row tile):
//     dX0[row][4ft+fi] += sum_r xk[row][4ht+r] * T[r]      registers acc0[ft], no reduction
//     dXk[row][4ht+r]  += x0[row][4ft+fi] * T[r]           registers accK, the 4 fi groups
//                                                          are added once per h tile
// h tiles run from the last to the first: once tile ht is done its rows of the X tile in LDS
// are dead (also as x0 rows of the masked layer, which needs ft <= ht), so dXk replaces them
// in place and the kernel needs no LDS beyond the X tiles.
// The image keeps every h tile's row of f tiles padded to a multiple of kDxRing (zero tiles that are loaded,
// never multiplied): a tile's slot in the kernel's ring of registers is then its position in the row modulo
// kDxRing, a compile-time number in the unrolled row, while the stream of loads runs on across rows.
constexpr int kDxRing = 4;
__host__ __device__ inline int dx_row_tiles(int ht, int nft, int masked) {
  return masked ? (ht + 1 < nft ? ht + 1 : nft) : nft;
}
__host__ __device__ inline int dx_tiles_padded(int nft, int nht, int masked) {
  int n = 0;
  for (int ht = 0; ht < nht; ++ht) n += (dx_row_tiles(ht, nft, masked) + kDxRing - 1) & ~(kDxRing - 1);
  return n;
}

template <int S>
__device__ __forceinline__ void pack_dx_body(const CinDims& c, const float* __restrict__ w, int nft, int nht,
                                             int ntiles, float* __restrict__ wdx, int64_t idx) {
  constexpr int Sp = (S + 3) & ~3;
  const int64_t total = (int64_t)ntiles * 64 * Sp;  // ntiles = dx_tiles_padded()
  if (idx >= total) return;
  const int s = (int)(idx % Sp);
  const int lane = (int)((idx / Sp) & 63);
  int t = (int)(idx / (64 * Sp));
  // tile order of the kernel: ht descending, ft ascending (masked: ft <= ht), rows padded
  int ht = nht - 1, ft = 0;
  bool real = false;
  for (; ht >= 0; --ht) {
    const int lim = dx_row_tiles(ht, nft, c.masked);
    const int plim = (lim + kDxRing - 1) & ~(kDxRing - 1);
    if (t < plim) {
      ft = t;
      real = t < lim;
      break;
    }
    t -= plim;
  }
  const int i = lane & 15, k = lane >> 4;
  const int f = 4 * ft + (i >> 2), h = 4 * ht + (i & 3), n = 4 * s + k;
  const bool ok = real && f < c.F && h < c.H && n < c.L && (!c.masked || h > f);
  wdx[idx] = ok ? w[(int64_t)(f * c.H + h) * cin_ldw(c) + n] : 0.f;
}

template <int S>
__global__ void cin_pack_dx_kernel(CinDims c, const float* __restrict__ w, int nft, int nht,
                                   int ntiles, float* __restrict__ wdx) {
  pack_dx_body<S>(c, w, nft, nht, ntiles, wdx, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Filter images of several layers in one launch (xdfm_cin_pack): a step packs once, ahead of
// the layer calls, instead of once per call on the critical path.
struct PackJob {
  CinDims c;
  const float* w;
  float* fwd;  // forward image or nullptr
  float* dx;   // input-gradient image or nullptr
  int noq, nks, nsteps_img, NQ, swap;
  int nft, nht, ntiles, SS;
  unsigned first_block, fwd_blocks, dx_blocks;
};
constexpr int kMaxPackJobs = 8;
struct PackJobs {
  PackJob j[kMaxPackJobs];
  int n;
};

__global__ void cin_pack_all_kernel(PackJobs jobs) {
  xdfm::Stamp stamp_(xdfm::kStampCinPack);
  int k = 0;
  while (k + 1 < jobs.n && blockIdx.x >= jobs.j[k + 1].first_block) ++k;
  const PackJob& J = jobs.j[k];
  const unsigned local = blockIdx.x - J.first_block;
  if (local < J.fwd_blocks) {
    pack_fwd_body(J.c, J.w, J.noq, J.nks, J.nsteps_img, J.NQ, J.swap, J.fwd,
                  (int64_t)local * blockDim.x + threadIdx.x);
  } else {
    const int64_t idx = (int64_t)(local - J.fwd_blocks) * blockDim.x + threadIdx.x;
    if (J.SS == 3)
      pack_dx_body<3>(J.c, J.w, J.nft, J.nht, J.ntiles, J.dx, idx);
    else
      pack_dx_body<5>(J.c, J.w, J.nft, J.nht, J.ntiles, J.dx, idx);
  }
}

struct Dx2Args {
  const float* wdx;
  float* d_xk;
  float* d_x0;
  int SPB, ntiles;
  // d_x0 = this layer's contribution + add0 + add1 (either may be null; add0 == d_x0 is the plain
  // "accumulate into").  The first layer to run writes, the last one can fold another producer's
  // gradient (the DNN tower's) in, so the table gradient reads ONE array per token.
  const float* add0;
  const float* add1;
};
struct BwdIn {  // what both backward kernels read
  CinDims c;
  const float* x0;
  const float* xk;
  const float* d_xnext;
  const float* d_pooled;
  int pool_stride, pool_off;
};

// SH < S ("cut", D a multiple of 4 so that a lane's 4 rows lie in one sample): a pooled channel's upstream
// gradient is the same for every row of a sample, so the pooled channels' part of T is ONE product chain
// per lane (steps P0..S-1, B = d_pooled of the lane's sample, hidden channels of a shared step zeroed)
// and the 4 row tiles' chains run over the hidden channels only (steps 0..SH-1, pooled channels of a
// shared step zeroed), starting from that part: SH 4 + (S - P0) matrix instructions per pair tile instead
// of 4 S (15 for 20 at n_hidden = 10, L = 20).  The filter image is the same either way.
template <int NFT, int S, int SH, int P0>
__device__ __forceinline__ void dx2_body(const BwdIn& in, const Dx2Args& ax, int bid, float* lds) {
  constexpr bool CUT = SH < S;
  constexpr int SPL = CUT ? S - P0 : 1;
  const CinDims& c = in.c;
  const float* __restrict__ x0 = in.x0;
  const float* __restrict__ xk = in.xk;
  const float* __restrict__ wdx = ax.wdx;
  const float* __restrict__ d_xnext = in.d_xnext;
  const float* __restrict__ d_pooled = in.d_pooled;
  const int pool_stride = in.pool_stride, pool_off = in.pool_off;
  float* d_xk = ax.d_xk;
  float* d_x0 = ax.d_x0;
  const int SPB = ax.SPB, ntiles = ax.ntiles;
  constexpr int RT = 4;
  constexpr int ROWS = 64 * RT;
  constexpr int Tp = ROWS + kPad;
  constexpr int Sp = (S + 3) & ~3;
  constexpr int S4 = Sp / 4;
  const bool shared_x = (xk == x0);
  const int Hp = (c.H + 3) & ~3, Fp = (c.F + 3) & ~3;
  const int nht = Hp >> 2, nft = Fp >> 2;
  float* xs = lds;                            // [Hp][Tp]  xk, then dXk (+ dX0 when shared)
  float* as = shared_x ? xs : xs + Hp * Tp;   // [Fp][Tp]  x0, then dX0

  const int64_t b0 = (int64_t)bid * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  const int cols = nb * c.D;
  stage_cols(xk, b0, nb, c.H, Hp, c.D, xs, Tp);
  if (!shared_x) stage_cols(x0, b0, nb, c.F, Fp, c.D, as, Tp);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int m0 = wave * 64 + 4 * j;  // this lane's 4 rows
  // B operands: dcurr[row][4s + g] for the lane's 4 rows, straight from the upstream tensors
  float dc[SH][RT];
  float dcp[SPL];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int m = m0 + rt;
    const bool valid = m < cols;
    const int mc = valid ? m : 0;
    const int bl = fdiv(mc, 1.0f / (float)c.D), d = mc - bl * c.D;
    const int64_t b = b0 + bl;
#pragma unroll
    for (int s = 0; s < SH; ++s) {
      const int n = 4 * s + g;
      float v = 0.f;
      if (CUT) {
        if (valid && n < c.n_hidden) v = d_xnext[(b * c.n_hidden + n) * c.D + d];
      } else if (valid && n < c.L) {
        v = n < c.n_hidden ? d_xnext[(b * c.n_hidden + n) * c.D + d]
                           : d_pooled[b * pool_stride + pool_off + n - c.n_hidden];
      }
      dc[s][rt] = v;
    }
    if (CUT && rt == 0) {
#pragma unroll
      for (int s = P0; s < S; ++s) {
        const int n = 4 * s + g;
        dcp[s - P0] = (valid && n >= c.n_hidden && n < c.L) ? d_pooled[b * pool_stride + pool_off + n - c.n_hidden] : 0.f;
      }
    }
  }
  __syncthreads();  // the X tiles are staged

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc0[NFT];  // [ft], component rt: dX0[m0+rt][4ft+g]
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) acc0[ft] = zero4;

  // The filter image streams through a ring of kDxRing tiles of registers, requested kDxRing - 1 tiles
  // (~2000 cycles of matrix work) ahead of their use: one tile ahead, every tile waited for the level-2
  // cache (a third of a wave's cycles: profiles/r02_pmc_sq.txt).  Rows are padded in the image, so the slot
  // of position ft is ft % kDxRing in every row.
  const f32x4* img = reinterpret_cast<const f32x4*>(wdx) + (size_t)lane * S4;
  const int tlast = ntiles - 1;
  constexpr int NPOS = (NFT + kDxRing - 1) & ~(kDxRing - 1);
  f32x4 ring[kDxRing][S4];
#pragma unroll
  for (int u = 0; u < kDxRing - 1; ++u)
#pragma unroll
    for (int q = 0; q < S4; ++q) ring[u][q] = img[(size_t)(u < tlast ? u : tlast) * 64 * S4 + q];
  int tp = 0;  // padded index of the row's first tile
  const float* xlane = xs + m0;
  const float* alane = as + g * Tp + m0;

  for (int ht = nht - 1; ht >= 0; --ht) {
    const int lim = dx_row_tiles(ht, nft, c.masked);
    const int plim = (lim + kDxRing - 1) & ~(kDxRing - 1);
    f32x4 xkv[4];  // [r], component rt: xk[m0+rt][4ht+r]
#pragma unroll
    for (int r = 0; r < 4; ++r) xkv[r] = *reinterpret_cast<const f32x4*>(xlane + (4 * ht + r) * Tp);
    f32x4 accK[RT];  // [rt], component r: this fi group's part of dXk[m0+rt][4ht+r]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) accK[rt] = zero4;
    f32x4 U[2][RT];
    f32x4 x0v[2];
    auto fold = [&](int ft, const f32x4 (&T)[RT], const f32x4& xv) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc0[ft][rt] = fmaf(xkv[r][rt], T[rt][r], acc0[ft][rt]);
          accK[rt][r] = fmaf(xv[rt], T[rt][r], accK[rt][r]);
        }
    };
#pragma unroll
    for (int ft = 0; ft < NPOS; ++ft) {
      if (ft < plim) {
        const int tn = tp + ft + kDxRing - 1 < tlast ? tp + ft + kDxRing - 1 : tlast;  // (clamped at the end)
#pragma unroll
        for (int q = 0; q < S4; ++q) ring[(ft + kDxRing - 1) % kDxRing][q] = img[(size_t)tn * 64 * S4 + q];
      }
      if (ft < NFT && ft < lim) {
        const f32x4(&a_cur)[S4] = ring[ft % kDxRing];
        x0v[ft & 1] = *reinterpret_cast<const f32x4*>(alane + 4 * ft * Tp);
        if constexpr (CUT) {
          f32x4 up = zero4;
#pragma unroll
          for (int s = P0; s < S; ++s) up = mfma(a_cur[s >> 2][s & 3], dcp[s - P0], up);
#pragma unroll
          for (int s = 0; s < SH; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              U[ft & 1][rt] = mfma(a_cur[s >> 2][s & 3], dc[s][rt], s == 0 ? up : U[ft & 1][rt]);
        } else {
#pragma unroll
          for (int s = 0; s < S; ++s)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              U[ft & 1][rt] = mfma(a_cur[s >> 2][s & 3], dc[s][rt], s == 0 ? zero4 : U[ft & 1][rt]);
        }
      }
      // the previous tile is folded in while this one multiplies
      if (ft >= 1 && ft - 1 < NFT && ft - 1 < lim) fold(ft - 1, U[(ft - 1) & 1], x0v[(ft - 1) & 1]);
    }
    if (NPOS == NFT && NFT - 1 < lim) fold(NFT - 1, U[(NFT - 1) & 1], x0v[(NFT - 1) & 1]);
    tp += plim;
    // add the 4 fi groups: after the two swaps lane group g holds the total of r = g
    f32x4 tot;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(accK[rt][0]), __float_as_uint(accK[rt][2]), false, false);
      auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(accK[rt][1]), __float_as_uint(accK[rt][3]), false, false);
      const float p0 = __uint_as_float(s02[0]) + __uint_as_float(s02[1]);
      const float p1 = __uint_as_float(s13[0]) + __uint_as_float(s13[1]);
      auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
      tot[rt] = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);
    }
    *reinterpret_cast<f32x4*>(xs + (4 * ht + g) * Tp + m0) = tot;  // rows 4ht.. of xk are dead
  }
  // dX0: lane (g, j) owns feature 4ft+g of its rows
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) {
    if (ft < nft) {
      f32x4* dst = reinterpret_cast<f32x4*>(as + (4 * ft + g) * Tp + m0);
      f32x4 v = acc0[ft];
      if (shared_x) v += *dst;  // layer 0: both roles are x0
      *dst = v;
    }
  }
  __syncthreads();
  const float scale = c.masked ? 2.f : 1.f;
  // results go out along d: [sample][feature][d] <- LDS [feature][row]; 16 bytes per lane when
  // 4 | D (the index arithmetic uses fdiv: this loop is a fifth of the kernel's VALU work otherwise)
  auto flush = [&](float* dst, const float* img, int R, const float* add0, const float* add1) {
    if ((c.D & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(add0) |
                            reinterpret_cast<uintptr_t>(add1)) & 15) == 0) {
      const int D4 = c.D >> 2, RD4 = R * D4;
      const float invRD4 = 1.0f / (float)RD4, invD4 = 1.0f / (float)D4;
      f32x4* base4 = reinterpret_cast<f32x4*>(dst + b0 * R * c.D);
      for (int idx = threadIdx.x; idx < nb * RD4; idx += blockDim.x) {
        const int bl = fdiv(idx, invRD4);
        const int rem = idx - bl * RD4;
        const int f = fdiv(rem, invD4);
        const int d4 = rem - f * D4;
        f32x4 v = *reinterpret_cast<const f32x4*>(img + f * Tp + bl * c.D + 4 * d4) * scale;
        if (add0) v = reinterpret_cast<const f32x4*>(add0 + b0 * R * c.D)[idx] + v;
        if (add1) v = v + reinterpret_cast<const f32x4*>(add1 + b0 * R * c.D)[idx];
        base4[idx] = v;
      }
    } else {
      const int RD = R * c.D;
      const float invRD = 1.0f / (float)RD, invD = 1.0f / (float)c.D;
      float* base = dst + b0 * R * c.D;
      for (int idx = threadIdx.x; idx < nb * RD; idx += blockDim.x) {
        const int bl = fdiv(idx, invRD);
        const int rem = idx - bl * RD;
        const int f = fdiv(rem, invD);
        const int d = rem - f * c.D;
        float v = img[f * Tp + bl * c.D + d] * scale;
        if (add0) v = add0[b0 * R * c.D + idx] + v;
        if (add1) v = v + add1[b0 * R * c.D + idx];
        base[idx] = v;
      }
    }
  };
  flush(d_x0, as, c.F, ax.add0, ax.add1);
  if (!shared_x) flush(d_xk, xs, c.H, nullptr, nullptr);
}

// ----------------------------------------------------------------- filter gradient
// tile t -> (ft, ht); the masked layer enumerates ht >= ft only, ft-major
__host__ __device__ inline int n_tiles(int nft, int nht, int masked) {
  return masked ? nft * (nft + 1) / 2 : nft * nht;
}
__host__ __device__ inline void tile_of(int t, int nft, int nht, int masked, int* ft, int* ht) {
  if (!masked) {
    *ft = t / nht;
    *ht = t - *ft * nht;
    return;
  }
  int f = 0, rem = t;
  while (rem >= nft - f) {
    rem -= nft - f;
    ++f;
  }
  *ft = f;
  *ht = f + rem;
}
__host__ __device__ inline int tile_index(int ft, int ht, int nft, int nht, int masked) {
  if (!masked) return ft * nht + ht;
  return ft * nft - ft * (ft - 1) / 2 + (ht - ft);
}

// Block = 4 waves sharing chunks of 64 rows; wave w owns PT pair tiles (4 f x 4 h each).
//   A[i=pair][k=row] = x0[row][f_i] * xk[row][h_i]     one v_mul per operand
//   B[k=row][j=ch]   = dcurr[row][c0 + j]
//   D[pair=(lane>>4)*4+r][ch=lane&15]
// Built to issue little besides MFMAs and that one multiply (fp32 MFMAs and VALU ops share the
// SIMD lanes): the chunk sits in LDS as [feature][row] and [channel][row], the MFMA's k index
// is mapped to rows as row = 8*stage + 2*k + step, so one ds_read_b64 at a loop-invariant
// address plus an immediate offset feeds two k-steps; the next chunk is fetched into registers
// (x0/xk from their tensors, dcurr straight from d_xnext / d_pooled) while this one multiplies
// and goes to the other LDS buffer afterwards: one barrier per chunk.
struct DwArgs {
  int64_t rows_total;
  int rows_per_split;
  float* part;
};

// SD > 0 (NT = 2, D = 8 SD rows per sample, chunks start on sample boundaries): the channel tiles are
// re-cut as [hidden channels | pooled channels].  A pooled channel's upstream gradient is the same for
// the D rows of a sample, so its filter gradient is sum_b d_pooled[b,n] (sum_d Z[(b,d),p]): the wave adds
// the Z fragments of a sample's SD stages (one v_add beside each product of the hidden tile) and issues
// ONE product per sample for the pooled tile instead of 2 SD -- 5 matrix instructions per sample and
// pair tile instead of 8 at D = 16.  (The d-sum of a lane covers its own rows; the instruction's sum over
// k joins the four lane groups.)
template <int PT, int NT, int XU, int SD>
__device__ __forceinline__ void dw_body(const BwdIn& in, const DwArgs& aw, int bx, int by, int bz,
                                        float* lds) {
  static_assert(SD == 0 || NT == 2, "the hidden | pooled cut needs two channel tiles");
  const CinDims& c = in.c;
  const float* __restrict__ x0 = in.x0;
  const float* __restrict__ xk = in.xk;
  const float* __restrict__ d_xnext = in.d_xnext;
  const float* __restrict__ d_pooled = in.d_pooled;
  const int pool_stride = in.pool_stride, pool_off = in.pool_off;
  const int64_t rows_total = aw.rows_total;
  const int rows_per_split = aw.rows_per_split;
  float* __restrict__ part = aw.part;
  constexpr int RC = 64;    // rows per chunk
  constexpr int RCp = 80;   // row stride of the LDS images (floats)
  constexpr int NCH = NT * 16;
  const bool shared_x = (xk == x0);
  const int Hp = (c.H + 3) & ~3, Fp = (c.F + 3) & ~3;
  const int nht = Hp >> 2, nft = Fp >> 2;
  const int ntile = n_tiles(nft, nht, c.masked);
  const int buf_floats = (Hp + (shared_x ? 0 : Fp) + NCH) * RCp;
  // buffer b: xs [Hp][RCp], as [Fp][RCp] (xs when shared), dcs [NCH][RCp]
  const int as_off = shared_x ? 0 : Hp * RCp;
  const int dc_off = (Hp + (shared_x ? 0 : Fp)) * RCp;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int fi = j >> 2, hi = j & 3;
  const int c0 = bz * NCH;
  const int t0 = (by * 4 + wave) * PT;

  // LDS addresses of this lane's operands in the current buffer; they move to the other buffer
  // and back with one add per chunk
  int a_ptr[PT], x_ptr[PT], b_ptr[NT];  // float offsets into lds
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    int ft = 0, ht = 0;
    if (t0 + p < ntile) tile_of(t0 + p, nft, nht, c.masked, &ft, &ht);
    a_ptr[p] = as_off + (4 * ft + fi) * RCp + 2 * kk;
    x_ptr[p] = (4 * ht + hi) * RCp + 2 * kk;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b_ptr[nt] = dc_off + (nt * 16 + j) * RCp + 2 * kk;
  int buf_step = buf_floats;  // +buf_floats, -buf_floats, ...

  f32x4 acc[PT][NT];
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[p][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t r_beg = (int64_t)bx * rows_per_split;
  int64_t r_end = r_beg + rows_per_split;
  if (r_end > rows_total) r_end = rows_total;
  const int HD = c.H * c.D, FD = c.F * c.D;

  // staging: thread t serves row t & 63 of the chunk and the features / channels 4u + (its wave).
  // Addresses are a wave-uniform base (scalar registers) plus one 32-bit lane offset, and LDS
  // stores use immediate offsets, so nothing but the values themselves stays live over the loop.
  const int srow = threadIdx.x & 63;
  const int sq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // column of the channel tiles -> channel (>= L: none).  SD > 0: columns 0..15 are the hidden channels,
  // 16..31 the pooled ones.
  auto ch_of = [&](int col) {
    if (SD == 0) return col;
    if (col < 16) return col < c.n_hidden ? col : c.L;
    return c.n_hidden + (col - 16);
  };
  float vx[XU], va[XU], vd[NCH / 4];
  bool row_ok = false;  // of the fetched chunk; values are masked when they go to LDS, so that
                        // nothing waits for the loads before the multiply loop
  auto fetch = [&](int64_t r0) {
    const int64_t m = r0 + srow;
    row_ok = m < r_end;
    const int64_t mc = row_ok ? m : r_beg;
    const int b = (int)(mc / c.D);  // B*F*D < 2^31 (checked on the host)
    const int d = (int)(mc - (int64_t)b * c.D);
    const int xo = b * HD + d, ao = b * FD + d;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int h = 4 * u + sq;
      const float* hb = xk + (h < c.H ? h : c.H - 1) * c.D;  // uniform
      vx[u] = hb[xo];
    }
    if (!shared_x) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int f = 4 * u + sq;
        const float* fb = x0 + (f < c.F ? f : c.F - 1) * c.D;
        va[u] = fb[ao];
      }
    }
    const int no = b * c.n_hidden * c.D + d, po = b * pool_stride + pool_off - c.n_hidden;
#pragma unroll
    for (int u = 0; u < NCH / 4; ++u) {
      const int ch = ch_of(c0 + 4 * u + sq);
      const int chc = ch < c.L ? ch : c.L - 1;  // uniform
      if (chc < c.n_hidden)
        vd[u] = (d_xnext + chc * c.D)[no];
      else
        vd[u] = (d_pooled + chc)[po];
    }
  };
  auto stash = [&](float* buf) {
    float* base = buf + sq * RCp + srow;
#pragma unroll
    for (int u = 0; u < XU; ++u)
      if (4 * u + sq < Hp) base[4 * u * RCp] = (row_ok && 4 * u + sq < c.H) ? vx[u] : 0.f;
    if (!shared_x) {
#pragma unroll
      for (int u = 0; u < XU; ++u)
        if (4 * u + sq < Fp) base[as_off + 4 * u * RCp] = (row_ok && 4 * u + sq < c.F) ? va[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NCH / 4; ++u)
      base[dc_off + 4 * u * RCp] = (row_ok && ch_of(c0 + 4 * u + sq) < c.L) ? vd[u] : 0.f;
  };

  fetch(r_beg);
  stash(lds);
  __syncthreads();
  int cur = 0;
  for (int64_t r0 = r_beg; r0 < r_end; r0 += RC) {
    const bool more = r0 + RC < r_end;
    if (more) fetch(r0 + RC);
    if (t0 < ntile) {
      // stage s = rows 8s .. 8s+7 (two k-steps), in two units of PT/2 tiles: the operands of the
      // next unit are read while this one multiplies
      constexpr int HP = (PT + 1) / 2;  // unit 0: tiles 0..HP-1, unit 1: tiles HP..PT-1
      struct Unit {
        f32x2 a[HP], x[HP];
      };
      f32x2 bS[2][NT];
      float zs[SD > 0 ? PT : 1];  // SD > 0: this lane's Z summed over its rows of the current sample
      auto issue = [&](Unit& un, int s, int hf) {
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          if (hf * HP + p < PT) {
            un.a[p] = *reinterpret_cast<const f32x2*>(lds + a_ptr[hf * HP + p] + 8 * s);
            un.x[p] = *reinterpret_cast<const f32x2*>(lds + x_ptr[hf * HP + p] + 8 * s);
          }
        }
        if (hf == 0) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bS[s & 1][nt] = *reinterpret_cast<const f32x2*>(lds + b_ptr[nt] + 8 * s);
        }
      };
      auto compute = [&](const Unit& un, int s, int hf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < HP; ++p) {
            if (hf * HP + p < PT) {
              const float z = un.a[p][ks] * un.x[p][ks];
              if constexpr (SD == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  acc[hf * HP + p][nt] = mfma(z, bS[s & 1][nt][ks], acc[hf * HP + p][nt]);
              } else {
                acc[hf * HP + p][0] = mfma(z, bS[s & 1][0][ks], acc[hf * HP + p][0]);
                const bool first = (s % SD == 0) && ks == 0, last = (s % SD == SD - 1) && ks == 1;
                zs[hf * HP + p] = first ? z : zs[hf * HP + p] + z;
                if (last) acc[hf * HP + p][1] = mfma(zs[hf * HP + p], bS[s & 1][1][ks], acc[hf * HP + p][1]);
              }
            }
          }
      };
      Unit u0, u1;
      issue(u0, 0, 0);
#pragma unroll
      for (int s = 0; s < RC / 8; ++s) {
        issue(u1, s, 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(u0, s, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < RC / 8) issue(u0, s + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        compute(u1, s, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) {
      stash(lds + (cur ^ 1) * buf_floats);  // nobody reads that buffer during this chunk
      __syncthreads();
      cur ^= 1;
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        a_ptr[p] += buf_step;
        x_ptr[p] += buf_step;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_ptr[nt] += buf_step;
      buf_step = -buf_step;
    }
  }
  // part[split][tile*16 + pair][L]
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    if (t0 + p < ntile) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ch = ch_of(c0 + nt * 16 + j);
        if (ch < c.L) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t slot = (int64_t)bx * ntile * 16 + (int64_t)(t0 + p) * 16 + kk * 4 + r;
            part[slot * c.L + ch] = acc[p][nt][r];
          }
        }
      }
    }
  }
}

template <int NFT, int S, int SH, int P0>
__global__ __launch_bounds__(256) void cin_bwd_dx2_mfma_kernel(BwdIn in, Dx2Args ax) {
  xdfm::Stamp stamp_(xdfm::kStampCinDx);
  extern __shared__ float lds[];
  dx2_body<NFT, S, SH, P0>(in, ax, blockIdx.x, lds);
}

template <int PT, int NT, int XU, int SD>
__global__ __launch_bounds__(256) void cin_bwd_dw_mfma_kernel(BwdIn in, DwArgs aw) {
  xdfm::Stamp stamp_(xdfm::kStampCinDw);
  extern __shared__ float lds[];
  dw_body<PT, NT, XU, SD>(in, aw, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// d_w[(f,h), n] = scale * sum over splits; masked pairs get 0.  A block sums 32 elements:
// 8 lanes-groups each add every 8th split in order, then the 8 partial sums are added in
// group order (fixed association, bit-identical from run to run).
__global__ __launch_bounds__(256) void cin_dw_unpack_kernel(CinDims c, const float* __restrict__ part,
                                                            int n_splits, float* __restrict__ d_w) {
  xdfm::Stamp stamp_(xdfm::kStampCinDwUnpack);
  __shared__ float red[8][33];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t idx = (int64_t)blockIdx.x * 32 + e;
  const int64_t total = (int64_t)c.F * c.H * c.L;
  float s = 0.f;
  bool live = false;
  if (idx < total) {
    const int p = (int)(idx / c.L);
    const int n = (int)(idx - (int64_t)p * c.L);
    const int f = p / c.H, h = p - f * c.H;
    const int nft = (c.F + 3) >> 2, nht = (c.H + 3) >> 2;
    if (!(c.masked && h <= f)) {
      live = true;
      const int t = tile_index(f >> 2, h >> 2, nft, nht, c.masked);
      const int ntile = n_tiles(nft, nht, c.masked);
      const int64_t slot = (int64_t)t * 16 + (f & 3) * 4 + (h & 3);
      const int64_t stride = (int64_t)ntile * 16 * c.L;
      const float* src = part + slot * c.L + n;
      // 8 loads in flight, added in split order
      for (int k0 = g; k0 < n_splits; k0 += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 8 * u;
          v[u] = src[(k < n_splits ? k : k0) * stride];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (k0 + 8 * u < n_splits) ? v[u] : 0.f;
      }
    }
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && idx < total) {
    float t = 0.f;
    if (live) {
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q][e];
    }
    const int64_t pr = idx / c.L;
    d_w[pr * cin_ldw(c) + (idx - pr * c.L)] = t * (c.masked ? 2.f : 1.f);
  }
}

// upstream gradient of curr in [m][L] order (m = b*D+d)
__global__ void cin_dcurr_kernel(CinDims c, const float* __restrict__ d_xnext,
                                 const float* __restrict__ d_pooled, int pool_stride, int pool_off,
                                 float* __restrict__ dc_mn) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = c.B * c.D * c.L;
  if (idx >= total) return;
  const int64_t m = idx / c.L;
  const int n = (int)(idx - m * c.L);
  const int64_t b = m / c.D;
  const int d = (int)(m - b * c.D);
  dc_mn[idx] = (n < c.n_hidden) ? d_xnext[(b * c.n_hidden + n) * c.D + d]
                                : d_pooled[b * pool_stride + pool_off + n - c.n_hidden];
}

constexpr size_t kLdsLimit = 160 * 1024;

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return XDFM_OK;
}

// SMAX instantiation of the input-gradient kernel that serves S = ceil(L/4) channel steps
int smax_for(int S) { return S <= 3 ? 3 : (S <= 5 ? 5 : (S <= 16 ? 16 : 64)); }

int dw_splits(int64_t rows) {
  int64_t s = (rows + 511) / 512;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

}  // namespace

bool mfma_supported(int64_t B, int F, int H, int D, int L) {
  if (D > 64 || F > 128 || H > 128 || F < 1 || H < 1) return false;
  const int Hp = (H + 3) & ~3, Fp = (F + 3) & ~3;
  const int S = (L + 3) / 4;
  if (S > 64) return false;
  const int Lp = 4 * smax_for(S) + 1;
  const size_t dx_lds = ((size_t)(2 * Hp + 2 * Fp) * (64 + kPad) + (size_t)4 * Hp * Lp) * sizeof(float);
  if (dx_lds > kLdsLimit) return false;
  // forward tile: both inputs and the pooled channels of 256 rows
  if ((size_t)(Hp + Fp + L) * (256 + kPad) * sizeof(float) > kLdsLimit) return false;
  // the filter-gradient kernel addresses its inputs with 32-bit element offsets
  if (B * (int64_t)(F > H ? F : H) * D >= (int64_t)1 << 31 || B * (int64_t)L * D >= (int64_t)1 << 31) return false;
  return true;
}

// XDFM_CIN_FWD_STATIC=0 keeps the masked layer on the generic forward kernel (A/B runs, tests of both)
static bool fwd_static_off() {
  const char* e = getenv("XDFM_CIN_FWD_STATIC");
  return e && e[0] == '0';
}

// which input is the K side of the forward product: the longer feature axis
static bool fwd_swap(int F, int H, int masked) { return !masked && F > H; }

size_t fwd_mfma_workspace_floats(int64_t B, int F, int H, int D, int L) {
  (void)B;
  (void)D;
  // unmasked count with the longer axis as K bounds every case
  const int Ki = F > H ? F : H, Ko = F > H ? H : F;
  const int noq = (Ko + 3) / 4, nks = (Ki + 3) / 4, NQ = (L + 3) / 4;
  const int nsteps_img = (fwd2_steps(noq, nks, 0) + 15) & ~15;
  return (size_t)NQ * nsteps_img * 64 + nsteps_img;
}

// w == nullptr: `ws` already holds the image (xdfm_cin_pack)
int fwd_mfma(const CinDims& c, const float* x0, const float* xk, const float* w, float* xnext,
             float* pooled, int pool_stride, int pool_off, float* ws, hipStream_t st) {
  const bool swap = fwd_swap(c.F, c.H, c.masked);
  const float* xi = swap ? x0 : xk;
  const float* xo = swap ? xk : x0;
  const int Ki = swap ? c.F : c.H, Ko = swap ? c.H : c.F;
  const int Kip = (Ki + 3) & ~3, Kop = (Ko + 3) & ~3;
  const int noq = Kop / 4, nks = Kip / 4, NQ = (c.L + 3) / 4;
  const int nsteps = fwd2_steps(noq, nks, c.masked);
  const int ngrp = (nsteps + 3) / 4;
  // the image holds a multiple of 4 groups (16 steps) per nq: the kernel's load ring runs four
  // groups per trip
  const int nsteps_img = (nsteps + 15) & ~15;
  const int64_t img_floats = (int64_t)NQ * nsteps_img * 64;
  if (w)
    XDFM_LAUNCH("cin_pack_fwd_kernel", st, cin_pack_fwd_kernel, dim3((unsigned)((img_floats + 255) / 256)),
                dim3(256), 0, c, w, noq, nks, nsteps_img, NQ, swap ? 1 : 0, ws);
  constexpr int ROWS = 256;
  XDFM_REQUIRE(c.D <= ROWS, "xdfm_cin_fwd: dim %d does not fit a row tile", c.D);
  const int SPB = ROWS / c.D;
  if (c.masked && xi == xo && (c.D & 3) == 0 && nks <= 10 && !fwd_static_off() &&
      (reinterpret_cast<uintptr_t>(xi) & 15) == 0 && (reinterpret_cast<uintptr_t>(xnext) & 15) == 0) {
    const size_t lds_bytes = (size_t)(c.L - c.n_hidden) * (ROWS + kPad) * sizeof(float);
    XDFM_REQUIRE(lds_bytes <= kLdsLimit, "xdfm_cin_fwd: pooled channels do not fit the LDS");
    dim3 grid((unsigned)((c.B + SPB - 1) / SPB));
#define CALLF(N)                                                                                            \
  case N: {                                                                                                 \
    int rc = allow_lds(cin_fwd_static_kernel<N>, lds_bytes);                                                \
    if (rc) return rc;                                                                                      \
    XDFM_LAUNCH("cin_fwd_static_kernel<" #N ">", st, cin_fwd_static_kernel<N>, grid, dim3(256), lds_bytes, \
                c, xi, ws, xnext, pooled, pool_stride, pool_off, SPB, nsteps_img / 4, NQ);                  \
  } break
    switch (nks) {
      CALLF(1);
      CALLF(2);
      CALLF(3);
      CALLF(4);
      CALLF(5);
      CALLF(6);
      CALLF(7);
      CALLF(8);
      CALLF(9);
      CALLF(10);
    }
#undef CALLF
    XDFM_LAUNCH_CHECK();
    return XDFM_OK;
  }
  const int lds_rows = Kip + ((xi == xo) ? 0 : Kop) + (c.L - c.n_hidden);
  const size_t lds_bytes = (size_t)lds_rows * (ROWS + kPad) * sizeof(float);
  XDFM_REQUIRE(lds_bytes <= kLdsLimit, "xdfm_cin_fwd: tile does not fit the LDS");
  dim3 grid((unsigned)((c.B + SPB - 1) / SPB));
  int rc = allow_lds(cin_fwd2_mfma_kernel, lds_bytes);
  if (rc) return rc;
  XDFM_LAUNCH("cin_fwd2_mfma_kernel", st, cin_fwd2_mfma_kernel, grid, dim3(256), lds_bytes, c, xi, xo, Ki, Ko,
              ws, reinterpret_cast<const int*>(ws + img_floats), xnext, pooled, pool_stride, pool_off, SPB,
              ngrp, NQ);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// filter image of the input-gradient kernels: the larger of the two layouts (the unmasked tile
// count bounds the masked one)
static size_t wpk_region_floats(int F, int H, int L) {
  const int nft = (F + 3) / 4, nht = (H + 3) / 4, Hp = (H + 3) & ~3, S = (L + 3) / 4;
  const size_t old_img = (size_t)nft * 4 * Hp * (4 * smax_for(S) + 1);
  const size_t new_img = (size_t)dx_tiles_padded(nft, nht, 0) * 64 * (S <= 3 ? 4 : 8);  // S <= 5 only
  return ((old_img > new_img ? old_img : new_img) + 3) & ~(size_t)3;
}

size_t bwd_mfma_workspace_floats(int64_t B, int F, int H, int D, int L, int masked) {
  const int nft = (F + 3) / 4, nht = (H + 3) / 4;
  const int ntile = n_tiles(nft, nht, masked);
  const size_t wpk = wpk_region_floats(F, H, L);
  return (((size_t)B * D * L + 3) & ~(size_t)3) + wpk + (size_t)dw_splits(B * D) * ntile * 16 * L;
}

static bool dx2_applies(int F, int H, int L, bool shared_x) {
  const int Hp = (H + 3) & ~3, Fp = (F + 3) & ~3, S = (L + 3) / 4;
  return Fp / 4 <= 10 && S <= 5 && (size_t)(Hp + (shared_x ? 0 : Fp)) * (256 + kPad) * sizeof(float) <= kLdsLimit;
}

// w == nullptr: `packed_dx` holds the input-gradient image (xdfm_cin_pack)
int bwd_mfma(const CinDims& c, const float* x0, const float* xk, const float* w, const float* packed_dx,
             const float* d_xnext, const float* d_pooled, int pool_stride, int pool_off, float* d_xk,
             float* d_x0, float* d_w, float* ws, int phases, hipStream_t st, const float* add0, const float* add1) {
  float* dc_mn = ws;
  float* wpk = ws + (((size_t)c.B * c.D * c.L + 3) & ~(size_t)3);  // 16-byte aligned: staged as float4
  const size_t wpk_floats = (size_t)((c.F + 3) / 4) * 4 * ((c.H + 3) & ~3) * (4 * smax_for((c.L + 3) / 4) + 1);
  float* part = wpk + wpk_region_floats(c.F, c.H, c.L);
  const int Hp = (c.H + 3) & ~3, Fp = (c.F + 3) & ~3;
  const int nht = Hp / 4, nft = Fp / 4;
  const int S = (c.L + 3) / 4;
  const bool shared_x = (xk == x0);
  const BwdIn in{c, x0, xk, d_xnext, d_pooled, pool_stride, pool_off};

  // ---- filter gradient geometry
  const int64_t rows = c.B * c.D;
  int n_splits = dw_splits(rows);
  int rows_per_split = (int)((rows + n_splits - 1) / n_splits);
  // the hidden | pooled cut of the channel tiles (dw_body): both kinds present and at most 16 each, a
  // sample = 1, 2 or 4 stages of 8 rows, splits (and with them the chunks of 64 rows) on sample boundaries
  const int n_pooled = c.L - c.n_hidden;
  int SD = 0;
  if (d_pooled && c.n_hidden > 0 && c.n_hidden <= 16 && n_pooled > 0 && n_pooled <= 16 &&
      (c.D == 8 || c.D == 16 || c.D == 32)) {
    SD = c.D / 8;
    rows_per_split = (rows_per_split + 63) / 64 * 64;
    n_splits = (int)((rows + rows_per_split - 1) / rows_per_split);  // no split starts past the end
  }
  const int ntile = n_tiles(nft, nht, c.masked);
  const int nt_total = SD ? 2 : (c.L + 15) / 16;
  const int NT = nt_total >= 2 ? 2 : 1;
  // pair tiles per wave: 7 when that wastes fewer wave slots than 8 (55 tiles: 56 vs 64)
  const int maxp = Hp > Fp ? Hp : Fp;
  const int slots8 = (ntile + 31) / 32 * 32, slots7 = (ntile + 27) / 28 * 28;
  const int PT = (maxp <= 40 && slots7 < slots8) ? 7 : 8;
  const size_t dw_lds = (size_t)2 * (Hp + (shared_x ? 0 : Fp) + NT * 16) * 80 * sizeof(float);
  const dim3 dw_grid((unsigned)n_splits, (unsigned)((ntile + 4 * PT - 1) / (4 * PT)),
                     (unsigned)((nt_total + NT - 1) / NT));
  const DwArgs aw{rows, rows_per_split, part};

  // ---- input gradients
  const size_t dx2_lds = (size_t)(Hp + (shared_x ? 0 : Fp)) * (256 + kPad) * sizeof(float);
  XDFM_REQUIRE(w || (packed_dx && dx2_applies(c.F, c.H, c.L, shared_x)),
               "xdfm_cin_bwd_packed: this shape has no packed input-gradient path");
  XDFM_REQUIRE(phases == 3 || !w, "xdfm_cin_bwd: the phases split needs the packed path");
  const bool dw_partial = (phases & (XDFM_CIN_BWD_DW | XDFM_CIN_BWD_DW_PARTIAL)) != 0;
  const bool dw_reduce = (phases & (XDFM_CIN_BWD_DW | XDFM_CIN_BWD_DW_REDUCE)) != 0;
  if (!(phases & 1)) {
    // filter gradient only
  } else if (nft <= 10 && S <= 5 && dx2_lds <= kLdsLimit) {
    const int SPB = 256 / c.D;
    const int ntiles = dx_tiles_padded(nft, nht, c.masked);
    const int n_dx = (int)((c.B + SPB - 1) / SPB);
    const Dx2Args ax{w ? wpk : packed_dx, d_xk, d_x0, SPB, ntiles, add0, add1};
    const int SS = S <= 3 ? 3 : 5;
    if (w) {
      const int64_t total = (int64_t)ntiles * 64 * ((SS + 3) & ~3);
      if (SS == 3)
        XDFM_LAUNCH("cin_pack_dx_kernel<3>", st, cin_pack_dx_kernel<3>, dim3((unsigned)((total + 255) / 256)),
                    dim3(256), 0, c, w, nft, nht, ntiles, wpk);
      else
        XDFM_LAUNCH("cin_pack_dx_kernel<5>", st, cin_pack_dx_kernel<5>, dim3((unsigned)((total + 255) / 256)),
                    dim3(256), 0, c, w, nft, nht, ntiles, wpk);
    }
    {
#define CALL2X(NF, SS_, SH_, P0_)                                                                      \
  do {                                                                                                 \
    int rc = allow_lds(cin_bwd_dx2_mfma_kernel<NF, SS_, SH_, P0_>, dx2_lds);                           \
    if (rc) return rc;                                                                                 \
    XDFM_LAUNCH("cin_bwd_dx2_mfma_kernel<" #NF ", " #SS_ ", " #SH_ ">", st,                            \
                (cin_bwd_dx2_mfma_kernel<NF, SS_, SH_, P0_>), dim3((unsigned)n_dx), dim3(256), dx2_lds, in, ax); \
  } while (0)
      // the hidden | pooled cut of the channel steps (dx2_body): n_hidden in 9..11 of up to 20 channels,
      // or 5..7 of up to 12
      const bool can_cut = d_pooled && (c.D & 3) == 0 && c.n_hidden > 0 && c.n_hidden < c.L;
      const int shh = (c.n_hidden + 3) / 4, p0 = c.n_hidden / 4;
      const bool cut5 = can_cut && SS == 5 && shh == 3 && p0 == 2, cut3 = can_cut && SS == 3 && shh == 2 && p0 == 1;
#define CALL2(NF, SS_)                   \
  do {                                   \
    if (SS_ == 5 && cut5)                \
      CALL2X(NF, 5, 3, 2);               \
    else if (SS_ == 3 && cut3)           \
      CALL2X(NF, 3, 2, 1);               \
    else                                 \
      CALL2X(NF, SS_, SS_, SS_);         \
  } while (0)
      if (nft <= 3) {
        if (SS == 3)
          CALL2(3, 3);
        else
          CALL2(3, 5);
      } else {
        if (SS == 3)
          CALL2(10, 3);
        else
          CALL2(10, 5);
      }
#undef CALL2X
#undef CALL2
    }
  } else {
    XDFM_REQUIRE(add0 == d_x0 && !add1, "xdfm_cin_bwd: this shape only accumulates into d_x0");
    {
      const int64_t total = c.B * c.D * c.L;
      XDFM_LAUNCH("cin_dcurr_kernel", st, cin_dcurr_kernel, dim3((unsigned)((total + 255) / 256)),
                  dim3(256), 0, c, d_xnext, d_pooled, pool_stride, pool_off, dc_mn);
    }
    const int SPB = 64 / c.D;
    const int Lp = 4 * smax_for(S) + 1;
    const size_t lds_bytes =
        ((size_t)((shared_x ? 2 : 2) * Hp + (shared_x ? 1 : 2) * Fp) * (64 + kPad) + (size_t)4 * Hp * Lp) *
        sizeof(float);
    dim3 grid((unsigned)((c.B + SPB - 1) / SPB));
    XDFM_LAUNCH("cin_pack_w_kernel", st, cin_pack_w_kernel, dim3((unsigned)((wpk_floats + 255) / 256)),
                dim3(256), 0, c, w, Hp, Lp, nft, wpk);
    const bool pre = (size_t)Hp * Lp <= 4 * 256;  // the f-tile filter image fits 4 float4 per thread
#define CALL_P(NH, SM, PF)                                                                             \
  do {                                                                                                 \
    int rc = allow_lds(cin_bwd_dx_mfma_kernel<NH, SM, PF>, lds_bytes);                                 \
    if (rc) return rc;                                                                                 \
    XDFM_LAUNCH("cin_bwd_dx_mfma_kernel<" #NH ", " #SM ">", st, (cin_bwd_dx_mfma_kernel<NH, SM, PF>),  \
                grid, dim3(256), lds_bytes, c, x0, xk, wpk, dc_mn, d_xk, d_x0, SPB, Lp);               \
  } while (0)
#define CALL_S(NH, SM)          \
  do {                          \
    if (pre)                    \
      CALL_P(NH, SM, true);     \
    else                        \
      CALL_P(NH, SM, false);    \
  } while (0)
#define BY_S(NH)               \
  do {                         \
    if (S <= 3)                \
      CALL_S(NH, 3);           \
    else if (S <= 5)           \
      CALL_S(NH, 5);           \
    else if (S <= 16)          \
      CALL_P(NH, 16, false);   \
    else                       \
      CALL_P(NH, 64, false);   \
  } while (0)
    if (nht <= 3)
      BY_S(3);
    else if (nht <= 10)
      BY_S(10);
    else
      BY_S(32);
#undef BY_S
#undef CALL_S
#undef CALL_P
  }
  // ---- filter gradient
  if (dw_partial) {
#define CALLW(PTV, NTV, XUV, SDV)                                                                     \
  do {                                                                                                \
    int rc = allow_lds(cin_bwd_dw_mfma_kernel<PTV, NTV, XUV, SDV>, dw_lds);                           \
    if (rc) return rc;                                                                                \
    XDFM_LAUNCH("cin_bwd_dw_mfma_kernel<" #PTV ", " #NTV ", " #SDV ">", st,                           \
                (cin_bwd_dw_mfma_kernel<PTV, NTV, XUV, SDV>), dw_grid, dim3(256), dw_lds, in, aw);    \
  } while (0)
#define CALLW_SD(PTV, XUV)      \
  do {                          \
    if (SD == 1)                \
      CALLW(PTV, 2, XUV, 1);    \
    else if (SD == 2)           \
      CALLW(PTV, 2, XUV, 2);    \
    else if (SD == 4)           \
      CALLW(PTV, 2, XUV, 4);    \
    else                        \
      CALLW(PTV, 2, XUV, 0);    \
  } while (0)
    if (NT == 2) {
      if (maxp > 40)
        CALLW_SD(8, 32);
      else if (PT == 7)
        CALLW_SD(7, 10);
      else
        CALLW_SD(8, 10);
    } else {
      if (maxp > 40)
        CALLW(8, 1, 32, 0);
      else if (PT == 7)
        CALLW(7, 1, 10, 0);
      else
        CALLW(8, 1, 10, 0);
    }
#undef CALLW_SD
#undef CALLW
  }
  if (dw_reduce) {
    const int64_t total = (int64_t)c.F * c.H * c.L;
    XDFM_LAUNCH("cin_dw_unpack_kernel", st, cin_dw_unpack_kernel, dim3((unsigned)((total + 31) / 32)),
                dim3(256), 0, c, part, n_splits, d_w);
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

void pack_floats(int64_t B, int F, int H, int D, int L, int masked, size_t* fwd_floats, size_t* bwd_floats) {
  *fwd_floats = *bwd_floats = 0;
  if (!mfma_supported(B, F, H, D, L)) return;
  *fwd_floats = (fwd_mfma_workspace_floats(B, F, H, D, L) + 3) & ~(size_t)3;
  // one image serves both roles of the layer's inputs: shared_x only makes the LDS need smaller
  if (dx2_applies(F, H, L, false)) {
    const int nft = (F + 3) / 4, nht = (H + 3) / 4, S = (L + 3) / 4;
    *bwd_floats = (size_t)dx_tiles_padded(nft, nht, masked) * 64 * (S <= 3 ? 4 : 8);
  }
}

int pack_all(const xdfm_cin_pack_job* jobs, int n_jobs, int64_t B, int D, hipStream_t st) {
  XDFM_REQUIRE(n_jobs >= 0 && n_jobs <= kMaxPackJobs, "xdfm_cin_pack: at most %d jobs per call", kMaxPackJobs);
  PackJobs pj;
  memset(&pj, 0, sizeof(pj));
  unsigned blocks = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const xdfm_cin_pack_job& in = jobs[k];
    XDFM_REQUIRE(in.w && in.F > 0 && in.H > 0 && in.L > 0, "xdfm_cin_pack: bad job %d", k);
    size_t ff, bf;
    pack_floats(B, in.F, in.H, D, in.L, in.masked, &ff, &bf);
    XDFM_REQUIRE(!in.fwd_image || ff, "xdfm_cin_pack: job %d has no packed forward path", k);
    XDFM_REQUIRE(!in.bwd_image || bf, "xdfm_cin_pack: job %d has no packed backward path", k);
    PackJob& J = pj.j[pj.n];
    J.c = CinDims{B, in.F, in.H, D, in.L, 0, in.masked ? 1 : 0, in.ldw};
    J.w = in.w;
    J.fwd = in.fwd_image;
    J.dx = in.bwd_image;
    J.first_block = blocks;
    if (in.fwd_image) {
      const bool swap = fwd_swap(in.F, in.H, in.masked);
      const int Ki = swap ? in.F : in.H, Ko = swap ? in.H : in.F;
      J.noq = ((Ko + 3) & ~3) / 4;
      J.nks = ((Ki + 3) & ~3) / 4;
      J.NQ = (in.L + 3) / 4;
      J.swap = swap ? 1 : 0;
      J.nsteps_img = (fwd2_steps(J.noq, J.nks, in.masked) + 15) & ~15;
      J.fwd_blocks = (unsigned)(((int64_t)J.NQ * J.nsteps_img * 64 + 255) / 256);
    }
    if (in.bwd_image) {
      J.nft = (in.F + 3) / 4;
      J.nht = (in.H + 3) / 4;
      J.ntiles = dx_tiles_padded(J.nft, J.nht, in.masked);
      J.SS = (in.L + 3) / 4 <= 3 ? 3 : 5;
      J.dx_blocks = (unsigned)(((int64_t)J.ntiles * 64 * ((J.SS + 3) & ~3) + 255) / 256);
    }
    if (J.fwd_blocks + J.dx_blocks == 0) continue;
    blocks += J.fwd_blocks + J.dx_blocks;
    ++pj.n;
  }
  if (!blocks) return XDFM_OK;
  XDFM_LAUNCH("cin_pack_all_kernel", st, cin_pack_all_kernel, dim3(blocks), dim3(256), 0, pj);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

}  // namespace xdfm_cin

XDFM_STAMP_UNIT(stamp_unit_cin)
