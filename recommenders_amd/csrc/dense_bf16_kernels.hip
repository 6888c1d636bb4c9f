// Dense layer y = act(x W + b) and its backward on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) for the wide DNN tower of BASELINE configs[4]
// (2496 -> 1024 -> 512) and the pooled-channel products of the CIN.  Reference:
// xDeepFM.py:491-507 (xw_plus_b + activation).  Same arguments and results as xdfm_dense_fwd /
// xdfm_dense_bwd up to the bf16 rounding of the operands (2e-2 relative bar).
//
// One kernel, three operand conventions (C[M,N] = sum_k A(m,k) B(k,n)):
//   forward          A = x[m][k]                     B = w[k][n]        C = act(. + b[n]) -> y
//   input gradient   A = dz[m][k] = dy * act'(y)     B = w[n][k]        C -> dx (written or added to)
//   weight gradient  A = x[k][m], row K_in = ones    B = dz[k][n]       C -> dw, last row -> db
// The operands are read as fp32 from where the fp32 path keeps them, dz is formed while staging (no
// buffer of its own), and tiles are rounded to bf16 on their way into LDS.  Block tile 128 x 128,
// k step 32, four waves of 64 x 64; the next tile's global loads fly during the current tile's
// products.  Products whose output has too few tiles to fill the chip are split along k into a
// workspace and summed in split order (bit-identical reruns).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "act.h"
#include "cin_bf16.h"
#include "xdfm_common.h"

using namespace xdfm_bf16;

namespace {

enum { kFwd = 0, kDx = 1, kDw = 2 };
constexpr int kBM = 128, kBN = 128, kBK = 32;
constexpr int kPitch = 20;  // uint32 per LDS row: 32 bf16 + 16 bytes

struct GemmArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* y;
  const float* dy;
  float* out;   // y / dx / dw (or the split workspace)
  float* out2;  // db
  int64_t B;
  int K, Nout, act, accumulate;
  int M, N, R;          // the product's sizes: C[M,N], reduction length R
  int splits, r_per_split;
};

template <int MODE>
__device__ __forceinline__ float a_at(const GemmArgs& g, int m, int k) {
  if (m >= g.M || k >= g.R) return 0.f;
  if (MODE == kFwd) return g.x[(int64_t)m * g.K + k];
  if (MODE == kDx) {
    const int64_t i = (int64_t)m * g.Nout + k;
    return g.act == XDFM_ACT_IDENTITY ? g.dy[i] : g.dy[i] * act_grad(g.y[i], g.act);
  }
  return m < g.K ? g.x[(int64_t)k * g.K + m] : 1.f;
}
template <int MODE>
__device__ __forceinline__ float b_at(const GemmArgs& g, int k, int n) {
  if (n >= g.N || k >= g.R) return 0.f;
  if (MODE == kFwd) return g.w[(int64_t)k * g.Nout + n];
  if (MODE == kDx) return g.w[(int64_t)n * g.Nout + k];
  const int64_t i = (int64_t)k * g.Nout + n;
  return g.act == XDFM_ACT_IDENTITY ? g.dy[i] : g.dy[i] * act_grad(g.y[i], g.act);
}

template <int MODE>
__device__ __forceinline__ void emit(const GemmArgs& g, int m, int n, float v) {
  if (m >= g.M || n >= g.N) return;
  if (MODE == kFwd) {
    g.out[(int64_t)m * g.Nout + n] = activate(v + g.bias[n], g.act);
  } else if (MODE == kDx) {
    float* dst = g.out + (int64_t)m * g.K + n;
    *dst = g.accumulate ? *dst + v : v;
  } else if (m < g.K) {
    g.out[(int64_t)m * g.Nout + n] = v;
  } else {
    g.out2[n] = v;
  }
}

// A_KFAST / B_KFAST: which index is contiguous in global memory decides which one consecutive
// threads walk (coalescing); either way a thread packs pairs along k.
// FAST (reduction length a multiple of 32 per split, operands below 2^31 elements): staging without
// branches -- rows past the end read the last row and are zeroed (or set to the bias row's ones) when
// stored, addresses are a per-thread offset computed once plus a pointer that advances per k step, and no
// select sits between a load and the products it is meant to fly beside.
template <int MODE, bool FAST>
__global__ __launch_bounds__(256, 2) void dense_bf16_kernel(GemmArgs g, float* __restrict__ part) {
  __shared__ __align__(16) uint32_t as[2][kBM * kPitch];
  __shared__ __align__(16) uint32_t bs[2][kBN * kPitch];
  constexpr bool A_KFAST = MODE != kDw, B_KFAST = MODE == kDx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int split = blockIdx.z;
  const int k_lo = split * g.r_per_split;
  int k_hi = k_lo + g.r_per_split;
  if (k_hi > g.R) k_hi = g.R;
  const int steps = k_hi > k_lo ? (k_hi - k_lo + kBK - 1) / kBK : 0;
  // the split's k range is cut off through R: rows past k_hi read as zero
  GemmArgs gs = g;
  gs.R = k_hi;

  float av[8][2], bv[8][2];
  float ay[8][2], by[8][2];  // FAST: the saved outputs that go with dy (act' is applied when stored)
  // FAST: operand pointers at this thread's first element of the split, element offsets of its 8 loads
  const float* pa = nullptr;
  const float* pa2 = nullptr;
  const float* pb = nullptr;
  const float* pb2 = nullptr;
  int offa[8], offb[8];
  uint32_t keepa = 0, keepb = 0, onesa = 0;
  int stepa = 0, stepb = 0;
  if (FAST) {
    // k-fast operands: 8 rows x one pair of k; row-fast operands: one row x 8 pairs of k
    const int lda = MODE == kFwd ? g.K : (MODE == kDx ? g.Nout : g.K);
    const int ldb = g.Nout;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (A_KFAST) {
        const int row = m0 + (tid >> 4) + 16 * i;
        offa[i] = (row < g.M ? row : g.M - 1) * lda + 2 * (tid & 15);
        keepa |= (row < g.M ? 1u : 0u) << i;
      } else {  // kDw: A(m,k) = x[k][m], row K_in = ones
        const int row = m0 + (tid & 127);
        const int rc = row < g.K ? row : g.K - 1;
        offa[i] = 2 * ((tid >> 7) + 2 * i) * lda + rc;
        keepa |= (row < g.K ? 1u : 0u) << i;
        onesa |= (row == g.K ? 1u : 0u) << i;
      }
      if (B_KFAST) {  // kDx: B(k,n) = w[n][k]
        const int row = n0 + (tid >> 4) + 16 * i;
        offb[i] = (row < g.N ? row : g.N - 1) * ldb + 2 * (tid & 15);
        keepb |= (row < g.N ? 1u : 0u) << i;
      } else {  // kFwd: w[k][n]; kDw: dz[k][n]
        const int row = n0 + (tid & 127);
        offb[i] = 2 * ((tid >> 7) + 2 * i) * ldb + (row < g.N ? row : g.N - 1);
        keepb |= (row < g.N ? 1u : 0u) << i;
      }
    }
    pa = (MODE == kDx ? g.dy : g.x) + (A_KFAST ? (int64_t)k_lo : (int64_t)k_lo * lda);
    pa2 = MODE == kDx ? g.y + k_lo : nullptr;
    pb = (MODE == kDw ? g.dy : g.w) + (B_KFAST ? (int64_t)k_lo : (int64_t)k_lo * ldb);
    pb2 = MODE == kDw ? g.y + (int64_t)k_lo * ldb : nullptr;
    stepa = A_KFAST ? kBK : kBK * lda;
    stepb = B_KFAST ? kBK : kBK * ldb;
  }
  const bool with_act = g.act != XDFM_ACT_IDENTITY;
  auto issue_fast = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (A_KFAST) {
        const float2 v = *reinterpret_cast<const float2*>(pa + offa[i]);
        av[i][0] = v.x;
        av[i][1] = v.y;
        if (MODE == kDx && with_act) {
          const float2 u = *reinterpret_cast<const float2*>(pa2 + offa[i]);
          ay[i][0] = u.x;
          ay[i][1] = u.y;
        }
      } else {
        av[i][0] = pa[offa[i]];
        av[i][1] = pa[offa[i] + g.K];
      }
      if (B_KFAST) {
        const float2 v = *reinterpret_cast<const float2*>(pb + offb[i]);
        bv[i][0] = v.x;
        bv[i][1] = v.y;
      } else {
        bv[i][0] = pb[offb[i]];
        bv[i][1] = pb[offb[i] + g.Nout];
        if (MODE == kDw && with_act) {
          by[i][0] = pb2[offb[i]];
          by[i][1] = pb2[offb[i] + g.Nout];
        }
      }
    }
    pa += stepa;
    pb += stepb;
    if (MODE == kDx) pa2 += stepa;
    if (MODE == kDw) pb2 += stepb;
  };
  auto stash_fast = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a0 = av[i][0], a1 = av[i][1], b0 = bv[i][0], b1 = bv[i][1];
      if (MODE == kDx && with_act) {
        a0 *= act_grad(ay[i][0], g.act);
        a1 *= act_grad(ay[i][1], g.act);
      }
      if (MODE == kDw && with_act) {
        b0 *= act_grad(by[i][0], g.act);
        b1 *= act_grad(by[i][1], g.act);
      }
      const bool ka = (keepa >> i) & 1, kb = (keepb >> i) & 1;
      const float fill = (MODE == kDw && ((onesa >> i) & 1)) ? 1.f : 0.f;
      const int rowa = A_KFAST ? (tid >> 4) + 16 * i : (tid & 127);
      const int kpa = A_KFAST ? (tid & 15) : (tid >> 7) + 2 * i;
      as[buf][rowa * kPitch + kpa] = pack_bf16(ka ? a0 : fill, ka ? a1 : fill);
      const int rowb = B_KFAST ? (tid >> 4) + 16 * i : (tid & 127);
      const int kpb = B_KFAST ? (tid & 15) : (tid >> 7) + 2 * i;
      bs[buf][rowb * kPitch + kpb] = pack_bf16(kb ? b0 : 0.f, kb ? b1 : 0.f);
    }
  };
  auto issue = [&](int k0) {
    if (FAST) {
      issue_fast();
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = A_KFAST ? (tid >> 4) + 16 * i : (tid & 127);
      const int kp = A_KFAST ? (tid & 15) : (tid >> 7) + 2 * i;
      av[i][0] = a_at<MODE>(gs, m0 + row, k0 + 2 * kp);
      av[i][1] = a_at<MODE>(gs, m0 + row, k0 + 2 * kp + 1);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = B_KFAST ? (tid >> 4) + 16 * i : (tid & 127);
      const int kp = B_KFAST ? (tid & 15) : (tid >> 7) + 2 * i;
      bv[i][0] = b_at<MODE>(gs, k0 + 2 * kp, n0 + row);
      bv[i][1] = b_at<MODE>(gs, k0 + 2 * kp + 1, n0 + row);
    }
  };
  auto stash = [&](int buf) {
    if (FAST) {
      stash_fast(buf);
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = A_KFAST ? (tid >> 4) + 16 * i : (tid & 127);
      const int kp = A_KFAST ? (tid & 15) : (tid >> 7) + 2 * i;
      as[buf][row * kPitch + kp] = pack_bf16(av[i][0], av[i][1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = B_KFAST ? (tid >> 4) + 16 * i : (tid & 127);
      const int kp = B_KFAST ? (tid & 15) : (tid >> 7) + 2 * i;
      bs[buf][row * kPitch + kp] = pack_bf16(bv[i][0], bv[i][1]);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  if (steps > 0) {
    issue(k_lo);
    stash(0);
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const bool more = s + 1 < steps;
    if (more) issue(k_lo + (s + 1) * kBK);
    const uint32_t* A = as[s & 1];
    const uint32_t* Bt = bs[s & 1];
#pragma unroll
    for (int ks = 0; ks < kBK / 16; ++ks) {
      const int col = 8 * ks + 4 * half;
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = as_frag(*reinterpret_cast<const uint4*>(A + (wm + 32 * i + r) * kPitch + col));
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = as_frag(*reinterpret_cast<const uint4*>(Bt + (wn + 32 * j + r) * kPitch + col));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma(a[i], b[j], acc[i][j]);
    }
    if (more) stash((s + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm + 32 * i + tile_row(e, half), n = n0 + wn + 32 * j + r;
        if (g.splits == 1)
          emit<MODE>(g, m, n, acc[i][j][e]);
        else if (m < g.M && n < g.N)
          part[((int64_t)split * g.M + m) * g.N + n] = acc[i][j][e];
      }
}

// ---------------------------------------------------------------- the same product with 16-byte loads
// The kernel above brings its operands in with 4- and 8-byte loads: 24 load instructions per thread and k step,
// and a CU's address unit spends as long on a 256-byte wave load as on a 1 KB one -- the products of configs[4]
// ran at the rate those instructions issue (7 % matrix-core busy).  Here (K and Nout multiples of 4, operands
// 16-byte aligned, the reduction a multiple of 32 per split) every load is 16 bytes:
//   * an operand whose k index is the fast one in memory (x in the forward product, dz and w in the input
//     gradient): a thread takes 4 consecutive k of 4 rows; LDS keeps [row][pairs of k] as above;
//   * an operand whose ROW index is the fast one (w in the forward product, x and dz in the weight gradient): a
//     thread takes 4 consecutive rows of 2 x 2 consecutive k and stores the 4 packed pairs side by side; LDS keeps
//     [pair of k][row] (pitch 136 words: the two halves of a wave read banks 32 apart), and a lane's fragment is 4
//     words from 4 rows of that image instead of one 16-byte read.
constexpr int kPitchT = 136;
#ifndef XDFM_DENSE_REMAP
#define XDFM_DENSE_REMAP 1
#endif

template <int MODE>
__global__ __launch_bounds__(256, 2) void dense_bf16_v4_kernel(GemmArgs g, float* __restrict__ part) {
  __shared__ __align__(16) uint32_t as[2][kBM * kPitch];
  __shared__ __align__(16) uint32_t bs[2][kBN * kPitch];
  constexpr bool A_KFAST = MODE != kDw, B_KFAST = MODE == kDx;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
  // Workgroups go to the 8 XCDs round robin by linear id, and each XCD has its own L2.  With the column
  // tile as the fast grid index, the 8 column tiles that share a row tile's operand land on 8 different
  // XCDs and every XCD pulls the whole row operand through the fabric (counters: half of the L2 requests
  // missed, 3.6 x the operands' bytes fetched).  Remapped: XCD k works on the k-th eighth of the tiles in
  // row-major order, so a row tile's column tiles share one L2.
  int tx = blockIdx.x, ty = blockIdx.y;
  {
    const unsigned tiles = gridDim.x * gridDim.y;
    if (XDFM_DENSE_REMAP && (tiles & 7u) == 0) {
      const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
      const unsigned t = (lin & 7u) * (tiles >> 3) + (lin >> 3);
      ty = (int)(t / gridDim.x);
      tx = (int)(t - (unsigned)ty * gridDim.x);
    }
  }
  const int m0 = ty * kBM, n0 = tx * kBN;
  const int split = blockIdx.z;
  const int k_lo = split * g.r_per_split;
  int k_hi = k_lo + g.r_per_split;
  if (k_hi > g.R) k_hi = g.R;
  const int steps = k_hi > k_lo ? (k_hi - k_lo) / kBK : 0;
  const bool with_act = g.act != XDFM_ACT_IDENTITY;
  const int lda = MODE == kDx ? g.Nout : g.K, ldb = g.Nout;
  const int limA = MODE == kDw ? g.K : g.M;  // rows of A that exist in memory (kDw: row K is the row of ones)
  const int limB = g.N;

  // per-thread sources: k-fast: 4 rows (i), one quad of k; row-fast: 2 pairs of k (i), one quad of rows
  const float* pa[4];
  const float* pa2[4];
  const float* pb[4];
  const float* pb2[4];
  bool ka[4], kb[4];
  bool onesA = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pa[i] = pa2[i] = pb[i] = pb2[i] = nullptr;
    ka[i] = kb[i] = false;
  }
  if (A_KFAST) {
    const float* base = MODE == kDx ? g.dy : g.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + (tid >> 3) + 32 * i;
      ka[i] = row < limA;
      const int64_t off = (int64_t)(ka[i] ? row : limA - 1) * lda + k_lo + 4 * (tid & 7);
      pa[i] = base + off;
      if (MODE == kDx) pa2[i] = g.y + off;
    }
  } else {  // kDw: A(m,k) = x[k][m]
    const int rq = m0 + 4 * (tid & 31);
    ka[0] = ka[1] = rq < limA;
    onesA = rq == limA;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = k_lo + 2 * ((tid >> 5) + 8 * i);
      pa[i] = g.x + (int64_t)kk * lda + (ka[0] ? rq : limA - 4);
    }
  }
  if (B_KFAST) {  // kDx: B(k,n) = w[n][k]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = n0 + (tid >> 3) + 32 * i;
      kb[i] = row < limB;
      pb[i] = g.w + (int64_t)(kb[i] ? row : limB - 1) * ldb + k_lo + 4 * (tid & 7);
    }
  } else {  // kFwd: w[k][n]; kDw: dz[k][n]
    const int rq = n0 + 4 * (tid & 31);
    kb[0] = kb[1] = rq < limB;
    const float* base = MODE == kDw ? g.dy : g.w;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = k_lo + 2 * ((tid >> 5) + 8 * i);
      const int64_t off = (int64_t)kk * ldb + (kb[0] ? rq : limB - 4);
      pb[i] = base + off;
      if (MODE == kDw) pb2[i] = g.y + off;
    }
  }
  const int64_t stepa = A_KFAST ? kBK : (int64_t)kBK * lda;
  const int64_t stepb = B_KFAST ? kBK : (int64_t)kBK * ldb;

  // (loads two k steps ahead through a second register set were tried: 236 - 256 registers with spills, the
  // forward product no faster, the backward ones 40 % slower)
  f4 av[4], ay[4], bv[4], by4[4];
  auto issue = [&]() {
    if (A_KFAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = *reinterpret_cast<const f4*>(pa[i]);
        if (MODE == kDx && with_act) ay[i] = *reinterpret_cast<const f4*>(pa2[i]);
        pa[i] += stepa;
        if (MODE == kDx) pa2[i] += stepa;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[2 * i] = *reinterpret_cast<const f4*>(pa[i]);
        av[2 * i + 1] = *reinterpret_cast<const f4*>(pa[i] + lda);
        pa[i] += stepa;
      }
    }
    if (B_KFAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bv[i] = *reinterpret_cast<const f4*>(pb[i]);
        pb[i] += stepb;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bv[2 * i] = *reinterpret_cast<const f4*>(pb[i]);
        bv[2 * i + 1] = *reinterpret_cast<const f4*>(pb[i] + ldb);
        if (MODE == kDw && with_act) {
          by4[2 * i] = *reinterpret_cast<const f4*>(pb2[i]);
          by4[2 * i + 1] = *reinterpret_cast<const f4*>(pb2[i] + ldb);
        }
        pb[i] += stepb;
        if (MODE == kDw) pb2[i] += stepb;
      }
    }
  };
  auto stash = [&](int buf) {
    if (A_KFAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f4 v = av[i];
        if (MODE == kDx && with_act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= act_grad(ay[i][e], g.act);
        }
        if (!ka[i]) v = f4{0.f, 0.f, 0.f, 0.f};
        const int row = (tid >> 3) + 32 * i;
        *reinterpret_cast<uint2*>(&as[buf][row * kPitch + 2 * (tid & 7)]) =
            make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f4 v0 = av[2 * i], v1 = av[2 * i + 1];
        if (!ka[0]) {
          const float one = onesA ? 1.f : 0.f;
          v0 = f4{one, 0.f, 0.f, 0.f};
          v1 = v0;
        }
        const int kp = (tid >> 5) + 8 * i;
        *reinterpret_cast<uint4*>(&as[buf][kp * kPitchT + 4 * (tid & 31)]) =
            make_uint4(pack_bf16(v0[0], v1[0]), pack_bf16(v0[1], v1[1]), pack_bf16(v0[2], v1[2]), pack_bf16(v0[3], v1[3]));
      }
    }
    if (B_KFAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f4 v = bv[i];
        if (!kb[i]) v = f4{0.f, 0.f, 0.f, 0.f};
        const int row = (tid >> 3) + 32 * i;
        *reinterpret_cast<uint2*>(&bs[buf][row * kPitch + 2 * (tid & 7)]) =
            make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f4 v0 = bv[2 * i], v1 = bv[2 * i + 1];
        if (MODE == kDw && with_act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] *= act_grad(by4[2 * i][e], g.act);
            v1[e] *= act_grad(by4[2 * i + 1][e], g.act);
          }
        }
        if (!kb[0]) {
          v0 = f4{0.f, 0.f, 0.f, 0.f};
          v1 = v0;
        }
        const int kp = (tid >> 5) + 8 * i;
        *reinterpret_cast<uint4*>(&bs[buf][kp * kPitchT + 4 * (tid & 31)]) =
            make_uint4(pack_bf16(v0[0], v1[0]), pack_bf16(v0[1], v1[1]), pack_bf16(v0[2], v1[2]), pack_bf16(v0[3], v1[3]));
      }
    }
  };
  auto frag = [&](const uint32_t* T, bool kfast, int row, int col) {
    if (kfast) return as_frag(*reinterpret_cast<const uint4*>(T + row * kPitch + col));
    uint4 q;
    q.x = T[(col + 0) * kPitchT + row];
    q.y = T[(col + 1) * kPitchT + row];
    q.z = T[(col + 2) * kPitchT + row];
    q.w = T[(col + 3) * kPitchT + row];
    return as_frag(q);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  auto products = [&](int buf) {
    const uint32_t* A = as[buf];
    const uint32_t* Bt = bs[buf];
#pragma unroll
    for (int ks = 0; ks < kBK / 16; ++ks) {
      const int col = 8 * ks + 4 * half;
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = frag(A, A_KFAST, wm + 32 * i + r, col);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = frag(Bt, B_KFAST, wn + 32 * j + r, col);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma(a[i], b[j], acc[i][j]);
    }
  };
  if (steps > 0) {
    issue();
    stash(0);
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) issue();
    products(s & 1);
    if (s + 1 < steps) stash((s + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm + 32 * i + tile_row(e, half), n = n0 + wn + 32 * j + r;
        if (g.splits == 1)
          emit<MODE>(g, m, n, acc[i][j][e]);
        else if (m < g.M && n < g.N)
          part[((int64_t)split * g.M + m) * g.N + n] = acc[i][j][e];
      }
}

template <int MODE>
__global__ void dense_bf16_reduce_kernel(GemmArgs g, const float* __restrict__ part) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)g.M * g.N) return;
  float s = 0.f;
  for (int k = 0; k < g.splits; ++k) s += part[(int64_t)k * g.M * g.N + idx];
  emit<MODE>(g, (int)(idx / g.N), (int)(idx % g.N), s);
}

int splits_for(int M, int N, int R) {
  const int64_t tiles = (int64_t)((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  if (tiles >= 192) return 1;
  int64_t s = (384 + tiles - 1) / tiles;
  const int64_t by_len = R / 256;
  if (s > by_len) s = by_len;
  if (s > 8) s = 8;
  return (int)(s < 1 ? 1 : s);
}

template <int MODE>
int launch(GemmArgs g, float* workspace, hipStream_t st) {
  g.splits = splits_for(g.M, g.N, g.R);
  g.r_per_split = ((g.R + g.splits - 1) / g.splits + kBK - 1) / kBK * kBK;
  XDFM_REQUIRE(g.splits == 1 || workspace, "xdfm_dense_bf16: this product is split along k and needs the workspace");
  const dim3 grid((unsigned)((g.N + kBN - 1) / kBN), (unsigned)((g.M + kBM - 1) / kBM), (unsigned)g.splits);
  const char* name = MODE == kFwd ? "dense_bf16_kernel<fwd>" : (MODE == kDx ? "dense_bf16_kernel<dx>" : "dense_bf16_kernel<dw>");
  const int64_t lim = (int64_t)1 << 31;
  // even leading dimensions where pairs of k are loaded as 8 bytes; R a multiple of 32 (so is every split)
  const bool fast = g.R % kBK == 0 && g.B * (int64_t)g.K < lim && g.B * (int64_t)g.Nout < lim &&
                    (MODE == kFwd ? g.K % 2 == 0 : (MODE == kDx ? g.Nout % 2 == 0 : true));
  static const bool v4_on = !(getenv("XDFM_DENSE_V4") && getenv("XDFM_DENSE_V4")[0] == '0');
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool v4 = v4_on && fast && g.K % 4 == 0 && g.Nout % 4 == 0 && g.K >= 4 && g.Nout >= 4 && al16(g.x) &&
                  al16(g.w) && (MODE == kFwd || (al16(g.y) && al16(g.dy)));
  if (v4)
    XDFM_LAUNCH(MODE == kFwd ? "dense_bf16_v4_kernel<fwd>" : (MODE == kDx ? "dense_bf16_v4_kernel<dx>" : "dense_bf16_v4_kernel<dw>"),
                st, (dense_bf16_v4_kernel<MODE>), grid, dim3(256), 0, g, workspace);
  else if (fast)
    XDFM_LAUNCH(name, st, (dense_bf16_kernel<MODE, true>), grid, dim3(256), 0, g, workspace);
  else
    XDFM_LAUNCH(name, st, (dense_bf16_kernel<MODE, false>), grid, dim3(256), 0, g, workspace);
  if (g.splits > 1) {
    const int64_t total = (int64_t)g.M * g.N;
    XDFM_LAUNCH("dense_bf16_reduce_kernel", st, dense_bf16_reduce_kernel<MODE>, dim3((unsigned)((total + 255) / 256)),
                dim3(256), 0, g, workspace);
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

size_t ws_for(int M, int N, int R) {
  const int s = splits_for(M, N, R);
  return s > 1 ? (size_t)s * M * N : 0;
}

bool sizes_ok(int64_t B, int K, int Nout) {
  return B >= 0 && B < ((int64_t)1 << 31) && K > 0 && Nout > 0 && (int64_t)K * Nout < ((int64_t)1 << 31) &&
         B * (int64_t)(K > Nout ? K : Nout) < ((int64_t)1 << 40);
}

}  // namespace

extern "C" int xdfm_dense_bf16_workspace_floats(int64_t B, int32_t K, int32_t Nout, size_t* floats) {
  XDFM_REQUIRE(floats && sizes_ok(B, K, Nout), "xdfm_dense_bf16_workspace_floats: bad argument");
  size_t a = ws_for((int)B, Nout, K), b = ws_for((int)B, K, Nout), c = ws_for(K + 1, Nout, (int)B);
  *floats = a > b ? (a > c ? a : c) : (b > c ? b : c);
  return XDFM_OK;
}

extern "C" int xdfm_dense_bf16_fwd(const float* x, const float* w, const float* b, int64_t B, int32_t K,
                                   int32_t Nout, int32_t act, float* y, float* workspace, void* stream) {
  XDFM_REQUIRE(x && w && b && y, "xdfm_dense_bf16_fwd: null argument");
  XDFM_REQUIRE(sizes_ok(B, K, Nout), "xdfm_dense_bf16_fwd: bad sizes");
  XDFM_REQUIRE(act >= XDFM_ACT_IDENTITY && act <= XDFM_ACT_ELU, "xdfm_dense_bf16_fwd: unknown activation %d", act);
  if (B == 0) return XDFM_OK;
  GemmArgs g{x, w, b, nullptr, nullptr, y, nullptr, B, K, Nout, act, 0, (int)B, Nout, K, 1, 0};
  return launch<kFwd>(g, workspace, (hipStream_t)stream);
}

extern "C" int xdfm_dense_bf16_bwd(const float* x, const float* w, const float* y, const float* dy, int64_t B,
                                   int32_t K, int32_t Nout, int32_t act, float* dx, int32_t accumulate_dx,
                                   float* dw, float* db, float* workspace, void* stream) {
  XDFM_REQUIRE(x && w && y && dy, "xdfm_dense_bf16_bwd: null argument");
  XDFM_REQUIRE((dw && db) || (!dw && !db && dx), "xdfm_dense_bf16_bwd: dw and db come together; without them dx must be given");
  XDFM_REQUIRE(sizes_ok(B, K, Nout), "xdfm_dense_bf16_bwd: bad sizes");
  XDFM_REQUIRE(act >= XDFM_ACT_IDENTITY && act <= XDFM_ACT_ELU, "xdfm_dense_bf16_bwd: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (dw) {
      XDFM_HIP(hipMemsetAsync(dw, 0, (size_t)K * Nout * sizeof(float), st));
      XDFM_HIP(hipMemsetAsync(db, 0, (size_t)Nout * sizeof(float), st));
    }
    return XDFM_OK;
  }
  if (dw) {
    GemmArgs g{x, w, nullptr, y, dy, dw, db, B, K, Nout, act, 0, K + 1, Nout, (int)B, 1, 0};
    int rc = launch<kDw>(g, workspace, st);
    if (rc) return rc;
  }
  if (dx) {
    GemmArgs g{x, w, nullptr, y, dy, dx, nullptr, B, K, Nout, act, accumulate_dx, (int)B, K, Nout, 1, 0};
    int rc = launch<kDx>(g, workspace, st);
    if (rc) return rc;
  }
  return XDFM_OK;
}
