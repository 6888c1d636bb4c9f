// A CIN layer whose channels are ALL pooled (n_hidden = 0: the last layer, xDeepFM.py:253-270), small
// shapes, as one forward and one backward kernel on the f32 matrix cores.
//
// pooled[b,n] = sum_d sum_{f,h} x0[b,f,d] xk[b,h,d] W[(f,h),n]: with the sum over d taken first,
//     S[b,(f,h)] = sum_d x0[b,f,d] xk[b,h,d],     pooled[b,n] = sum_{(f,h)} S[b,(f,h)] W[(f,h),n]
// a sample costs F*H*(D + L) multiply-adds instead of F*H*D*L, and S never leaves the registers.
// A wave owns a sample at a time.  S^T (H x F) = xk x0^T is a 16x16x4 product per 16 fields whose
// operands are one 16-byte load per lane straight from HBM (lane (g, c) = (lane / 16, lane % 16)
// supplies columns d = 4g..4g+3 of row c: the k index of the instruction is only a label, so step s
// takes d = 4g + s on both operands).  The result leaves lane (g, c) with S[f = 16t + c, h = 4g + r],
// r = 0..3, and the lane keeps the filter rows of exactly those pairs in registers for all its
// samples: pooled is 4 T LN multiply-adds and a wave sum per channel.  Backward, per sample:
//     dS = d_pooled W^T                      (same registers, same layout)
//     dW[(f,h),n] += S d_pooled[n]           (registers, over the wave's samples; then waves, then blocks:
//                                             fixed association, bit-identical reruns)
//     d_x0[f,:] (+)= sum_h dS[f,h] xk[h,:]   (16x16x4 products whose A operand IS the dS registers)
//     d_xk[h,:]  = sum_f dS[f,h] x0[f,:]     (dS through LDS into the A layout; x0 rows as B)
// No LDS traffic but the dS transpose; on BASELINE configs[2] (39 x 10 pairs, 10 channels) this layer
// takes about 8 + 16 us instead of the 26 + 37 + 40 + 10 us of the per-(sample, d) kernels.
// The same sums in another association: 1e-6 relative.  Served: F <= 64, H <= 16, D <= 16 and a
// multiple of 4, L <= 16, 4 ceil(F/16) L' <= 144
// (L' = L rounded up to 4, 8, 10, 12 or 16), unmasked (the engine keeps the general kernels for everything else).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "xdfm_common.h"
#include "stamp.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kSpwFwd = 4;  // samples per wave: 2 waves per SIMD at B = 8192
constexpr int kSpwBwd = 8;  // one wave per SIMD (W and dW are 8 T LN registers), one block per CU

struct PoolDims {
  int64_t B;
  int F, H, D, L, ldw;
};

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// The waves of a block share nothing while they walk their samples (each has its own LDS rows), so
// the order between a wave's own LDS stores and loads is all that is needed: the DS unit serves one
// wave's requests in order, and the fences keep the compiler from moving them.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v + the value of the lane a DPP pattern names (one v_add with a DPP operand: no LDS round trip)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// The sum of v over the wave, in every lane: lanes 1 and 2 apart within quads, then the mirrored
// half and whole row of 16, then the four rows through scalar registers.  Fixed association.
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// The filter, from [P][L] in HBM to each lane's registers wr[t][r][n] = W[(f = 16t + c, h = 4g + r), n]
// (0 outside F x H x L), through an LDS image in register order, [(t, r, n)][lane] with rows of 65
// floats: a thread takes a pair's row (L loads in flight, requested by filter_fetch ahead of the
// samples' operands, which return after them) and scatters it down a column of the image, one bank
// per channel; the lanes then read the image row by row without a bank conflict.
constexpr int kImgPitch = 65;

template <int T, int LN>
__device__ __forceinline__ void filter_fetch(const PoolDims& c, const float* __restrict__ w, float (&v)[T][LN]) {
  const int P = c.F * c.H;  // <= 256 T
#pragma unroll
  for (int u = 0; u < T; ++u) {
    const int p = threadIdx.x + 256 * u;
    const float* row = w + (int64_t)(p < P ? p : P - 1) * c.ldw;
#pragma unroll
    for (int n = 0; n < LN; ++n) v[u][n] = row[n < c.L ? n : 0];
  }
}

template <int T, int LN>
__device__ __forceinline__ void filter_stage(const PoolDims& c, const float (&v)[T][LN], float* __restrict__ img,
                                             float (&wr)[T][4][LN]) {
  const int P = c.F * c.H;
#pragma unroll
  for (int u = 0; u < T; ++u) {
    const int p = threadIdx.x + 256 * u;
    if (p < P) {
      const int f = p / c.H, h = p - f * c.H;
      float* col = img + ((4 * (f >> 4) + (h & 3)) * LN) * kImgPitch + 16 * (h >> 2) + (f & 15);
#pragma unroll
      for (int n = 0; n < LN; ++n) col[n * kImgPitch] = n < c.L ? v[u][n] : 0.f;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, lc = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (an AND, not a select: a select made the compiler branch around each load and wait for it
      // there, 60 LDS latencies in a row; slots of pairs outside F x H were never written)
      const unsigned keep = (16 * t + lc < c.F && 4 * g + r < c.H) ? 0xffffffffu : 0u;
#pragma unroll
      for (int n = 0; n < LN; ++n) {
        const unsigned bits = __builtin_bit_cast(unsigned, img[((4 * t + r) * LN + n) * kImgPitch + lane]);
        wr[t][r][n] = __builtin_bit_cast(float, bits & keep);
      }
    }
  __syncthreads();  // the LDS is reused below
}

// Offsets (floats, within a sample) of this lane's 16-byte operand loads: x0 row min(16t + c, F-1)
// and xk row min(c, H-1), columns 4g..4g+3 (column group clamped into the row; its values are
// replaced by 0 where 4g >= D).
template <int T>
__device__ __forceinline__ void operand_offsets(const PoolDims& c, int (&oa)[T], int& ob, bool& live_cols) {
  const int lane = threadIdx.x & 63, lc = lane & 15, g = lane >> 4;
  live_cols = 4 * g < c.D;
  const int col = live_cols ? 4 * g : 0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int f = 16 * t + lc;
    oa[t] = (f < c.F ? f : c.F - 1) * c.D + col;
  }
  ob = (lc < c.H ? lc : c.H - 1) * c.D + col;
}

__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// S^T tiles of one sample from its operands
template <int T>
__device__ __forceinline__ void pair_sums(const f32x4 (&xa)[T], const f32x4& xb, bool live_cols, f32x4 (&S)[T]) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 b = live_cols ? xb : zero;
#pragma unroll
  for (int t = 0; t < T; ++t) S[t] = zero;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < T; ++t) S[t] = mfma(b[s], xa[t][s], S[t]);
}

// HEAD: the rest of the forward pass in the same launch (xdfm_head_loss's arithmetic, one sample per wave
// pass instead of one per thread): the CIN head over the pooled channels of ALL layers (the other layers'
// were written by their kernels before), the logit, the prediction, the sample's loss term, d loss / d logit
// and d_pooled.  The mean of the loss is left as one partial sum per block (xdfm_loss_finish_partials).
struct PoolHead {
  const float *lin, *fm, *dnn, *bias;  // parts of the logit beside the CIN's (null: absent); bias goes with lin
  const float *head_w, *head_b, *labels;
  int n_total;  // pooled channels of all layers = row length of pooled / d_pooled (<= 64)
  float inv_count;
  int classification, loss_kind;
  float *cin_out, *logit, *pred, *g_logit, *d_pooled, *partial;
};

constexpr float kLogEps = 1e-7f;  // tf.compat.v1.losses.log_loss default epsilon

__device__ __forceinline__ float sigmoidf(float z) {
  // same split as Eigen's logistic: no overflow for large |z|
  if (z >= 0.f) return 1.f / (1.f + expf(-z));
  const float e = expf(z);
  return e / (1.f + e);
}

template <int T, int LN, bool HEAD>
__global__ __launch_bounds__(256, (T * LN <= 30 ? 2 : 1)) void cin_pooled_fwd_kernel(
    PoolDims c, const float* __restrict__ x0, const float* __restrict__ xk, const float* __restrict__ w,
    float* __restrict__ pooled, int pool_stride, int pool_off, PoolHead hd) {
  xdfm::Stamp stamp_(xdfm::kStampPooledFwd);
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float loss_acc = 0.f;  // HEAD: this wave's loss terms, in sample order
  const int FD = c.F * c.D, HD = c.H * c.D;
  int oa[T], ob;
  bool live_cols;
  operand_offsets<T>(c, oa, ob, live_cols);
  const int64_t first = ((int64_t)blockIdx.x * 4 + wave) * kSpwFwd;
  float wv[T][LN];
  filter_fetch<T, LN>(c, w, wv);
  // the operands of all the wave's samples, in sample order (clamped: a sample past the end repeats
  // the last one and stores nothing)
  f32x4 xa[kSpwFwd][T], xb[kSpwFwd];
#pragma unroll
  for (int s = 0; s < kSpwFwd; ++s) {
    const int64_t b = first + s < c.B ? first + s : c.B - 1;
#pragma unroll
    for (int t = 0; t < T; ++t) xa[s][t] = load4(x0 + b * FD + oa[t]);
    xb[s] = load4(xk + b * HD + ob);
  }
  // HEAD: what the head of every sample needs from memory, requested with the operands (a load per
  // sample inside the loop below would add its latency four times over)
  float h_other[kSpwFwd], h_lin[kSpwFwd], h_fm[kSpwFwd], h_dnn[kSpwFwd], h_y[kSpwFwd];
  float h_w = 0.f, h_b0 = 0.f, h_bias = 0.f;
  if constexpr (HEAD) {
    const int ch = lane < hd.n_total ? lane : 0;
    h_w = lane < hd.n_total ? hd.head_w[ch] : 0.f;
    h_b0 = hd.head_b[0];
    h_bias = hd.bias ? hd.bias[0] : 0.f;
#pragma unroll
    for (int s = 0; s < kSpwFwd; ++s) {
      const int64_t b = first + s < c.B ? first + s : c.B - 1;
      h_other[s] = pooled[b * pool_stride + ch];  // (this layer's channels: replaced below)
      h_lin[s] = hd.lin ? hd.lin[b] : 0.f;
      h_fm[s] = hd.fm ? hd.fm[b] : 0.f;
      h_dnn[s] = hd.dnn ? hd.dnn[b] : 0.f;
      h_y[s] = hd.labels[b];
    }
  }
  float wr[T][4][LN];
  filter_stage<T, LN>(c, wv, lds, wr);
#pragma unroll
  for (int s = 0; s < kSpwFwd; ++s) {
    const int64_t b = first + s;
    f32x4 S[T];
    pair_sums<T>(xa[s], xb[s], live_cols, S);
    float acc[LN];
#pragma unroll
    for (int n = 0; n < LN; ++n) acc[n] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < LN; ++n) acc[n] = fmaf(S[t][r], wr[t][r][n], acc[n]);
    float mine = 0.f;
#pragma unroll
    for (int n = 0; n < LN; ++n) {
      const float v = wave_sum(acc[n]);
      if (lane == n) mine = v;
    }
    if (b < c.B && lane < c.L) pooled[b * pool_stride + pool_off + lane] = mine;
    if constexpr (HEAD) {
      // lane j < n_total holds pooled channel j of the sample: this layer's from the sums above
      const int src = lane - pool_off;
      const float own = __shfl(mine, src < 0 ? 0 : (src > 63 ? 63 : src));
      if (b < c.B) {  // (wave-uniform)
        const float wj = h_w;
        float r = 0.f;
        if (lane < hd.n_total) r = (src >= 0 && src < c.L) ? own : h_other[s];
        const float base = wave_sum(r), dot = wave_sum(r * wj);
        const float cin = base + (dot + h_b0);
        float z = 0.f;
        bool first_part = true;
        if (hd.lin) {
          z = h_lin[s] + h_bias;
          first_part = false;
        }
        if (hd.fm) {
          z = first_part ? h_fm[s] : z + h_fm[s];
          first_part = false;
        }
        z = first_part ? cin : z + cin;
        if (hd.dnn) z = z + h_dnn[s];
        const float p = hd.classification ? sigmoidf(z) : z;
        const float y = h_y[s];
        const float dp_dz = hd.classification ? p * (1.f - p) : 1.f;
        float g;
        if (hd.loss_kind == XDFM_LOSS_LOG) {
          loss_acc += -y * logf(p + kLogEps) - (1.f - y) * logf(1.f - p + kLogEps);
          g = (-y / (p + kLogEps) + (1.f - y) / (1.f - p + kLogEps)) * dp_dz * hd.inv_count;
        } else {  // XDFM_LOSS_CROSS_ENTROPY
          loss_acc += fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
          g = (sigmoidf(z) - y) * hd.inv_count;
        }
        if (lane == 0) {
          hd.cin_out[b] = cin;
          hd.logit[b] = z;
          hd.pred[b] = p;
          hd.g_logit[b] = g;
        }
        if (lane < hd.n_total) hd.d_pooled[b * hd.n_total + lane] = g + g * wj;
      }
    }
  }
  if constexpr (HEAD) {
    // the block's share of the loss sum: its waves in wave order (the filter image in LDS is dead)
    __syncthreads();
    if (lane == 0) lds[wave] = loss_acc;
    __syncthreads();
    if (threadIdx.x == 0) hd.partial[blockIdx.x] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
  }
}

template <int T, int LN>
__global__ __launch_bounds__(256) void cin_pooled_bwd_kernel(PoolDims c, const float* __restrict__ x0,
                                                             const float* __restrict__ xk, const float* __restrict__ w,
                                                             const float* __restrict__ d_pooled, int pool_stride,
                                                             int pool_off, float* __restrict__ d_xk,
                                                             float* __restrict__ d_x0, int accumulate_x0,
                                                             float* __restrict__ part) {
  xdfm::Stamp stamp_(xdfm::kStampPooledBwd);
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane & 15, g = lane >> 4;
  const int FD = c.F * c.D, HD = c.H * c.D;
  int oa[T], ob;
  bool live_cols;
  operand_offsets<T>(c, oa, ob, live_cols);
  // second-layout offsets: one float per lane, column min(c, D-1)
  const int col = lc < c.D ? lc : c.D - 1;
  int ok2[4], o02[4 * T], oo[T][4];
  bool keep0[T][4], keepk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int h = 4 * g + r;
    ok2[r] = (h < c.H ? h : c.H - 1) * c.D + col;
    keepk[r] = h < c.H && lc < c.D;
  }
#pragma unroll
  for (int st = 0; st < 4 * T; ++st) {
    const int f = 4 * st + g;
    o02[st] = (f < c.F ? f : c.F - 1) * c.D + col;
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * g + r;
      oo[t][r] = (f < c.F ? f : c.F - 1) * c.D + col;
      keep0[t][r] = f < c.F && lc < c.D;
    }
  float wv[T][LN];
  filter_fetch<T, LN>(c, w, wv);
  const int64_t first = ((int64_t)blockIdx.x * 4 + wave) * kSpwBwd;
  const int64_t b0 = first < c.B ? first : c.B - 1;
  f32x4 xa[T], xb;
#pragma unroll
  for (int t = 0; t < T; ++t) xa[t] = load4(x0 + b0 * FD + oa[t]);
  xb = load4(xk + b0 * HD + ob);
  float rdp = d_pooled[b0 * pool_stride + pool_off + (lane < c.L ? lane : 0)];
  float wr[T][4][LN], dw[T][4][LN];
  filter_stage<T, LN>(c, wv, lds, wr);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < LN; ++n) dw[t][r][n] = 0.f;
  float* dsl = lds + wave * (16 * T * 16);  // [16 T][16]: dS[f][h] of the wave's sample
  for (int s = 0; s < kSpwBwd; ++s) {
    const int64_t b = first + s;
    if (b >= c.B) break;
    // everything this sample still needs from HBM, requested ahead of the products
    const float* x0b = x0 + b * FD;
    const float* xkb = xk + b * HD;
    float* dx0b = d_x0 + b * FD;
    float xk2[4], x02[4 * T];
    f32x4 old[T];
#pragma unroll
    for (int r = 0; r < 4; ++r) xk2[r] = xkb[ok2[r]];
#pragma unroll
    for (int st = 0; st < 4 * T; ++st) x02[st] = x0b[o02[st]];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) old[t][r] = dx0b[oo[t][r]];
    f32x4 S[T];
    pair_sums<T>(xa, xb, live_cols, S);
    float dp[LN];
    {
      const float mine = lane < c.L ? rdp : 0.f;
#pragma unroll
      for (int n = 0; n < LN; ++n) dp[n] = lane_value(mine, n);
    }
    // the next sample's first-layout operands
    const int64_t bn = (s + 1 < kSpwBwd && b + 1 < c.B) ? b + 1 : b;
#pragma unroll
    for (int t = 0; t < T; ++t) xa[t] = load4(x0 + bn * FD + oa[t]);
    xb = load4(xk + bn * HD + ob);
    rdp = d_pooled[bn * pool_stride + pool_off + (lane < c.L ? lane : 0)];
    f32x4 dS[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < LN; ++n) {
          v = fmaf(dp[n], wr[t][r][n], v);
          dw[t][r][n] = fmaf(S[t][r], dp[n], dw[t][r][n]);
        }
        dS[t][r] = v;
      }
      *reinterpret_cast<f32x4*>(dsl + (16 * t + lc) * 16 + 4 * g) = dS[t];
    }
    wave_sync();
    // d_x0 rows 16t + 4g + r', column c
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = accumulate_x0 ? old[t][r] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = mfma(dS[t][r], xk2[r], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (keep0[t][r]) dx0b[oo[t][r]] = acc[r];
    }
    // d_xk rows 4g + r, column c
    {
      // four chains (a product waits 40 cycles for the one it adds to)
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      f32x4 part4[4] = {zero, zero, zero, zero};
#pragma unroll
      for (int st = 0; st < 4 * T; ++st) part4[st & 3] = mfma(dsl[(4 * st + g) * 16 + lc], x02[st], part4[st & 3]);
      const f32x4 acc = (part4[0] + part4[1]) + (part4[2] + part4[3]);
      float* dxkb = d_xk + b * HD;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (keepk[r]) dxkb[ok2[r]] = acc[r];
    }
    wave_sync();
  }
  // The block's filter-gradient partial, in register order [(t, r, n)][lane]: waves 1 and 3 lay theirs
  // down, waves 0 and 2 add them, wave 2 lays its sum down, wave 0 adds it and writes.
  constexpr int kN = 4 * T * LN;
  __syncthreads();
  float* slot = lds + (wave >> 1) * kN * 64 + lane;
  if (wave & 1) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < LN; ++n) slot[((t * 4 + r) * LN + n) * 64] = dw[t][r][n];
  }
  __syncthreads();
  if (!(wave & 1)) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < LN; ++n) dw[t][r][n] += slot[((t * 4 + r) * LN + n) * 64];
  }
  __syncthreads();
  if (wave == 2) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < LN; ++n) lds[((t * 4 + r) * LN + n) * 64 + lane] = dw[t][r][n];
  }
  __syncthreads();
  if (wave == 0) {
    float* dst = part + (size_t)blockIdx.x * kN * 64 + lane;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < LN; ++n) {
          const int i = (t * 4 + r) * LN + n;
          dst[i * 64] = dw[t][r][n] + lds[i * 64 + lane];
        }
  }
}

// d_w = the sum of the blocks' partials: a block per register slot i = (t, r, n) (64 lanes = 64 pairs),
// the partials dealt to 16 waves as consecutive ranges (8 loads in flight each), the waves' sums then
// added in wave order.
__global__ __launch_bounds__(1024) void cin_pooled_dw_reduce_kernel(PoolDims c, const float* __restrict__ part,
                                                                    int n_blocks, int LN, int kN,
                                                                    float* __restrict__ d_w) {
  xdfm::Stamp stamp_(xdfm::kStampPooledDwReduce);
  __shared__ float sums[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x;
  const int per = (n_blocks + 15) / 16;
  const int k_beg = wave * per, k_end = (k_beg + per < n_blocks) ? k_beg + per : n_blocks;
  float s = 0.f;
  for (int k0 = k_beg; k0 < k_end; k0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(k0 + u < k_end ? k0 + u : k0) * kN + i) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (k0 + u < k_end) ? v[u] : 0.f;
  }
  sums[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    const int n = i % LN, tr = i / LN, t = tr >> 2, r = tr & 3;
    const int f = 16 * t + (lane & 15), h = 4 * (lane >> 4) + r;
    if (f < c.F && h < c.H && n < c.L) {
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 16; ++wv) v += sums[wv][lane];
      d_w[(int64_t)(f * c.H + h) * c.ldw + n] = v;
    }
  }
}

int tiles_for(int F) { return (F + 15) / 16; }
int ln_for(int L) { return L <= 4 ? 4 : L <= 8 ? 8 : L <= 10 ? 10 : L <= 12 ? 12 : 16; }

// a lane keeps 4 T LN filter values and as many gradient sums: beyond 144 the backward kernel spills
constexpr int kMaxRegs = 144;

bool supported(int F, int H, int D, int L) {
  return F > 0 && F <= 64 && H > 0 && H <= 16 && D > 0 && D <= 16 && (D & 3) == 0 && L > 0 && L <= 16 &&
         4 * tiles_for(F) * ln_for(L) <= kMaxRegs;
}

PoolDims make(int64_t B, int F, int H, int D, int L, int ldw) {
  PoolDims c;
  c.B = B;
  c.F = F;
  c.H = H;
  c.D = D;
  c.L = L;
  c.ldw = ldw ? ldw : L;
  return c;
}

size_t lds_bytes(const PoolDims& c, bool bwd) {
  const int T = tiles_for(c.F), LN = ln_for(c.L);
  size_t floats = (size_t)(4 * T * LN) * kImgPitch;  // the filter's image
  if (bwd) {
    const size_t rows = 4 * (size_t)(16 * T * 16), slots = 2 * (size_t)(4 * T * LN) * 64;
    if (floats < rows) floats = rows;
    if (floats < slots) floats = slots;
  }
  return floats * sizeof(float);
}

// kernels by (tiles of 16 fields, channels): a lane's registers hold 4 T LN filter values (and, backward,
// as many gradient sums)
template <int T, int LN>
const void* kernel_of(bool bwd, bool head) {
  if constexpr (4 * T * LN > kMaxRegs) return nullptr;  // not served (supported())
  else return bwd ? reinterpret_cast<const void*>(cin_pooled_bwd_kernel<T, LN>)
                  : (head ? reinterpret_cast<const void*>(cin_pooled_fwd_kernel<T, LN, true>)
                          : reinterpret_cast<const void*>(cin_pooled_fwd_kernel<T, LN, false>));
}
template <int T>
const void* pick_l(int LN, bool bwd, bool head) {
  switch (LN) {
    case 4: return kernel_of<T, 4>(bwd, head);
    case 8: return kernel_of<T, 8>(bwd, head);
    case 10: return kernel_of<T, 10>(bwd, head);
    case 12: return kernel_of<T, 12>(bwd, head);
    default: return kernel_of<T, 16>(bwd, head);
  }
}
const void* pick(int F, int L, bool bwd, bool head = false) {
  switch (tiles_for(F)) {
    case 1: return pick_l<1>(ln_for(L), bwd, head);
    case 2: return pick_l<2>(ln_for(L), bwd, head);
    case 3: return pick_l<3>(ln_for(L), bwd, head);
    default: return pick_l<4>(ln_for(L), bwd, head);
  }
}

unsigned n_blocks(int64_t B, int spw) { return (unsigned)((B + 4 * spw - 1) / (4 * spw)); }

}  // namespace

extern "C" int xdfm_cin_pooled_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                                size_t* floats) {
  XDFM_REQUIRE(floats && B >= 0, "xdfm_cin_pooled_workspace_floats: bad argument");
  *floats = 0;
  if (!supported(F, H, D, L)) return XDFM_OK;
  const unsigned blocks = n_blocks(B, kSpwBwd);
  *floats = (size_t)(blocks > 0 ? blocks : 1) * 4 * tiles_for(F) * ln_for(L) * 64;
  return XDFM_OK;
}

namespace {
int launch_fwd(const char* who, const float* x0, const float* xk, const float* w, int64_t B, int F, int H, int D,
               int L, int ldw, float* pooled, int pool_stride, int pool_off, const PoolHead* head, void* stream) {
  XDFM_REQUIRE(supported(F, H, D, L), "%s: shape not served (xdfm_cin_pooled_workspace_floats says 0)", who);
  XDFM_REQUIRE(B >= 0, "%s: bad sizes", who);
  if (B == 0) return XDFM_OK;
  XDFM_REQUIRE(x0 && xk && w && pooled, "%s: null argument", who);
  const PoolDims c = make(B, F, H, D, L, ldw);
  const size_t lds = lds_bytes(c, false);
  const void* fn = pick(F, L, false, head != nullptr);
  if (lds > 64 * 1024) XDFM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  PoolHead hd;
  memset(&hd, 0, sizeof(hd));
  if (head) hd = *head;
  {
    xdfm::ProfScope prof(head ? "cin_pooled_fwd_head_kernel" : "cin_pooled_fwd_kernel", stream);
    void* args[] = {(void*)&c,      (void*)&x0,          (void*)&xk,       (void*)&w,
                    (void*)&pooled, (void*)&pool_stride, (void*)&pool_off, (void*)&hd};
    XDFM_HIP(hipLaunchKernel(fn, dim3(n_blocks(B, kSpwFwd)), dim3(256), args, lds, (hipStream_t)stream));
  }
  return XDFM_OK;
}
}  // namespace

extern "C" int xdfm_cin_pooled_fwd(const float* x0, const float* xk, const float* w, int64_t B, int32_t F,
                                   int32_t H, int32_t D, int32_t L, int32_t ldw, float* pooled,
                                   int32_t pool_stride, int32_t pool_off, void* stream) {
  return launch_fwd("xdfm_cin_pooled_fwd", x0, xk, w, B, F, H, D, L, ldw, pooled, pool_stride, pool_off, nullptr,
                    stream);
}

extern "C" int xdfm_cin_pooled_fwd_head(const float* x0, const float* xk, const float* w, int64_t B, int32_t F,
                                        int32_t H, int32_t D, int32_t L, int32_t ldw, float* pooled,
                                        int32_t pool_off, const float* part0, const float* part1,
                                        const float* part3, const float* bias, const float* head_w,
                                        const float* head_b, int32_t n, const float* labels, int64_t mean_count,
                                        int32_t classification, int32_t loss_kind, float* logit, float* pred,
                                        float* cin_out, float* g_logit, float* d_pooled, float* scratch,
                                        int32_t* n_partial, void* stream) {
  XDFM_REQUIRE(head_w && head_b && labels && logit && pred && cin_out && g_logit && d_pooled && scratch && n_partial,
               "xdfm_cin_pooled_fwd_head: null argument");
  XDFM_REQUIRE(n > 0 && n <= 64 && pool_off >= 0 && pool_off + L <= n, "xdfm_cin_pooled_fwd_head: bad channel counts");
  XDFM_REQUIRE(loss_kind == XDFM_LOSS_LOG || loss_kind == XDFM_LOSS_CROSS_ENTROPY,
               "xdfm_cin_pooled_fwd_head: log_loss or cross_entropy_loss (square_loss needs the mean first)");
  XDFM_REQUIRE(B > 0 && (mean_count == 0 || mean_count >= B), "xdfm_cin_pooled_fwd_head: bad batch / mean_count");
  XDFM_REQUIRE(!bias || part0, "xdfm_cin_pooled_fwd_head: bias belongs to part0 (the linear part)");
  const unsigned blocks = n_blocks(B, kSpwFwd);
  XDFM_REQUIRE(blocks <= 1024, "xdfm_cin_pooled_fwd_head: at most %d samples per call", 1024 * 4 * kSpwFwd);
  PoolHead hd;
  hd.lin = part0;
  hd.fm = part1;
  hd.dnn = part3;
  hd.bias = bias;
  hd.head_w = head_w;
  hd.head_b = head_b;
  hd.labels = labels;
  hd.n_total = n;
  hd.inv_count = 1.f / (float)(mean_count > 0 ? mean_count : B);
  hd.classification = classification;
  hd.loss_kind = loss_kind;
  hd.cin_out = cin_out;
  hd.logit = logit;
  hd.pred = pred;
  hd.g_logit = g_logit;
  hd.d_pooled = d_pooled;
  hd.partial = scratch;
  *n_partial = (int32_t)blocks;
  return launch_fwd("xdfm_cin_pooled_fwd_head", x0, xk, w, B, F, H, D, L, ldw, pooled, n, pool_off, &hd, stream);
}

extern "C" int xdfm_cin_pooled_bwd(const float* x0, const float* xk, const float* w, const float* d_pooled,
                                   int32_t pool_stride, int32_t pool_off, int64_t B, int32_t F, int32_t H, int32_t D,
                                   int32_t L, int32_t ldw, float* d_xk, float* d_x0, int32_t accumulate_x0, float* d_w,
                                   float* workspace, void* stream) {
  XDFM_REQUIRE(supported(F, H, D, L),
               "xdfm_cin_pooled_bwd: shape not served (xdfm_cin_pooled_workspace_floats says 0)");
  XDFM_REQUIRE(B >= 0, "xdfm_cin_pooled_bwd: bad sizes");
  XDFM_REQUIRE(workspace && (d_w || x0), "xdfm_cin_pooled_bwd: null d_w / workspace");
  const PoolDims c = make(B, F, H, D, L, ldw);
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = n_blocks(B, kSpwBwd);
  const int LN = ln_for(L), kN = 4 * tiles_for(F) * LN;
  if (B > 0 && x0) {  // x0 == NULL: only the reduction of the partial sums an earlier call left
    XDFM_REQUIRE(xk && w && d_pooled && d_x0 && d_xk, "xdfm_cin_pooled_bwd: null argument");
    const size_t lds = lds_bytes(c, true);
    const void* fn = pick(F, L, true);
    if (lds > 64 * 1024) XDFM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    xdfm::ProfScope prof("cin_pooled_bwd_kernel", stream);
    void* args[] = {(void*)&c,        (void*)&x0,   (void*)&xk,   (void*)&w,           (void*)&d_pooled, (void*)&pool_stride,
                    (void*)&pool_off, (void*)&d_xk, (void*)&d_x0, (void*)&accumulate_x0, (void*)&workspace};
    XDFM_HIP(hipLaunchKernel(fn, dim3(blocks), dim3(256), args, lds, st));
  }
  if (d_w)  // d_w == NULL: the filter gradient stays as per-block partial sums in the workspace
    XDFM_LAUNCH("cin_pooled_dw_reduce_kernel", st, cin_pooled_dw_reduce_kernel, dim3((unsigned)kN), dim3(1024), 0, c,
                workspace, (int)blocks, LN, kN, d_w);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

XDFM_STAMP_UNIT(stamp_unit_pooled)
