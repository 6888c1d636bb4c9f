// Compressed Interaction Network (CIN), one layer forward / backward, fp32, gfx950.
//
// Reference: recommenders/models/deeprec/models/xDeepFM.py:191-267.  The reference
// materialises Z[b,d,f*H+h] = X0[b,f,d]*Xk[b,h,d] (:192-200), masks layer 0 (:208-217) and
// runs a width-1 conv1d, i.e. a GEMM over f*H+h (:220-222).  Here Z is never stored:
//
//  * forward / input gradients: one lane per m = (sample, d).  The lane keeps its L output
//    channels (or upstream gradients) in registers and walks the (f,h) pairs; the filter
//    row W[f*H+h, :] is the same for every lane, so it is fetched with scalar loads and fed
//    to v_fma as an SGPR operand.  fp32 MFMA runs at the fp32 VALU rate on gfx950 and would
//    pad L=20/10 up to 32, so the VALU form is the faster fp32 one; X0/Xk tiles are staged
//    through LDS transposed to [h][lane] (conflict-free reads).
//  * filter gradient: one lane per (f,h) pair, registers hold dW[pair, 0..L); the batch is
//    split over blocks, the per-split partial sums are added in split order by a second
//    kernel (deterministic, no float atomics).
//
// Layouts: x0[B][F][D], xk[B][H][D], w[F*H][L], xnext[B][n_hidden][D], pooled[B][stride].
#include <hip/hip_runtime.h>

#include "act.h"
#include "cin_common.h"
#include "xdfm_common.h"

using xdfm_cin::CinDims;

namespace {

constexpr int kThreads = 256;
constexpr int kPad = 16;  // LDS row = blockDim + kPad floats: staging writes of 2 h-rows hit disjoint banks
constexpr int kMaxSplits = 256;

// stage src[b0 .. b0+nb)[R][D] into lds[r * Tp + bl*D + d]
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, int64_t b0, int nb, int R,
                                           int D, float* lds, int Tp) {
  const int total = nb * R * D;
  const float* base = src + b0 * R * D;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int bl = idx / (R * D);
    const int rem = idx - bl * R * D;
    const int r = rem / D;
    const int d = rem - r * D;
    lds[r * Tp + bl * D + d] = base[idx];
  }
}

template <int NCH, bool FULL>
__device__ __forceinline__ void fwd_accumulate(const CinDims& c, const float* __restrict__ w,
                                               const float* xs, const float* as, int Tp, int c0,
                                               float (&out)[NCH]) {
  const int tid = threadIdx.x;
  for (int f = 0; f < c.F; ++f) {
    const float af = as[f * Tp + tid];
    for (int h = c.masked ? f + 1 : 0; h < c.H; ++h) {
      const float t = af * xs[h * Tp + tid];
      const float* __restrict__ wrow = w + (int64_t)(f * c.H + h) * cin_ldw(c) + c0;
#pragma unroll
      for (int n = 0; n < NCH; ++n)
        if (FULL || c0 + n < c.L) out[n] = fmaf(t, wrow[n], out[n]);
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(kThreads) void cin_fwd_kernel(
    CinDims c, const float* __restrict__ x0, const float* __restrict__ xk,
    const float* __restrict__ w, float* __restrict__ xnext, float* __restrict__ pooled,
    int pool_stride, int pool_off, int SPB) {
  extern __shared__ float lds[];
  const int Tp = blockDim.x + kPad;
  float* xs = lds;
  const bool shared_x = (xk == x0);
  float* as = shared_x ? xs : lds + c.H * Tp;
  const int64_t b0 = (int64_t)blockIdx.x * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  stage_tile(xk, b0, nb, c.H, c.D, xs, Tp);
  if (!shared_x) stage_tile(x0, b0, nb, c.F, c.D, as, Tp);
  __syncthreads();

  const int tid = threadIdx.x;
  const bool on = tid < nb * c.D;
  const int c0 = blockIdx.y * NCH;
  float out[NCH];
#pragma unroll
  for (int n = 0; n < NCH; ++n) out[n] = 0.f;
  if (on) {
    if (c0 + NCH <= c.L)
      fwd_accumulate<NCH, true>(c, w, xs, as, Tp, c0, out);
    else
      fwd_accumulate<NCH, false>(c, w, xs, as, Tp, c0, out);
  }
  const float scale = c.masked ? 2.f : 1.f;  // exact, so applying it last changes no bit
  const int bl = tid / c.D;
  const int d = tid - bl * c.D;
  const int64_t b = b0 + bl;

  __syncthreads();  // everyone is done with xs/as: reuse the LDS for the pooled channels
#pragma unroll
  for (int n = 0; n < NCH; ++n) {
    const int ch = c0 + n;
    const float v = out[n] * scale;
    if (ch < c.L) {
      if (ch < c.n_hidden) {
        if (on) xnext[(b * c.n_hidden + ch) * c.D + d] = v;
      } else {
        lds[n * Tp + tid] = on ? v : 0.f;
      }
    }
  }
  __syncthreads();
  if (on && d == 0) {
#pragma unroll
    for (int n = 0; n < NCH; ++n) {
      const int ch = c0 + n;
      if (ch < c.L && ch >= c.n_hidden) {
        float s = 0.f;
        for (int dd = 0; dd < c.D; ++dd) s += lds[n * Tp + tid + dd];  // reduce_sum over D (:270)
        pooled[b * pool_stride + pool_off + ch - c.n_hidden] = s;
      }
    }
  }
}

// upstream gradient of curr in [m][L] order (m = b*D+d): the layout the scalar loads want
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

template <int NCH, bool FULL>
__device__ __forceinline__ void dx_accumulate(const CinDims& c, const float* __restrict__ w,
                                              const float* xs, const float* as, float* dxs, int Tp,
                                              int c0, const float (&dc)[NCH], float scale,
                                              float* __restrict__ d_x0_col /* + f*D per f */) {
  const int tid = threadIdx.x;
  for (int f = 0; f < c.F; ++f) {
    const float af = as[f * Tp + tid];
    float accf = 0.f;
    for (int h = c.masked ? f + 1 : 0; h < c.H; ++h) {
      const float* __restrict__ wrow = w + (int64_t)(f * c.H + h) * cin_ldw(c) + c0;
      float g = 0.f;
#pragma unroll
      for (int n = 0; n < NCH; ++n)
        if (FULL || c0 + n < c.L) g = fmaf(dc[n], wrow[n], g);
      accf = fmaf(xs[h * Tp + tid], g, accf);
      dxs[h * Tp + tid] = fmaf(af, g, dxs[h * Tp + tid]);  // own column: no other lane touches it
    }
    d_x0_col[f * c.D] += accf * scale;
  }
}

template <int NCH>
__global__ __launch_bounds__(kThreads) void cin_bwd_dx_kernel(
    CinDims c, const float* __restrict__ x0, const float* __restrict__ xk,
    const float* __restrict__ w, const float* __restrict__ dc_mn, int c0, int accumulate_xk,
    float* d_xk, float* d_x0, int SPB) {
  extern __shared__ float lds[];
  const int Tp = blockDim.x + kPad;
  float* xs = lds;
  const bool shared_x = (xk == x0);
  float* as = shared_x ? xs : lds + c.H * Tp;
  float* dxs = lds + (shared_x ? c.H : c.H + c.F) * Tp;
  const int64_t b0 = (int64_t)blockIdx.x * SPB;
  const int nb = (int)((c.B - b0 < SPB) ? (c.B - b0) : SPB);
  stage_tile(xk, b0, nb, c.H, c.D, xs, Tp);
  if (!shared_x) stage_tile(x0, b0, nb, c.F, c.D, as, Tp);
  const int tid = threadIdx.x;
  for (int h = 0; h < c.H; ++h) dxs[h * Tp + tid] = 0.f;
  __syncthreads();
  if (tid >= nb * c.D) return;  // no barrier below
  const int bl = tid / c.D;
  const int d = tid - bl * c.D;
  const int64_t b = b0 + bl;
  const int64_t m = b * c.D + d;
  float dc[NCH];
#pragma unroll
  for (int n = 0; n < NCH; ++n) dc[n] = (c0 + n < c.L) ? dc_mn[m * c.L + c0 + n] : 0.f;
  const float scale = c.masked ? 2.f : 1.f;
  float* d_x0_col = d_x0 + b * c.F * c.D + d;
  if (c0 + NCH <= c.L)
    dx_accumulate<NCH, true>(c, w, xs, as, dxs, Tp, c0, dc, scale, d_x0_col);
  else
    dx_accumulate<NCH, false>(c, w, xs, as, dxs, Tp, c0, dc, scale, d_x0_col);
  for (int h = 0; h < c.H; ++h) {
    const float v = dxs[h * Tp + tid] * scale;
    if (d_xk) {
      float* p = d_xk + (b * c.H + h) * c.D + d;
      *p = accumulate_xk ? *p + v : v;
    } else {
      d_x0_col[h * c.D] += v;  // layer 0: Xk is X0 (H == F)
    }
  }
}

// pair index -> (f, h); masked layers enumerate only f < h
__device__ __forceinline__ void pair_of(int q, int F, int H, int masked, int* f, int* h) {
  if (!masked) {
    *f = q / H;
    *h = q - *f * H;
    return;
  }
  int ff = 0, rem = q;
  while (rem >= H - 1 - ff) {
    rem -= H - 1 - ff;
    ++ff;
  }
  *f = ff;
  *h = ff + 1 + rem;
}

template <int NCH>
__global__ __launch_bounds__(kThreads) void cin_bwd_dw_kernel(
    CinDims c, const float* __restrict__ x0, const float* __restrict__ xk,
    const float* __restrict__ dc_mn, int P, int samples_per_split, int SUB,
    float* __restrict__ part) {
  extern __shared__ float lds[];
  const int Hp = c.H | 1, Fp = c.F | 1;
  float* xs = lds;                      // [SUB*D][Hp]
  float* as = lds + SUB * c.D * Hp;     // [SUB*D][Fp]
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = q < P;
  int f = 0, h = 0;
  if (valid) pair_of(q, c.F, c.H, c.masked, &f, &h);
  const int c0 = blockIdx.z * NCH;
  const bool full = (c0 + NCH <= c.L);
  float acc[NCH];
#pragma unroll
  for (int n = 0; n < NCH; ++n) acc[n] = 0.f;

  const int64_t s_beg = (int64_t)blockIdx.y * samples_per_split;
  int64_t s_end = s_beg + samples_per_split;
  if (s_end > c.B) s_end = c.B;
  for (int64_t b0 = s_beg; b0 < s_end; b0 += SUB) {
    const int nb = (int)((s_end - b0 < SUB) ? (s_end - b0) : SUB);
    __syncthreads();
    // transpose [b][h][d] -> [(b,d)][h]
    for (int idx = threadIdx.x; idx < nb * c.H * c.D; idx += blockDim.x) {
      const int bl = idx / (c.H * c.D);
      const int rem = idx - bl * c.H * c.D;
      const int hh = rem / c.D;
      const int d = rem - hh * c.D;
      xs[(bl * c.D + d) * Hp + hh] = xk[b0 * c.H * c.D + idx];
    }
    for (int idx = threadIdx.x; idx < nb * c.F * c.D; idx += blockDim.x) {
      const int bl = idx / (c.F * c.D);
      const int rem = idx - bl * c.F * c.D;
      const int ff = rem / c.D;
      const int d = rem - ff * c.D;
      as[(bl * c.D + d) * Fp + ff] = x0[b0 * c.F * c.D + idx];
    }
    __syncthreads();
    if (valid) {
      const int nm = nb * c.D;
      const float* __restrict__ dcrow = dc_mn + (b0 * c.D) * c.L + c0;
      for (int ml = 0; ml < nm; ++ml, dcrow += c.L) {
        const float z = as[ml * Fp + f] * xs[ml * Hp + h];
        if (full) {
#pragma unroll
          for (int n = 0; n < NCH; ++n) acc[n] = fmaf(z, dcrow[n], acc[n]);
        } else {
#pragma unroll
          for (int n = 0; n < NCH; ++n)
            if (c0 + n < c.L) acc[n] = fmaf(z, dcrow[n], acc[n]);
        }
      }
    }
  }
  if (!valid) return;
  const float scale = c.masked ? 2.f : 1.f;
  float* dst = part + ((int64_t)blockIdx.y * c.F * c.H + (f * c.H + h)) * c.L + c0;
#pragma unroll
  for (int n = 0; n < NCH; ++n)
    if (c0 + n < c.L) dst[n] = acc[n] * scale;
}

__global__ void cin_dw_reduce_kernel(CinDims c, const float* __restrict__ part, int n_splits,
                                     float* __restrict__ d_w) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)c.F * c.H * c.L;
  if (idx >= total) return;
  const int p = (int)(idx / c.L);
  const int f = p / c.H, h = p - f * c.H;
  float s = 0.f;
  if (!(c.masked && h <= f))
    for (int k = 0; k < n_splits; ++k) s += part[(int64_t)k * total + idx];
  d_w[(int64_t)p * cin_ldw(c) + (idx - (int64_t)p * c.L)] = s;
}

int pick_nch(int L) {
  static const int opts[] = {4, 8, 12, 16, 20, 24, 32};
  for (int o : opts)
    if (L <= o) return o;
  return 32;
}

// samples per block / block threads for the lane-per-(sample,d) kernels
void lane_geometry(int D, int rows, int* threads, int* spb, size_t* lds_bytes, int extra_rows) {
  int T = kThreads;
  // keep the tile within 64 KiB so at least two blocks share a CU
  while (T > 64 && (size_t)(rows + extra_rows) * (T + kPad) * sizeof(float) > 65536) T /= 2;
  if (T < D) T = ((D + 63) / 64) * 64;
  *threads = T;
  *spb = T / D;
  *lds_bytes = (size_t)(rows + extra_rows) * (T + kPad) * sizeof(float);
}

int check_dims(const char* who, int64_t B, int F, int H, int D, int L, int n_hidden, int masked) {
  XDFM_REQUIRE(B >= 0 && F > 0 && H > 0 && D > 0 && L > 0, "%s: bad sizes", who);
  XDFM_REQUIRE(n_hidden >= 0 && n_hidden <= L, "%s: n_hidden %d outside [0,%d]", who, n_hidden, L);
  XDFM_REQUIRE(!masked || F == H, "%s: the layer-0 mask needs H == F", who);
  XDFM_REQUIRE(D <= 1024, "%s: dim %d too large", who, D);
  return XDFM_OK;
}

int splits_for(int64_t B, int SUB) {
  int64_t tiles = (B + SUB - 1) / SUB;
  return (int)(tiles < kMaxSplits ? (tiles < 1 ? 1 : tiles) : kMaxSplits);
}

}  // namespace

#define DISPATCH_NCH(nch, CALL)  \
  switch (nch) {                 \
    case 4: { CALL(4); } break;  \
    case 8: { CALL(8); } break;  \
    case 12: { CALL(12); } break; \
    case 16: { CALL(16); } break; \
    case 20: { CALL(20); } break; \
    case 24: { CALL(24); } break; \
    default: { CALL(32); } break; \
  }

extern "C" int xdfm_cin_fwd_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                             size_t* floats) {
  XDFM_REQUIRE(floats && B >= 0 && F > 0 && H > 0 && D > 0 && L > 0,
               "xdfm_cin_fwd_workspace_floats: bad argument");
  *floats = xdfm_cin::mfma_supported(B, F, H, D, L) ? xdfm_cin::fwd_mfma_workspace_floats(B, F, H, D, L) : 0;
  return XDFM_OK;
}

extern "C" int xdfm_cin_fwd(const float* x0, const float* xk, const float* w, int64_t B, int32_t F,
                            int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked,
                            float* xnext, float* pooled, int32_t pool_stride, int32_t pool_off,
                            float* workspace, void* stream) {
  int rc = check_dims("xdfm_cin_fwd", B, F, H, D, L, n_hidden, masked);
  if (rc) return rc;
  XDFM_REQUIRE(x0 && xk && w, "xdfm_cin_fwd: null input");
  XDFM_REQUIRE(n_hidden == 0 || xnext, "xdfm_cin_fwd: xnext is null");
  XDFM_REQUIRE(n_hidden == L || pooled, "xdfm_cin_fwd: pooled is null");
  if (B == 0) return XDFM_OK;
  if (xdfm_cin::mfma_supported(B, F, H, D, L)) {
    CinDims cm{B, F, H, D, L, n_hidden, masked};
    XDFM_REQUIRE(workspace, "xdfm_cin_fwd: workspace is null (size it with xdfm_cin_fwd_workspace_floats)");
    return xdfm_cin::fwd_mfma(cm, x0, xk, w, xnext, pooled, pool_stride, pool_off, workspace,
                              (hipStream_t)stream);
  }
  const int nch = pick_nch(L);
  const int rows = (xk == x0) ? H : H + F;
  int threads, spb;
  size_t lds_bytes;
  lane_geometry(D, rows > nch ? rows : nch, &threads, &spb, &lds_bytes, 0);
  CinDims c{B, F, H, D, L, n_hidden, masked};
  dim3 grid((unsigned)((B + spb - 1) / spb), (unsigned)((L + nch - 1) / nch));
#define CALL(N)                                                                                  \
  XDFM_LAUNCH("cin_fwd_kernel<" #N ">", stream, cin_fwd_kernel<N>, grid, dim3(threads), lds_bytes, c, x0, \
                     xk, w, xnext, pooled, pool_stride, pool_off, spb)
  DISPATCH_NCH(nch, CALL);
#undef CALL
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_bwd_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                             size_t* floats) {
  XDFM_REQUIRE(floats && B >= 0 && F > 0 && H > 0 && D > 0 && L > 0,
               "xdfm_cin_bwd_workspace_floats: bad argument");
  const int SUB = (64 / D) > 0 ? (64 / D) : 1;
  *floats = (size_t)B * D * L + (size_t)splits_for(B, SUB) * F * H * L;
  if (xdfm_cin::mfma_supported(B, F, H, D, L)) {
    const size_t m = xdfm_cin::bwd_mfma_workspace_floats(B, F, H, D, L, 0);
    if (m > *floats) *floats = m;
  }
  return XDFM_OK;
}

extern "C" int xdfm_cin_bwd(const float* x0, const float* xk, const float* w, int64_t B, int32_t F,
                            int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked,
                            const float* d_xnext, const float* d_pooled, int32_t pool_stride,
                            int32_t pool_off, float* d_xk, float* d_x0, float* d_w, float* dw_part,
                            void* stream) {
  int rc = check_dims("xdfm_cin_bwd", B, F, H, D, L, n_hidden, masked);
  if (rc) return rc;
  XDFM_REQUIRE(x0 && xk && w && d_x0 && d_w && dw_part, "xdfm_cin_bwd: null argument");
  XDFM_REQUIRE(n_hidden == 0 || d_xnext, "xdfm_cin_bwd: d_xnext is null");
  XDFM_REQUIRE(n_hidden == L || d_pooled, "xdfm_cin_bwd: d_pooled is null");
  XDFM_REQUIRE((xk == x0) == (d_xk == nullptr),
               "xdfm_cin_bwd: d_xk must be NULL exactly when xk == x0 (layer 0)");
  hipStream_t st = (hipStream_t)stream;
  CinDims c{B, F, H, D, L, n_hidden, masked};
  if (B == 0) {
    XDFM_HIP(hipMemsetAsync(d_w, 0, (size_t)F * H * L * sizeof(float), st));
    return XDFM_OK;
  }
  if (xdfm_cin::mfma_supported(B, F, H, D, L))
    return xdfm_cin::bwd_mfma(c, x0, xk, w, nullptr, d_xnext, d_pooled, pool_stride, pool_off, d_xk, d_x0, d_w,
                              dw_part, 3, st, d_x0, nullptr);
  float* dc_mn = dw_part;
  float* part = dw_part + (size_t)B * D * L;
  {
    const int64_t total = B * D * L;
    XDFM_LAUNCH("cin_dcurr_kernel", st, cin_dcurr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c,
                       d_xnext, d_pooled, pool_stride, pool_off, dc_mn);
  }
  const int nch = pick_nch(L);
  const int chunks = (L + nch - 1) / nch;
  // input gradients
  {
    const int rows = ((xk == x0) ? H : H + F) + H;
    int threads, spb;
    size_t lds_bytes;
    lane_geometry(D, rows, &threads, &spb, &lds_bytes, 0);
    dim3 grid((unsigned)((B + spb - 1) / spb));
    for (int ch = 0; ch < chunks; ++ch) {
#define CALL(N)                                                                                   \
  XDFM_LAUNCH("cin_bwd_dx_kernel<" #N ">", st, cin_bwd_dx_kernel<N>, grid, dim3(threads), lds_bytes, c, x0, xk, w, dc_mn, \
                     ch * nch, ch > 0 ? 1 : 0, d_xk, d_x0, spb)
      DISPATCH_NCH(nch, CALL);
#undef CALL
    }
  }
  // filter gradient
  {
    const int SUB = (64 / D) > 0 ? (64 / D) : 1;
    const int n_splits = splits_for(B, SUB);
    const int samples_per_split = (int)((B + n_splits - 1) / n_splits);
    const int P = masked ? F * (F - 1) / 2 : F * H;
    const size_t lds_bytes = (size_t)SUB * D * ((H | 1) + (F | 1)) * sizeof(float);
    XDFM_REQUIRE(lds_bytes <= 160 * 1024, "xdfm_cin_bwd: tile does not fit the LDS (D=%d F=%d H=%d)",
                 D, F, H);
    if (P > 0) {
      dim3 grid((unsigned)((P + kThreads - 1) / kThreads), (unsigned)n_splits, (unsigned)chunks);
#define CALL(N)                                                                                  \
  XDFM_LAUNCH("cin_bwd_dw_kernel<" #N ">", st, cin_bwd_dw_kernel<N>, grid, dim3(kThreads), lds_bytes, c, x0, xk, dc_mn, P, \
                     samples_per_split, SUB, part)
      DISPATCH_NCH(nch, CALL);
#undef CALL
    }
    const int64_t total = (int64_t)F * H * L;
    XDFM_LAUNCH("cin_dw_reduce_kernel", st, cin_dw_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       c, part, P > 0 ? n_splits : 0, d_w);
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// ------------------------------------------------------------- filters packed ahead
extern "C" int xdfm_cin_pack_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L, int32_t masked,
                                    size_t* fwd_floats, size_t* bwd_floats) {
  XDFM_REQUIRE(fwd_floats && bwd_floats && B >= 0 && F > 0 && H > 0 && D > 0 && L > 0,
               "xdfm_cin_pack_floats: bad argument");
  xdfm_cin::pack_floats(B, F, H, D, L, masked, fwd_floats, bwd_floats);
  return XDFM_OK;
}

extern "C" int xdfm_cin_pack(const xdfm_cin_pack_job* jobs, int32_t n_jobs, int64_t B, int32_t D, void* stream) {
  XDFM_REQUIRE(jobs || n_jobs == 0, "xdfm_cin_pack: null jobs");
  XDFM_REQUIRE(B >= 0 && D > 0, "xdfm_cin_pack: bad sizes");
  if (B == 0 || n_jobs == 0) return XDFM_OK;
  return xdfm_cin::pack_all(jobs, n_jobs, B, D, (hipStream_t)stream);
}

extern "C" int xdfm_cin_fwd_packed(const float* x0, const float* xk, const float* fwd_image, int64_t B, int32_t F,
                                   int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked,
                                   float* xnext, float* pooled, int32_t pool_stride, int32_t pool_off,
                                   void* stream) {
  int rc = check_dims("xdfm_cin_fwd_packed", B, F, H, D, L, n_hidden, masked);
  if (rc) return rc;
  XDFM_REQUIRE(x0 && xk && fwd_image, "xdfm_cin_fwd_packed: null input");
  XDFM_REQUIRE(n_hidden == 0 || xnext, "xdfm_cin_fwd_packed: xnext is null");
  XDFM_REQUIRE(n_hidden == L || pooled, "xdfm_cin_fwd_packed: pooled is null");
  XDFM_REQUIRE(xdfm_cin::mfma_supported(B, F, H, D, L), "xdfm_cin_fwd_packed: this shape has no packed path");
  if (B == 0) return XDFM_OK;
  CinDims cm{B, F, H, D, L, n_hidden, masked};
  return xdfm_cin::fwd_mfma(cm, x0, xk, nullptr, xnext, pooled, pool_stride, pool_off,
                            const_cast<float*>(fwd_image), (hipStream_t)stream);
}

extern "C" int xdfm_cin_bwd_packed(const float* x0, const float* xk, const float* bwd_image, int64_t B, int32_t F,
                                   int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked,
                                   const float* d_xnext, const float* d_pooled, int32_t pool_stride,
                                   int32_t pool_off, float* d_xk, float* d_x0, float* d_w, float* dw_part,
                                   int32_t phases, const float* d_x0_in, const float* d_x0_in2, int32_t ldw,
                                   void* stream) {
  int rc = check_dims("xdfm_cin_bwd_packed", B, F, H, D, L, n_hidden, masked);
  if (rc) return rc;
  XDFM_REQUIRE(phases >= 1 && phases <= 15, "xdfm_cin_bwd_packed: phases is a mask of XDFM_CIN_BWD_*");
  XDFM_REQUIRE(x0 && xk && bwd_image && d_x0 && d_w && dw_part, "xdfm_cin_bwd_packed: null argument");
  XDFM_REQUIRE(n_hidden == 0 || d_xnext, "xdfm_cin_bwd_packed: d_xnext is null");
  XDFM_REQUIRE(n_hidden == L || d_pooled, "xdfm_cin_bwd_packed: d_pooled is null");
  XDFM_REQUIRE((xk == x0) == (d_xk == nullptr),
               "xdfm_cin_bwd_packed: d_xk must be NULL exactly when xk == x0 (layer 0)");
  XDFM_REQUIRE(xdfm_cin::mfma_supported(B, F, H, D, L), "xdfm_cin_bwd_packed: this shape has no packed path");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (phases & (XDFM_CIN_BWD_DW | XDFM_CIN_BWD_DW_REDUCE))
      XDFM_HIP(hipMemsetAsync(d_w, 0, (size_t)F * H * L * sizeof(float), st));
    return XDFM_OK;
  }
  XDFM_REQUIRE(ldw == 0 || ldw >= L, "xdfm_cin_bwd_packed: ldw must be 0 or >= L");
  CinDims c{B, F, H, D, L, n_hidden, masked, ldw};
  return xdfm_cin::bwd_mfma(c, x0, xk, nullptr, bwd_image, d_xnext, d_pooled, pool_stride, pool_off, d_xk, d_x0,
                            d_w, dw_part, phases, st, d_x0_in, d_x0_in2);
}

// ------------------------------------------------------------------ cross_activation
// `curr_out = self._activate(curr_out, hparams.cross_activation)` (xDeepFM.py:242, :402) sits
// between the layer's product and its split into hidden / pooled channels.  With a
// non-identity activation the layer kernels run with every channel "hidden" (raw curr into a
// buffer of its own) and these two elementwise kernels do the rest.
namespace {

// y = act(curr) in place; xnext[b,n,d] = y (n < n_hidden); pooled[b, off+n-n_hidden] = sum_d y
__global__ __launch_bounds__(256) void cin_act_fwd_kernel(float* __restrict__ curr, int64_t B, int L, int D,
                                                          int n_hidden, int act, float* __restrict__ xnext,
                                                          float* __restrict__ pooled, int pool_stride,
                                                          int pool_off) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, n)
  if (idx >= B * L) return;
  const int64_t b = idx / L;
  const int n = (int)(idx - b * L);
  float* row = curr + idx * D;
  float s = 0.f;
  for (int d = 0; d < D; ++d) {
    const float y = activate(row[d], act);
    row[d] = y;
    if (n < n_hidden) xnext[(b * n_hidden + n) * D + d] = y;
    s += y;  // reduce_sum over D in d order (:270)
  }
  if (n >= n_hidden) pooled[b * pool_stride + pool_off + n - n_hidden] = s;
}

// d_curr[b,n,d] = upstream(b,n,d) * act'(y)
__global__ __launch_bounds__(256) void cin_act_bwd_kernel(const float* __restrict__ y,
                                                          const float* __restrict__ d_xnext,
                                                          const float* __restrict__ d_pooled, int pool_stride,
                                                          int pool_off, int64_t B, int L, int D, int n_hidden,
                                                          int act, float* __restrict__ d_curr) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, n, d)
  if (idx >= B * L * D) return;
  const int d = (int)(idx % D);
  const int64_t bn = idx / D;
  const int64_t b = bn / L;
  const int n = (int)(bn - b * L);
  const float up = n < n_hidden ? d_xnext[(b * n_hidden + n) * D + d]
                                : d_pooled[b * pool_stride + pool_off + n - n_hidden];
  d_curr[idx] = up * act_grad(y[idx], act);
}

}  // namespace

extern "C" int xdfm_cin_act_fwd(float* curr, int64_t B, int32_t L, int32_t D, int32_t n_hidden, int32_t act,
                                float* xnext, float* pooled, int32_t pool_stride, int32_t pool_off,
                                void* stream) {
  XDFM_REQUIRE(curr && B >= 0 && L > 0 && D > 0 && n_hidden >= 0 && n_hidden <= L, "xdfm_cin_act_fwd: bad argument");
  XDFM_REQUIRE(act >= XDFM_ACT_IDENTITY && act <= XDFM_ACT_ELU, "xdfm_cin_act_fwd: unknown activation %d", act);
  XDFM_REQUIRE(n_hidden == 0 || xnext, "xdfm_cin_act_fwd: xnext is null");
  XDFM_REQUIRE(n_hidden == L || pooled, "xdfm_cin_act_fwd: pooled is null");
  if (B == 0) return XDFM_OK;
  XDFM_LAUNCH("cin_act_fwd_kernel", stream, cin_act_fwd_kernel, dim3((unsigned)((B * L + 255) / 256)), dim3(256), 0,
              curr, B, L, D, n_hidden, act, xnext, pooled, pool_stride, pool_off);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_act_bwd(const float* y, const float* d_xnext, const float* d_pooled, int32_t pool_stride,
                                int32_t pool_off, int64_t B, int32_t L, int32_t D, int32_t n_hidden, int32_t act,
                                float* d_curr, void* stream) {
  XDFM_REQUIRE(y && d_curr && B >= 0 && L > 0 && D > 0 && n_hidden >= 0 && n_hidden <= L,
               "xdfm_cin_act_bwd: bad argument");
  XDFM_REQUIRE(act >= XDFM_ACT_IDENTITY && act <= XDFM_ACT_ELU, "xdfm_cin_act_bwd: unknown activation %d", act);
  XDFM_REQUIRE(n_hidden == 0 || d_xnext, "xdfm_cin_act_bwd: d_xnext is null");
  XDFM_REQUIRE(n_hidden == L || d_pooled, "xdfm_cin_act_bwd: d_pooled is null");
  if (B == 0) return XDFM_OK;
  XDFM_LAUNCH("cin_act_bwd_kernel", stream, cin_act_bwd_kernel, dim3((unsigned)((B * L * D + 255) / 256)), dim3(256),
              0, y, d_xnext, d_pooled, pool_stride, pool_off, B, L, D, n_hidden, act, d_curr);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}
