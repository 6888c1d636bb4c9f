// Dense layers of the DNN tower and the two output heads on the fp32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32, exact fp32).
//
// Reference: xDeepFM.py:473-533 (`xw_plus_b` + `_active_layer`, base_model.py:324-353) and
// the CIN head xDeepFM.py:279-293.
//
//   forward   y[b,n]  = act(sum_k x[b,k] w[k,n] + bias[n])     M = rows, N = out,  K = in
//   d-input   dx[b,k] = sum_n dz[b,n] w[k,n]                    M = rows, N = in,   K = out
//   d-weight  dw[k,n] = sum_b x[b,k] dz[b,n], db[n] = sum_b dz  M = in,   N = out,  K = rows
// with dz = dy * act'(y) formed on the fly from the saved output.  The tower is tiny next to
// its input (x = the [B, F*D] embedding block): the kernels are HBM-bound on x, so the work of
// one row tile is split over the 4 waves of a block to keep enough loads in flight.
// Reductions over the batch run in a fixed split order (no float atomics).
//
// Layouts: x[B][K], w[K][Nout], b[Nout], y[B][Nout].
#include <hip/hip_runtime.h>

#include <cstring>

#include "act.h"
#include "xdfm_common.h"
#include "stamp.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------ forward
// Block = 4 waves on the same 16 rows; wave w takes the 16-wide k steps w, w+4, ...;
// lane (i = lane&15, kk = lane>>4) loads x[row i][k0 + 4kk .. +3] as one 16-byte load and
// feeds it to 4 MFMAs whose k slot kk stands for k = k0 + 4kk + s (s = 0..3).
template <int NT>
__global__ __launch_bounds__(256) void dense_fwd_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    int64_t B, int K, int Nout, int act, float* __restrict__ y) {
  xdfm::Stamp stamp_(xdfm::kStampDenseFwd);
  side_chain_priority();
  __shared__ float red[3][NT][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int c0 = blockIdx.y * (NT * 16);
  int64_t arow = row0 + i;
  if (arow >= B) arow = B - 1;  // clamped rows compute garbage that is never stored
  const float* xr = x + arow * K;
  const bool vec = ((K & 3) == 0);
  int choff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ch = c0 + nt * 16 + i;
    choff[nt] = ch < Nout ? ch : Nout - 1;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Stage {
    float a[4];
    float b[4][NT];
  };
  auto issue = [&](Stage& st, int k0) {
    const int k = k0 + 4 * kk;
    if (vec && k + 3 < K) {
      const float4 v = *reinterpret_cast<const float4*>(xr + k);
      st.a[0] = v.x;
      st.a[1] = v.y;
      st.a[2] = v.z;
      st.a[3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) st.a[s] = (k + s < K) ? xr[k + s] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kc = (k + s < K) ? k + s : K - 1;  // a is zero there
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) st.b[s][nt] = w[(unsigned)(kc * Nout + choff[nt])];
    }
  };
  auto compute = [&](const Stage& st) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma(st.a[s], st.b[s][nt], acc[nt]);
  };
  // this wave's steps: k0 = 16*wave, 16*(wave+4), ...; two stages in flight
  const int nsteps = (K + 15) / 16;
  Stage s0, s1;
  int st_i = wave;
  if (st_i < nsteps) issue(s0, 16 * st_i);
  while (st_i < nsteps) {
    const int n1 = st_i + 4, n2 = st_i + 8;
    if (n1 < nsteps) issue(s1, 16 * n1);
    __builtin_amdgcn_sched_barrier(0);
    compute(s0);
    __builtin_amdgcn_sched_barrier(0);
    if (n1 >= nsteps) break;
    if (n2 < nsteps) issue(s0, 16 * n2);
    __builtin_amdgcn_sched_barrier(0);
    compute(s1);
    __builtin_amdgcn_sched_barrier(0);
    st_i = n2;
  }
  // sum the 4 waves' partial tiles in wave order, then bias + activation + store (wave 0)
  if (wave > 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][nt][lane][r] = acc[nt][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = c0 + nt * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nt][r];
        v += red[0][nt][lane][r];
        v += red[1][nt][lane][r];
        v += red[2][nt][lane][r];
        const int64_t row = row0 + kk * 4 + r;
        if (row < B && ch < Nout) y[row * Nout + ch] = activate(v + bias[ch], act);
      }
    }
  }
}

// ---------------------------------------------------------------- input gradient
// Block = 4 waves on the same 16 rows; wave w takes the 16-wide column tiles w, w+4, ...
// A[i=row][k=n] = dz[row][4s+kk] (registers), B[k=n][j=col] = w[c0+j][4s+kk].
template <int SMAX>
__global__ __launch_bounds__(256) void dense_bwd_dx_mfma_kernel(
    const float* __restrict__ w, const float* __restrict__ y, const float* __restrict__ dy,
    int64_t B, int K, int Nout, int n0, int n1, int act, int accumulate, float* __restrict__ dx) {
  xdfm::Stamp stamp_(xdfm::kStampDenseBwdDx);
  side_chain_priority();
  // channels [n0, n1) of the layer's Nout: wide layers are summed over several launches
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int S = (n1 - n0 + 3) >> 2;
  float dzA[SMAX];
  {
    int64_t arow = row0 + i;
    const bool ok = arow < B;
    if (!ok) arow = B - 1;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const int n = n0 + 4 * s + kk;
      const int nc = n < n1 ? n : n1 - 1;
      const float g = dy[arow * Nout + nc] * act_grad(y[arow * Nout + nc], act);
      dzA[s] = (ok && s < S && n < n1) ? g : 0.f;
    }
  }
  const int ntile = (K + 15) / 16;
  for (int t = wave; t < ntile; t += 4) {
    const int col = 16 * t + i;
    const int colc = col < K ? col : K - 1;
    float bv[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const int n = n0 + 4 * s + kk;
      bv[s] = w[(unsigned)(colc * Nout + (n < n1 ? n : n1 - 1))];  // dzA is zero past n1
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SMAX; ++s) acc = mfma(dzA[s], bv[s], acc);
    if (col < K) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + kk * 4 + r;
        if (row < B) {
          float* p = dx + row * K + col;
          *p = accumulate ? *p + acc[r] : acc[r];
        }
      }
    }
  }
}

// --------------------------------------------------------------- weight gradient
// Wave = PT k tiles (16 input columns each; column K is the bias row, x = 1) x NT channel
// tiles, over the rows of one split.  A[i=k][slot=row] = x[row][k0+i], B[slot=row][j] = dz[row][c0+j].
template <int PT, int NT>
__device__ __forceinline__ void dense_dw_body(const float* __restrict__ x, const float* __restrict__ y,
                                              const float* __restrict__ dy, int64_t B, int K, int Nout, int act,
                                              int rows_per_split, float* __restrict__ part, int split, int by,
                                              int bz) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int ktiles = (K + 1 + 15) / 16;
  const int t0 = (by * 4 + wave) * PT;
  if (t0 >= ktiles) return;  // no barrier in this kernel
  const int c0 = bz * (NT * 16);
  int kcol[PT];
#pragma unroll
  for (int p = 0; p < PT; ++p) kcol[p] = 16 * (t0 + p) + i;
  int choff[NT];
  bool chok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ch = c0 + nt * 16 + i;
    chok[nt] = ch < Nout;
    choff[nt] = chok[nt] ? ch : Nout - 1;
  }
  f32x4 acc[PT][NT];
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[p][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t r_beg = (int64_t)split * rows_per_split;
  int64_t r_end = r_beg + rows_per_split;
  if (r_end > B) r_end = B;
  if (r_beg >= r_end) r_end = r_beg;  // an empty split still writes its zero partials
  // issue() only loads; masks and the activation gradient are applied in compute(), so that no
  // stage waits for its own loads (a select right after a load would)
  struct Stage {
    float a[PT];
    float g[NT], yv[NT];
    bool ok;
  };
  auto issue = [&](Stage& st, int64_t r) {
    st.ok = r < r_end;
    int64_t rc = st.ok ? r : r_end - 1;
    if (rc < 0) rc = 0;
    const float* xr = x + rc * K;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int k = kcol[p];
      st.a[p] = xr[k < K ? k : K - 1];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      st.g[nt] = dy[rc * Nout + choff[nt]];
      st.yv[nt] = y[rc * Nout + choff[nt]];
    }
  };
  auto compute = [&](const Stage& st) {
    float b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float g = st.g[nt] * act_grad(st.yv[nt], act);
      b[nt] = (chok[nt] && st.ok) ? g : 0.f;  // rows past the split contribute nothing
    }
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int k = kcol[p];
      const float a = (k < K) ? st.a[p] : (k == K ? 1.f : 0.f);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[p][nt] = mfma(a, b[nt], acc[p][nt]);
    }
  };
  // four stages (16 rows) in flight: the loop runs at memory latency otherwise
  Stage s0, s1, s2, s3;
  int64_t r = r_beg + kk;
  issue(s0, r);
  issue(s1, r + 4);
  issue(s2, r + 8);
  for (int64_t rb = r_beg; rb < r_end; rb += 16) {
    issue(s3, r + 12);
    __builtin_amdgcn_sched_barrier(0);
    compute(s0);
    __builtin_amdgcn_sched_barrier(0);
    issue(s0, r + 16);
    __builtin_amdgcn_sched_barrier(0);
    compute(s1);
    __builtin_amdgcn_sched_barrier(0);
    issue(s1, r + 20);
    __builtin_amdgcn_sched_barrier(0);
    compute(s2);
    __builtin_amdgcn_sched_barrier(0);
    issue(s2, r + 24);
    __builtin_amdgcn_sched_barrier(0);
    compute(s3);
    __builtin_amdgcn_sched_barrier(0);
    r += 16;
  }
  // part[split][k][n], k in [0, K] (row K = bias)
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = c0 + nt * 16 + i;
      if (ch < Nout) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int k = 16 * (t0 + p) + kk * 4 + rr;
          if (k <= K) part[((int64_t)split * (K + 1) + k) * Nout + ch] = acc[p][nt][rr];
        }
      }
    }
}

template <int PT, int NT>
__global__ __launch_bounds__(256) void dense_bwd_dw_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    int64_t B, int K, int Nout, int act, int rows_per_split, float* __restrict__ part) {
  xdfm::Stamp stamp_(xdfm::kStampDenseBwdDw);
  side_chain_priority();
  dense_dw_body<PT, NT>(x, y, dy, B, K, Nout, act, rows_per_split, part, blockIdx.x, blockIdx.y, blockIdx.z);
}

// dw / db = sum over splits: 8 thread groups add every 8th split in order, then the 8 partial
// sums are added in group order (fixed association)
__device__ __forceinline__ void dense_dw_reduce_body(const float* __restrict__ part, int n_splits, int K, int Nout,
                                                     float* __restrict__ dw, float* __restrict__ db, int block) {
  __shared__ float red[8][33];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t idx = (int64_t)block * 32 + e;
  const int64_t total = (int64_t)(K + 1) * Nout;
  float s = 0.f;
  if (idx < total) {
    // 8 loads in flight, added in split order
    for (int k0 = g; k0 < n_splits; k0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 8 * u;
        v[u] = part[(int64_t)(k < n_splits ? k : k0) * total + idx];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (k0 + 8 * u < n_splits) ? v[u] : 0.f;
    }
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && idx < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][e];
    if (idx < (int64_t)K * Nout)
      dw[idx] = t;
    else
      db[idx - (int64_t)K * Nout] = t;
  }
}

__global__ __launch_bounds__(256) void dense_dw_reduce_kernel(const float* __restrict__ part, int n_splits,
                                                              int K, int Nout, float* __restrict__ dw,
                                                              float* __restrict__ db) {
  xdfm::Stamp stamp_(xdfm::kStampDenseBwdDw);
  side_chain_priority();
  dense_dw_reduce_body(part, n_splits, K, Nout, dw, db, blockIdx.x);
}

// ------------------------------------------------------- several weight gradients, one launch
// A step's small weight gradients (the tower's layers, the linear bias, the CIN head) are a dozen launches
// of ~5 us each when taken one by one, on a stream that shares the chip with kernels which fill it: each of
// them waits for room (tools/stamp_timeline.py: 100-150 us for the lot).  Here they are the jobs of ONE
// launch over common row splits, and their partial sums are added by ONE second launch.
constexpr int kMaxDwJobs = 8;
struct DwJobDev {
  const float *x, *y, *dy;
  float *dw, *db;
  int K, Nout, act, nt2;
  int first_block;   // of the weight-gradient launch, in blockIdx.y
  int gy;            // its k-tile blocks (blockIdx.y of the single-job kernel); gz = blocks / gy
  int first_rblock;  // of the reduction launch
  int64_t part_off;  // floats
};
struct DwJobs {
  int n, n_splits, rows_per_split;
  int64_t B;
  DwJobDev j[kMaxDwJobs];
};

__global__ __launch_bounds__(256) void dense_bwd_dw_multi_kernel(DwJobs jobs, float* __restrict__ part) {
  xdfm::Stamp stamp_(xdfm::kStampDenseBwdDw);
  side_chain_priority();
  int q = 0;
  for (int k = 1; k < jobs.n; ++k)
    if ((int)blockIdx.y >= jobs.j[k].first_block) q = k;
  const DwJobDev& J = jobs.j[q];
  const int rel = (int)blockIdx.y - J.first_block;
  const int by = rel % J.gy, bz = rel / J.gy;
  if (J.nt2)
    dense_dw_body<4, 2>(J.x, J.y, J.dy, jobs.B, J.K, J.Nout, J.act, jobs.rows_per_split, part + J.part_off,
                        blockIdx.x, by, bz);
  else
    dense_dw_body<4, 1>(J.x, J.y, J.dy, jobs.B, J.K, J.Nout, J.act, jobs.rows_per_split, part + J.part_off,
                        blockIdx.x, by, bz);
}

__global__ __launch_bounds__(256) void dense_dw_reduce_multi_kernel(DwJobs jobs, const float* __restrict__ part) {
  xdfm::Stamp stamp_(xdfm::kStampDenseBwdDw);
  side_chain_priority();
  int q = 0;
  for (int k = 1; k < jobs.n; ++k)
    if ((int)blockIdx.x >= jobs.j[k].first_rblock) q = k;
  const DwJobDev& J = jobs.j[q];
  dense_dw_reduce_body(part + J.part_off, jobs.n_splits, J.K, J.Nout, J.dw, J.db, (int)blockIdx.x - J.first_rblock);
}

// ------------------------------------------------------------- a plain tower, one launch per pass
// in_dim -> size[0] -> ... -> size[n-1] -> 1 with no batch norm, dropout or softmax between the layers
// (xDeepFM.py:473-533): a row's activations depend on that row alone, so 16 rows go through ALL layers in
// one block.  The first layer (the only one with a long K) is the matrix-core loop of dense_fwd_mfma_kernel
// / dense_bwd_dx_mfma_kernel; the others are at most 64 wide and run as plain FMA chains over LDS.
constexpr int kTowerMaxLayers = 4;
constexpr int kTowerMaxWidth = 64;
struct TowerDev {
  int n, in_dim;
  int size[kTowerMaxLayers], act[kTowerMaxLayers];
  const float *w[kTowerMaxLayers], *b[kTowerMaxLayers];
  const float *w_out, *b_out;
  float* h[kTowerMaxLayers];
  float* d_h[kTowerMaxLayers];
};

template <int NT>
__global__ __launch_bounds__(256) void tower_fwd_kernel(TowerDev t, const float* __restrict__ x, int64_t B,
                                                        float* __restrict__ out) {
  xdfm::Stamp stamp_(xdfm::kStampDenseFwd);
  side_chain_priority();
  __shared__ float red[3][NT][64][4];
  __shared__ float hs[2][16][kTowerMaxWidth + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int K = t.in_dim, N1 = t.size[0];
  const float* __restrict__ w = t.w[0];
  int64_t arow = row0 + i;
  if (arow >= B) arow = B - 1;  // clamped rows compute values that are never stored
  const float* xr = x + arow * K;
  const bool vec = ((K & 3) == 0);
  int choff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ch = nt * 16 + i;
    choff[nt] = ch < N1 ? ch : N1 - 1;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  struct Stage {
    float a[4];
    float b[4][NT];
  };
  auto issue = [&](Stage& st, int k0) {
    const int k = k0 + 4 * kk;
    if (vec && k + 3 < K) {
      const float4 v = *reinterpret_cast<const float4*>(xr + k);
      st.a[0] = v.x;
      st.a[1] = v.y;
      st.a[2] = v.z;
      st.a[3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) st.a[s] = (k + s < K) ? xr[k + s] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kc = (k + s < K) ? k + s : K - 1;  // a is zero there
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) st.b[s][nt] = w[(unsigned)(kc * N1 + choff[nt])];
    }
  };
  auto compute = [&](const Stage& st) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma(st.a[s], st.b[s][nt], acc[nt]);
  };
  const int nsteps = (K + 15) / 16;
  Stage s0, s1;
  int st_i = wave;
  if (st_i < nsteps) issue(s0, 16 * st_i);
  while (st_i < nsteps) {
    const int n1 = st_i + 4, n2 = st_i + 8;
    if (n1 < nsteps) issue(s1, 16 * n1);
    __builtin_amdgcn_sched_barrier(0);
    compute(s0);
    __builtin_amdgcn_sched_barrier(0);
    if (n1 >= nsteps) break;
    if (n2 < nsteps) issue(s0, 16 * n2);
    __builtin_amdgcn_sched_barrier(0);
    compute(s1);
    __builtin_amdgcn_sched_barrier(0);
    st_i = n2;
  }
  if (wave > 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][nt][lane][r] = acc[nt][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* __restrict__ h0 = t.h[0];
    const float* __restrict__ b0 = t.b[0];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = nt * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nt][r];
        v += red[0][nt][lane][r];
        v += red[1][nt][lane][r];
        v += red[2][nt][lane][r];
        const int rl = kk * 4 + r;
        if (ch < N1) {
          v = activate(v + b0[ch], t.act[0]);
          hs[0][rl][ch] = v;
          if (row0 + rl < B) h0[(row0 + rl) * N1 + ch] = v;
        }
      }
    }
  }
  __syncthreads();
  int cur = 0;
  for (int l = 1; l < t.n; ++l) {
    const int Kl = t.size[l - 1], Nl = t.size[l];
    const float* __restrict__ wl = t.w[l];
    const float* __restrict__ bl = t.b[l];
    float* __restrict__ hl = t.h[l];
    for (int idx = threadIdx.x; idx < 16 * Nl; idx += 256) {
      const int rl = idx / Nl, ch = idx - rl * Nl;
      float s = 0.f;
      for (int k = 0; k < Kl; ++k) s = fmaf(hs[cur][rl][k], wl[k * Nl + ch], s);
      const float v = activate(s + bl[ch], t.act[l]);
      hs[cur ^ 1][rl][ch] = v;
      if (row0 + rl < B) hl[(row0 + rl) * Nl + ch] = v;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (threadIdx.x < 16 && row0 + threadIdx.x < B) {
    const int Kl = t.size[t.n - 1];
    float s = 0.f;
    for (int k = 0; k < Kl; ++k) s = fmaf(hs[cur][threadIdx.x][k], t.w_out[k], s);
    out[row0 + threadIdx.x] = s + t.b_out[0];
  }
}

// d_h[l] = gradient w.r.t. the output of hidden layer l (what the weight-gradient pass reads as dy),
// dx = gradient w.r.t. the tower's input, from g = d(loss)/d(out).
template <int SMAX>
__global__ __launch_bounds__(256) void tower_bwd_dx_kernel(TowerDev t, const float* __restrict__ g, int64_t B,
                                                           int accumulate, float* __restrict__ dx) {
  xdfm::Stamp stamp_(xdfm::kStampDenseBwdDx);
  side_chain_priority();
  __shared__ float ds[2][16][kTowerMaxWidth + 1];  // d(loss)/d(pre-activation) of the layer in hand
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  int cur = 0;
  {
    const int l = t.n - 1, Nl = t.size[l];
    const float* __restrict__ hl = t.h[l];
    float* __restrict__ dl = t.d_h[l];
    for (int idx = threadIdx.x; idx < 16 * Nl; idx += 256) {
      const int rl = idx / Nl, k = idx - rl * Nl;
      const int64_t row = row0 + rl;
      float dz = 0.f;
      if (row < B) {
        const float v = g[row] * t.w_out[k];
        dl[row * Nl + k] = v;
        dz = v * act_grad(hl[row * Nl + k], t.act[l]);
      }
      ds[cur][rl][k] = dz;
    }
  }
  __syncthreads();
  for (int l = t.n - 1; l >= 1; --l) {
    const int Kl = t.size[l - 1], Nl = t.size[l];
    const float* __restrict__ wl = t.w[l];
    const float* __restrict__ hp = t.h[l - 1];
    float* __restrict__ dp = t.d_h[l - 1];
    for (int idx = threadIdx.x; idx < 16 * Kl; idx += 256) {
      const int rl = idx / Kl, k = idx - rl * Kl;
      const int64_t row = row0 + rl;
      float s = 0.f;
      for (int n = 0; n < Nl; ++n) s = fmaf(ds[cur][rl][n], wl[k * Nl + n], s);
      float dz = 0.f;
      if (row < B) {
        dp[row * Kl + k] = s;
        dz = s * act_grad(hp[row * Kl + k], t.act[l - 1]);
      }
      ds[cur ^ 1][rl][k] = dz;
    }
    __syncthreads();
    cur ^= 1;
  }
  // first layer: dx[row][col] = sum_n dz[row][n] w[col][n] on the matrix cores (dense_bwd_dx_mfma_kernel)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int K = t.in_dim, N1 = t.size[0];
  const float* __restrict__ w = t.w[0];
  float dzA[SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    const int n = 4 * s + kk;
    dzA[s] = n < N1 ? ds[cur][i][n] : 0.f;
  }
  const int ntile = (K + 15) / 16;
  for (int tl = wave; tl < ntile; tl += 4) {
    const int col = 16 * tl + i;
    const int colc = col < K ? col : K - 1;
    float bv[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const int n = 4 * s + kk;
      bv[s] = w[(unsigned)(colc * N1 + (n < N1 ? n : N1 - 1))];  // dzA is zero past N1
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SMAX; ++s) acc = mfma(dzA[s], bv[s], acc);
    if (col < K) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + kk * 4 + r;
        if (row < B) {
          float* p = dx + row * K + col;
          *p = accumulate ? *p + acc[r] : acc[r];
        }
      }
    }
  }
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

// row splits of the weight-gradient pass: as many as keep every split at >= 32 rows and the
// partial sums ([split][K+1][Nout]) around 4 MB, at most 256
int splits_for(int64_t B, int K, int Nout) {
  int64_t s = (B + 31) / 32;
  const int64_t by_size = ((int64_t)1 << 20) / ((int64_t)(K + 1) * Nout);
  if (s > by_size) s = by_size;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

bool act_ok(int act) { return act >= XDFM_ACT_IDENTITY && act <= XDFM_ACT_ELU; }

}  // namespace

extern "C" int xdfm_dense_fwd(const float* x, const float* w, const float* b, int64_t B, int32_t K,
                              int32_t Nout, int32_t act, float* y, void* stream) {
  XDFM_REQUIRE(x && w && b && y, "xdfm_dense_fwd: null argument");
  XDFM_REQUIRE(B >= 0 && K > 0 && Nout > 0, "xdfm_dense_fwd: bad sizes");
  XDFM_REQUIRE((int64_t)K * Nout < ((int64_t)1 << 31), "xdfm_dense_fwd: weight too large");
  XDFM_REQUIRE(act_ok(act), "xdfm_dense_fwd: unknown activation %d", act);
  if (B == 0) return XDFM_OK;
  const int nt_total = (Nout + 15) / 16;
  const unsigned rows = (unsigned)((B + 15) / 16);
  if (nt_total >= 2) {
    dim3 grid(rows, (unsigned)((nt_total + 1) / 2));
    XDFM_LAUNCH("dense_fwd_mfma_kernel<2>", stream, dense_fwd_mfma_kernel<2>, grid, dim3(256), 0, x, w, b, B,
                K, Nout, act, y);
  } else {
    dim3 grid(rows, 1);
    XDFM_LAUNCH("dense_fwd_mfma_kernel<1>", stream, dense_fwd_mfma_kernel<1>, grid, dim3(256), 0, x, w, b, B,
                K, Nout, act, y);
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_dense_bwd_workspace_floats(int64_t B, int32_t K, int32_t Nout, size_t* floats) {
  XDFM_REQUIRE(floats && B >= 0 && K > 0 && Nout > 0, "xdfm_dense_bwd_workspace_floats: bad argument");
  *floats = (size_t)splits_for(B, K, Nout) * (K + 1) * Nout;
  return XDFM_OK;
}

extern "C" int xdfm_dense_bwd(const float* x, const float* w, const float* y, const float* dy,
                              int64_t B, int32_t K, int32_t Nout, int32_t act, float* dx,
                              int32_t accumulate_dx, float* dw, float* db, float* part,
                              void* stream) {
  XDFM_REQUIRE(x && w && y && dy, "xdfm_dense_bwd: null argument");
  XDFM_REQUIRE((dw && db && part) || (!dw && !db && dx),
               "xdfm_dense_bwd: dw, db and part come together; without them dx must be given");
  XDFM_REQUIRE(B >= 0 && K > 0 && Nout > 0, "xdfm_dense_bwd: bad sizes");
  XDFM_REQUIRE((int64_t)K * Nout < ((int64_t)1 << 31), "xdfm_dense_bwd: weight too large");
  XDFM_REQUIRE(act_ok(act), "xdfm_dense_bwd: unknown activation %d", act);
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (dw) {
      XDFM_HIP(hipMemsetAsync(dw, 0, (size_t)K * Nout * sizeof(float), st));
      XDFM_HIP(hipMemsetAsync(db, 0, (size_t)Nout * sizeof(float), st));
    }
    return XDFM_OK;
  }
  const int n_splits = splits_for(B, K, Nout);
  const int rows_per_split = (int)((B + n_splits - 1) / n_splits);
  const int nt_total = (Nout + 15) / 16;
  const int ktiles = (K + 1 + 15) / 16;
  if (dw) {
    constexpr int PT = 4;
    if (nt_total >= 2) {
      dim3 grid((unsigned)n_splits, (unsigned)((ktiles + 4 * PT - 1) / (4 * PT)), (unsigned)((nt_total + 1) / 2));
      XDFM_LAUNCH("dense_bwd_dw_mfma_kernel<4, 2>", st, (dense_bwd_dw_mfma_kernel<PT, 2>), grid, dim3(256), 0,
                  x, y, dy, B, K, Nout, act, rows_per_split, part);
    } else {
      dim3 grid((unsigned)n_splits, (unsigned)((ktiles + 4 * PT - 1) / (4 * PT)), 1);
      XDFM_LAUNCH("dense_bwd_dw_mfma_kernel<4, 1>", st, (dense_bwd_dw_mfma_kernel<PT, 1>), grid, dim3(256), 0,
                  x, y, dy, B, K, Nout, act, rows_per_split, part);
    }
    const int64_t tot = (int64_t)(K + 1) * Nout;
    XDFM_LAUNCH("dense_dw_reduce_kernel", st, dense_dw_reduce_kernel, dim3((unsigned)((tot + 31) / 32)),
                dim3(256), 0, part, n_splits, K, Nout, dw, db);
  }
  if (dx) {
    dim3 grid((unsigned)((B + 15) / 16));
    // 256 channels per launch; the launches of a wider layer add up in channel order
    for (int n0 = 0; n0 < Nout; n0 += 256) {
      const int n1 = n0 + 256 < Nout ? n0 + 256 : Nout;
      const int S = (n1 - n0 + 3) / 4;
      const int acc = (n0 > 0) ? 1 : accumulate_dx;
#define CALL(SM)                                                                                       \
  XDFM_LAUNCH("dense_bwd_dx_mfma_kernel<" #SM ">", st, dense_bwd_dx_mfma_kernel<SM>, grid, dim3(256), 0, w, \
              y, dy, B, K, Nout, n0, n1, act, acc, dx)
      if (S <= 1)
        CALL(1);
      else if (S <= 5)
        CALL(5);
      else if (S <= 16)
        CALL(16);
      else
        CALL(64);
#undef CALL
    }
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_dense_bwd_dw_multi_workspace_floats(const xdfm_dw_job* jobs, int32_t n_jobs, int64_t B,
                                                        size_t* floats) {
  XDFM_REQUIRE(jobs && floats && n_jobs > 0 && n_jobs <= kMaxDwJobs && B >= 0,
               "xdfm_dense_bwd_dw_multi_workspace_floats: bad argument (at most %d jobs)", kMaxDwJobs);
  int n_splits = 256;
  size_t per_split = 0;
  for (int k = 0; k < n_jobs; ++k) {
    XDFM_REQUIRE(jobs[k].K > 0 && jobs[k].Nout > 0, "xdfm_dense_bwd_dw_multi: bad sizes in job %d", k);
    const int s = splits_for(B, jobs[k].K, jobs[k].Nout);
    if (s < n_splits) n_splits = s;
    per_split += (((size_t)(jobs[k].K + 1) * jobs[k].Nout) + 3) & ~(size_t)3;
  }
  *floats = per_split * (size_t)n_splits;
  return XDFM_OK;
}

extern "C" int xdfm_dense_bwd_dw_multi(const xdfm_dw_job* jobs, int32_t n_jobs, int64_t B, float* part,
                                       void* stream) {
  XDFM_REQUIRE(jobs && part && n_jobs > 0 && n_jobs <= kMaxDwJobs && B >= 0,
               "xdfm_dense_bwd_dw_multi: bad argument (at most %d jobs)", kMaxDwJobs);
  hipStream_t st = (hipStream_t)stream;
  DwJobs dj;
  memset(&dj, 0, sizeof(dj));
  dj.n = n_jobs;
  dj.B = B;
  int n_splits = 256;
  for (int k = 0; k < n_jobs; ++k) {
    const xdfm_dw_job& in = jobs[k];
    XDFM_REQUIRE(in.x && in.y && in.dy && in.dw && in.db, "xdfm_dense_bwd_dw_multi: null pointer in job %d", k);
    XDFM_REQUIRE(in.K > 0 && in.Nout > 0 && (int64_t)in.K * in.Nout < ((int64_t)1 << 31),
                 "xdfm_dense_bwd_dw_multi: bad sizes in job %d", k);
    XDFM_REQUIRE(act_ok(in.act), "xdfm_dense_bwd_dw_multi: unknown activation %d", in.act);
    const int s = splits_for(B, in.K, in.Nout);
    if (s < n_splits) n_splits = s;
  }
  if (B == 0) {
    for (int k = 0; k < n_jobs; ++k) {
      XDFM_HIP(hipMemsetAsync(jobs[k].dw, 0, (size_t)jobs[k].K * jobs[k].Nout * sizeof(float), st));
      XDFM_HIP(hipMemsetAsync(jobs[k].db, 0, (size_t)jobs[k].Nout * sizeof(float), st));
    }
    return XDFM_OK;
  }
  dj.n_splits = n_splits;
  dj.rows_per_split = (int)((B + n_splits - 1) / n_splits);
  int blocks = 0, rblocks = 0;
  int64_t off = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const xdfm_dw_job& in = jobs[k];
    DwJobDev& J = dj.j[k];
    J.x = in.x;
    J.y = in.y;
    J.dy = in.dy;
    J.dw = in.dw;
    J.db = in.db;
    J.K = in.K;
    J.Nout = in.Nout;
    J.act = in.act;
    const int nt_total = (in.Nout + 15) / 16, ktiles = (in.K + 1 + 15) / 16;
    J.nt2 = nt_total >= 2;
    J.gy = (ktiles + 15) / 16;  // 4 waves x PT = 4 k tiles per block
    const int gz = J.nt2 ? (nt_total + 1) / 2 : 1;
    J.first_block = blocks;
    blocks += J.gy * gz;
    J.first_rblock = rblocks;
    rblocks += (int)(((int64_t)(in.K + 1) * in.Nout + 31) / 32);
    J.part_off = off;
    off += (int64_t)n_splits * ((((int64_t)(in.K + 1) * in.Nout) + 3) & ~(int64_t)3);
  }
  XDFM_LAUNCH("dense_bwd_dw_multi_kernel", st, dense_bwd_dw_multi_kernel, dim3((unsigned)n_splits, (unsigned)blocks),
              dim3(256), 0, dj, part);
  XDFM_LAUNCH("dense_dw_reduce_multi_kernel", st, dense_dw_reduce_multi_kernel, dim3((unsigned)rblocks), dim3(256), 0,
              dj, part);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

namespace {
int tower_dev(const xdfm_tower_desc* d, const char* who, bool grads, TowerDev* t) {
  XDFM_REQUIRE(d, "%s: null descriptor", who);
  XDFM_REQUIRE(d->n_hidden >= 1 && d->n_hidden <= kTowerMaxLayers && d->in_dim > 0,
               "%s: 1..%d hidden layers", who, kTowerMaxLayers);
  XDFM_REQUIRE(d->size[0] >= 1 && d->size[0] <= 32, "%s: the first hidden layer is at most 32 wide", who);
  XDFM_REQUIRE((int64_t)d->in_dim * d->size[0] < ((int64_t)1 << 31), "%s: weight too large", who);
  memset(t, 0, sizeof(*t));
  t->n = d->n_hidden;
  t->in_dim = d->in_dim;
  for (int l = 0; l < d->n_hidden; ++l) {
    XDFM_REQUIRE(d->size[l] >= 1 && d->size[l] <= kTowerMaxWidth, "%s: hidden layers are at most %d wide", who,
                 kTowerMaxWidth);
    XDFM_REQUIRE(act_ok(d->act[l]), "%s: unknown activation %d", who, d->act[l]);
    XDFM_REQUIRE(d->w[l] && d->b[l] && d->h[l] && (!grads || d->d_h[l]), "%s: null pointer at layer %d", who, l);
    t->size[l] = d->size[l];
    t->act[l] = d->act[l];
    t->w[l] = d->w[l];
    t->b[l] = d->b[l];
    t->h[l] = d->h[l];
    t->d_h[l] = d->d_h[l];
  }
  XDFM_REQUIRE(d->w_out && d->b_out, "%s: null output layer", who);
  t->w_out = d->w_out;
  t->b_out = d->b_out;
  return XDFM_OK;
}
}  // namespace

extern "C" int xdfm_tower_fwd(const xdfm_tower_desc* desc, const float* x, int64_t B, float* out, void* stream) {
  TowerDev t;
  int rc = tower_dev(desc, "xdfm_tower_fwd", false, &t);
  if (rc) return rc;
  XDFM_REQUIRE(x && out && B >= 0, "xdfm_tower_fwd: null argument");
  if (B == 0) return XDFM_OK;
  const dim3 grid((unsigned)((B + 15) / 16));
  if (t.size[0] > 16)
    XDFM_LAUNCH("tower_fwd_kernel<2>", stream, tower_fwd_kernel<2>, grid, dim3(256), 0, t, x, B, out);
  else
    XDFM_LAUNCH("tower_fwd_kernel<1>", stream, tower_fwd_kernel<1>, grid, dim3(256), 0, t, x, B, out);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_tower_bwd_dx(const xdfm_tower_desc* desc, const float* g, int64_t B, float* dx,
                                 int32_t accumulate_dx, void* stream) {
  TowerDev t;
  int rc = tower_dev(desc, "xdfm_tower_bwd_dx", true, &t);
  if (rc) return rc;
  XDFM_REQUIRE(g && dx && B >= 0, "xdfm_tower_bwd_dx: null argument");
  if (B == 0) return XDFM_OK;
  const dim3 grid((unsigned)((B + 15) / 16));
  if (t.size[0] > 20)
    XDFM_LAUNCH("tower_bwd_dx_kernel<8>", stream, tower_bwd_dx_kernel<8>, grid, dim3(256), 0, t, g, B,
                (int)accumulate_dx, dx);
  else
    XDFM_LAUNCH("tower_bwd_dx_kernel<5>", stream, tower_bwd_dx_kernel<5>, grid, dim3(256), 0, t, g, B,
                (int)accumulate_dx, dx);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_head_fwd(const float* pooled, const float* w, const float* bias, int64_t B,
                                 int32_t n, float* out, void* stream) {
  XDFM_REQUIRE(pooled && w && bias && out, "xdfm_cin_head_fwd: null argument");
  XDFM_REQUIRE(B >= 0 && n > 0, "xdfm_cin_head_fwd: bad sizes");
  if (B == 0) return XDFM_OK;
  XDFM_LAUNCH("cin_head_fwd_kernel", stream, cin_head_fwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
              pooled, w, bias, B, n, out);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_cin_head_bwd(const float* pooled, const float* w, const float* g, int64_t B,
                                 int32_t n, float* d_pooled, float* dw, float* db, float* part,
                                 void* stream) {
  XDFM_REQUIRE(pooled && w && g, "xdfm_cin_head_bwd: null argument");
  XDFM_REQUIRE(d_pooled || dw, "xdfm_cin_head_bwd: nothing to compute");
  XDFM_REQUIRE((dw && db && part) || (!dw && !db), "xdfm_cin_head_bwd: dw, db and part come together");
  XDFM_REQUIRE(B >= 0 && n > 0, "xdfm_cin_head_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (B > 0 && d_pooled) {
    const int64_t tot = B * n;
    XDFM_LAUNCH("cin_head_dpooled_kernel", st, cin_head_dpooled_kernel, dim3((unsigned)((tot + 255) / 256)),
                dim3(256), 0, w, g, B, n, d_pooled);
  }
  if (!dw) {
    XDFM_LAUNCH_CHECK();
    return XDFM_OK;
  }
  // dw, db: the weight-gradient path of a [n] -> [1] identity layer whose dy is g
  return xdfm_dense_bwd(pooled, w, g /* y unused for identity */, g, B, n, 1, XDFM_ACT_IDENTITY,
                        nullptr, 0, dw, db, part, stream);
}

// ------------------------------------------------------------------------ dropout
// `_active_layer` (base_model.py:324-337): with hparams.user_dropout the layer's pre-activation
// z goes through tf.nn.dropout(rate = 1 - keep) before the activation: y = act(z * m / keep),
// m ~ Bernoulli(keep).  TF's random stream cannot be reproduced; this one is a counter-based
// hash of (seed, layer, step, element), the step read from device memory so that a captured
// step draws fresh masks on every replay.  The scaled mask is kept for the backward.
namespace {

__device__ __forceinline__ uint32_t mix32(uint64_t x) {  // splitmix64 finaliser, upper half
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

__global__ __launch_bounds__(256) void dropout_act_fwd_kernel(float* __restrict__ z, float* __restrict__ mscale,
                                                              int64_t total, float keep, int act, uint64_t seed,
                                                              const float* __restrict__ step) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint64_t t = step ? (uint64_t)step[0] : 0ull;
  const uint32_t r = mix32(seed ^ (t * 0xd1342543de82ef95ull) ^ ((uint64_t)i * 0x2545f4914f6cdd1dull + 1));
  const float u = (float)(r >> 8) * (1.0f / 16777216.0f);  // [0, 1)
  const float ms = u < keep ? 1.0f / keep : 0.f;
  mscale[i] = ms;
  z[i] = activate(z[i] * ms, act);
}

__global__ __launch_bounds__(256) void dropout_act_bwd_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ dy,
                                                              const float* __restrict__ mscale, int64_t total,
                                                              int act, float* __restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  dz[i] = dy[i] * act_grad(y[i], act) * mscale[i];
}

}  // namespace

extern "C" int xdfm_dropout_act_fwd(float* z, float* mask_scale, int64_t B, int32_t n, float keep, int32_t act,
                                    uint64_t seed, const float* step, void* stream) {
  XDFM_REQUIRE(z && mask_scale && B >= 0 && n > 0, "xdfm_dropout_act_fwd: bad argument");
  XDFM_REQUIRE(keep > 0.f && keep <= 1.f, "xdfm_dropout_act_fwd: keep probability %g outside (0, 1]", (double)keep);
  XDFM_REQUIRE(act_ok(act), "xdfm_dropout_act_fwd: unknown activation %d", act);
  if (B == 0) return XDFM_OK;
  const int64_t total = B * n;
  XDFM_LAUNCH("dropout_act_fwd_kernel", stream, dropout_act_fwd_kernel, dim3((unsigned)((total + 255) / 256)),
              dim3(256), 0, z, mask_scale, total, keep, act, seed, step);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_dropout_act_bwd(const float* y, const float* dy, const float* mask_scale, int64_t B, int32_t n,
                                    int32_t act, float* dz, void* stream) {
  XDFM_REQUIRE(y && dy && mask_scale && dz && B >= 0 && n > 0, "xdfm_dropout_act_bwd: bad argument");
  XDFM_REQUIRE(act_ok(act), "xdfm_dropout_act_bwd: unknown activation %d", act);
  if (B == 0) return XDFM_OK;
  const int64_t total = B * n;
  XDFM_LAUNCH("dropout_act_bwd_kernel", stream, dropout_act_bwd_kernel, dim3((unsigned)((total + 255) / 256)),
              dim3(256), 0, y, dy, mask_scale, total, act, dz);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

XDFM_STAMP_UNIT(stamp_unit_dnn)
