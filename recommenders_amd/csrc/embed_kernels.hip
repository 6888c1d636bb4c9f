// Field-embedding lookup (+ linear + FM) forward and the deterministic
// sorted-key segment-sum backward with the fused row optimizer, for gfx950.
//
// Reference call sites replaced (recommenders/models/deeprec/models/):
//   xDeepFM.py:92-100   tf.nn.embedding_lookup_sparse(..., combiner="sum")
//   xDeepFM.py:128      sparse_dense_matmul(x, w)
//   xDeepFM.py:153-158  FM second order term
//   base_model.py:290/:286/:288 + :314-322  Adam / Adagrad / SGD apply_gradients on the table
//
// HBM layout: one shared table for all fields, fp32; row id at table + id*TS, its linear weight at
// lin_w + id*LS, optimizer slots at the same offsets from their own base pointers.  TS = D, LS = 1
// are the dense arrays table[N][D], lin_w[N]; a caller that updates sparse rows keeps a row, its
// linear weight and their slots in one record of whole 128-byte lines and passes pointers into
// it with TS = LS = the record length (every random row access costs a whole line: two lines
// per updated row instead of six).  emb_out[B][F][D].  The dnn view of a batch is CSR:
// row_ptr[B*F+1] over tokens (ids int64, weights fp32), rows = sample*F+field.
#include <hip/hip_runtime.h>

#include <cstring>

#include "drift.h"
#include "xdfm_common.h"
#include "stamp.h"

namespace {

constexpr int kWave = 64;

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  using T = float4;
};
template <>
struct Vec<2> {
  using T = float2;
};
template <>
struct Vec<1> {
  using T = float;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  using T = typename Vec<VEC>::T;
  T t = *reinterpret_cast<const T*>(p);
  memcpy(v, &t, sizeof(T));
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  using T = typename Vec<VEC>::T;
  T t;
  memcpy(&t, v, sizeof(T));
  *reinterpret_cast<T*>(p) = t;
}

// ---------------------------------------------------------------- row_ptr
__global__ void build_row_ptr_kernel(const int64_t* __restrict__ idx, int64_t nnz, int64_t n_rows,
                                     int32_t* __restrict__ row_ptr, int32_t* __restrict__ tok_row,
                                     int32_t* err_flag) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nnz) return;
  int64_t prev = (k == 0) ? -1 : idx[2 * (k - 1)];
  int64_t row = (k == nnz) ? n_rows : idx[2 * k];
  int err = 0;
  if (k < nnz && (row < 0 || row >= n_rows)) err |= XDFM_DEVERR_ROW_RANGE;
  if (row < prev) err |= XDFM_DEVERR_ROW_ORDER;
  prev = prev < -1 ? -1 : (prev > n_rows ? n_rows : prev);
  row = row < 0 ? 0 : (row > n_rows ? n_rows : row);
  if (k < nnz) tok_row[k] = (int32_t)(row >= n_rows ? n_rows - 1 : row);
  // rows in (prev, row] start at token k   (empty rows included)
  for (int64_t r = prev + 1; r <= row; ++r) row_ptr[r] = (int32_t)k;
  if (err) atomicOr(err_flag, err);
}

// ---------------------------------------------------------------- optimizer rows
struct OptDev {
  int kind;
  float lr, alpha, one_minus_b1, one_minus_b2, eps;
  const float* state;  // device {b1^t, b2^t, step size, t} or null
};

// One row's optimizer step in three parts, so that a caller can request the row early (the
// loads of a row that is about to be finished travel with the gather of its gradient).
__device__ __forceinline__ bool opt_uses_m(int kind) { return kind != XDFM_OPT_SGD; }
__device__ __forceinline__ bool opt_uses_v(int kind) {
  return kind == XDFM_OPT_ADAM || kind == XDFM_OPT_ADADELTA || kind == XDFM_OPT_FTRL;
}

template <int VEC>
struct RowVals {
  float p[VEC], m[VEC], v[VEC];
};

template <int VEC>
__device__ __forceinline__ void row_load(int kind, const float* p, const float* m, const float* v,
                                         RowVals<VEC>& r) {
  load_vec<VEC>(p, r.p);
  if (opt_uses_m(kind)) load_vec<VEC>(m, r.m);
  if (opt_uses_v(kind)) load_vec<VEC>(v, r.v);
}

template <int VEC>
__device__ __forceinline__ void row_update(const OptDev& o, RowVals<VEC>& r, const float (&g)[VEC]) {
  float(&pv)[VEC] = r.p;
  if (o.kind == XDFM_OPT_ADAM) {
    const float alpha = o.state ? o.state[2] : o.alpha;
    float(&mv)[VEC] = r.m;
    float(&vv)[VEC] = r.v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mv[i] = mv[i] + (g[i] - mv[i]) * o.one_minus_b1;
      vv[i] = vv[i] + (g[i] * g[i] - vv[i]) * o.one_minus_b2;
      pv[i] = pv[i] - (mv[i] * alpha) / (sqrtf(vv[i]) + o.eps);
    }
  } else if (o.kind == XDFM_OPT_ADAGRAD) {
    float(&av)[VEC] = r.m;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      av[i] = av[i] + g[i] * g[i];
      pv[i] = pv[i] - (g[i] * o.lr) / sqrtf(av[i]);
    }
  } else if (o.kind == XDFM_OPT_RMSPROP) {
    float(&av)[VEC] = r.m;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      av[i] = av[i] + (g[i] * g[i] - av[i]) * o.one_minus_b2;
      pv[i] = pv[i] - (g[i] * o.lr) / sqrtf(av[i] + o.eps);
    }
  } else if (o.kind == XDFM_OPT_ADADELTA) {
    float(&av)[VEC] = r.m;
    float(&uv)[VEC] = r.v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      av[i] = av[i] * (1.f - o.one_minus_b2) + g[i] * g[i] * o.one_minus_b2;
      const float upd = sqrtf(uv[i] + o.eps) / sqrtf(av[i] + o.eps) * g[i];
      uv[i] = uv[i] * (1.f - o.one_minus_b2) + upd * upd * o.one_minus_b2;
      pv[i] = pv[i] - upd * o.lr;
    }
  } else if (o.kind == XDFM_OPT_FTRL) {
    float(&av)[VEC] = r.m;
    float(&lv)[VEC] = r.v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float nw = av[i] + g[i] * g[i];
      lv[i] = lv[i] + g[i] - (sqrtf(nw) - sqrtf(av[i])) / o.lr * pv[i];
      pv[i] = -lv[i] / (sqrtf(nw) / o.lr);
      av[i] = nw;
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) pv[i] = pv[i] - g[i] * o.lr;
  }
}

template <int VEC>
__device__ __forceinline__ void row_store(int kind, float* p, float* m, float* v, const RowVals<VEC>& r) {
  if (opt_uses_m(kind)) store_vec<VEC>(m, r.m);
  if (opt_uses_v(kind)) store_vec<VEC>(v, r.v);
  store_vec<VEC>(p, r.p);
}

// ---------------------------------------------------------------- row ageing
// The reference's table gradient is dense (base_model.py:121-149 put regulariser ops on the table even
// at coefficient 0), so TF's Adam moves EVERY row EVERY step: a row without a token in
// the batch still takes m <- b1 m, v <- b2 v, p <- p - lr_t m / (sqrt(v) + eps).  The sparse-row update
// skips those steps; row_step[id] records the step a row's stored (p, m, v) are current for, and whoever
// reads the row applies the G = done - row_step steps it missed in closed form (drift.h):
//   * the forward gather and the owner-side row gather form the aged row in registers and write nothing;
//   * the row update ages (p, m, v) in the registers it loads them into anyway, then applies the
//     gradient: one read-modify-write per updated row and step, rows absent from the batch are not
//     touched at all;
//   * xdfm_embed_catchup writes aged rows back (before checkpoints / table exports).
// RMSProp / Adadelta only decay their accumulators (no gradient, no movement); FTRL re-derives the row
// from its slots (TF does so every step: rows that never had a gradient become 0); SGD and Adagrad
// have nothing to age.
struct AgeDev {
  int32_t* row_step;  // null: rows are always current
  int64_t SS;         // 4-byte words between consecutive entries of row_step
  const float *slot_m, *slot_v, *lin_m, *lin_v;  // the slots as the gathers read them
  const float4* tab;   // drift table (Adam), null for the other optimizers
  const float* state;  // opt_state: [3] = optimizer steps applied so far
  int step_off;        // rows are aged to (int)state[3] + step_off steps
  int kind;
  int packed_tail;  // lin_w, lin_m, lin_v, row_step are the four words of one aligned 16-byte group per row
  float lr, ebar, lg_b1, lg_b2;
};

__device__ __forceinline__ bool age_needed(int kind, int last, int G) {
  // last == 0: never updated; Adam's / Adadelta's slots are still zero: no drift (RMSProp's rms starts
  // at 1 and decays from the first step on, FTRL rewrites the row at its first step)
  return G > 0 && (last > 0 || kind == XDFM_OPT_RMSPROP || kind == XDFM_OPT_FTRL);
}

__device__ __forceinline__ float4 drift_coeffs(const AgeDev& a, const float4* cur_s, int last, int G) {
  if (a.kind != XDFM_OPT_ADAM) return make_float4(0.f, 0.f, 0.f, 0.f);
  if (G < xdfm::kDriftG) return cur_s[G];
  return a.tab[2 * xdfm::kDriftG + (last < xdfm::kDriftS ? last : xdfm::kDriftS)];
}

template <int VEC>
__device__ __forceinline__ void age_vals(const AgeDev& a, const float4 cf, RowVals<VEC>& r, int G) {
  if (a.kind == XDFM_OPT_ADAM) {
    // the hardware's square root and reciprocal (1 ulp each): the drift is orders of magnitude below the
    // parameter, the difference to the correctly rounded forms never reaches the parameter's last bits
    const float pw1 = exp2f((float)G * a.lg_b1), pw2 = exp2f((float)G * a.lg_b2);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float y = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(r.v[i]) + a.ebar);
      r.p[i] = r.p[i] - r.m[i] * y * (cf.x - y * (cf.y - y * cf.z));
      r.m[i] = r.m[i] * pw1;
      r.v[i] = r.v[i] * pw2;
    }
  } else if (a.kind == XDFM_OPT_RMSPROP) {
    const float pw = exp2f((float)G * a.lg_b2);  // ms <- ms + (0 - ms)(1 - decay), G times
#pragma unroll
    for (int i = 0; i < VEC; ++i) r.m[i] = r.m[i] * pw;
  } else if (a.kind == XDFM_OPT_ADADELTA) {
    const float pw = exp2f((float)G * a.lg_b2);  // accum and accum_update both decay by rho per step
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      r.m[i] = r.m[i] * pw;
      r.v[i] = r.v[i] * pw;
    }
  } else if (a.kind == XDFM_OPT_FTRL) {
    // TF's dense ApplyFtrl rewrites var = -linear / quadratic every step, gradient or not: a row that
    // never had a gradient (linear = 0) becomes 0 at the first step; an updated row is a fixed point
#pragma unroll
    for (int i = 0; i < VEC; ++i) r.p[i] = -r.v[i] / (sqrtf(r.m[i]) / a.lr);
  }
}

// A table row with what ageing it needs, requested in one go and resolved when the values are used.
template <int VEC>
struct RowReq {
  RowVals<VEC> r;
  RowVals<1> l;  // the row's linear weight and its slots (lanes with want_lin)
  int last;
};

template <int VEC, bool AGE>
__device__ __forceinline__ void row_request(const float* table, const float* lin_w, int64_t TS, int64_t LS,
                                            const AgeDev& ag, int64_t id, int c, bool want_lin, RowReq<VEC>& q) {
  const int64_t off = id * TS + c * VEC;
  load_vec<VEC>(table + off, q.r.p);
  q.l.p[0] = q.l.m[0] = q.l.v[0] = 0.f;
  q.last = 0;
  if (AGE) {
    if (opt_uses_m(ag.kind)) load_vec<VEC>(ag.slot_m + off, q.r.m);
    if (opt_uses_v(ag.kind)) load_vec<VEC>(ag.slot_v + off, q.r.v);
    if (ag.packed_tail) {
      const float4 t = *reinterpret_cast<const float4*>(lin_w + id * LS);
      q.l.p[0] = t.x;
      q.l.m[0] = t.y;
      q.l.v[0] = t.z;
      q.last = __float_as_int(t.w);
    } else {
      q.last = ag.row_step[id * ag.SS];
      if (want_lin) {
        q.l.p[0] = lin_w[id * LS];
        if (opt_uses_m(ag.kind)) q.l.m[0] = ag.lin_m[id * LS];
        if (opt_uses_v(ag.kind)) q.l.v[0] = ag.lin_v[id * LS];
      }
    }
  } else if (want_lin) {
    q.l.p[0] = lin_w[id * LS];
  }
}

template <int VEC, bool AGE>
__device__ __forceinline__ void row_resolve(const AgeDev& ag, const float4* cur_s, int done, bool want_lin,
                                            RowReq<VEC>& q) {
  if (!AGE) return;
  const int G = done - q.last;
  if (age_needed(ag.kind, q.last, G)) {
    const float4 cf = drift_coeffs(ag, cur_s, q.last, G);
    age_vals<VEC>(ag, cf, q.r, G);
    if (want_lin) age_vals<1>(ag, cf, q.l, G);
  }
}

// ---------------------------------------------------------------- forward
// One wave per sample.  LPR = D/VEC lanes cover one output row (VEC floats each), so a
// wave gathers 64/LPR rows per pass and writes them as one contiguous span of emb_out.
// The sample's per-dimension sums (FM) and its linear term are reduced through LDS in a
// fixed order, so the result does not depend on scheduling.
// AGE: rows are read together with their slots and row_step and aged in registers (see above).
struct FwdArgs {
  const float *table, *lin_w;
  int64_t TS, LS, N, B, nnz;
  int D, F;
  const int32_t* row_ptr;
  const int64_t* ids;
  const float* weights;
  float *emb_out, *lin_out, *fm_out, *fm_sum;
  int32_t* err_flag;
  AgeDev age;
};

template <int VEC, bool AGE>
__global__ __launch_bounds__(256) void embed_fwd_kernel(FwdArgs a) {
  xdfm::Stamp stamp_(xdfm::kStampEmbedFwd);
  __shared__ float red_s[4][kWave][VEC];
  __shared__ float red_q[4][kWave][VEC];
  __shared__ float red_l[4][kWave];
  __shared__ float4 cur_s[AGE ? xdfm::kDriftG : 1];
  const float* __restrict__ table = a.table;
  const float* __restrict__ lin_w = a.lin_w;
  const int64_t TS = a.TS, LS = a.LS, N = a.N, B = a.B, nnz = a.nnz;
  const int D = a.D, F = a.F;
  const int32_t* __restrict__ row_ptr = a.row_ptr;
  const int64_t* __restrict__ ids = a.ids;
  const float* __restrict__ weights = a.weights;
  float* __restrict__ emb_out = a.emb_out;
  float* lin_out = a.lin_out;
  float* fm_out = a.fm_out;
  float* fm_sum = a.fm_sum;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  const int LPR = D / VEC;
  const int RPP = kWave / LPR;  // rows per pass
  const int g = lane / LPR;
  const int c = lane - g * LPR;
  const bool lane_on = (g < RPP) && (b < B);
  const bool want_fm = (fm_out != nullptr) || (fm_sum != nullptr);
  // the linear weight rides with lane c == 0 of a row (under ageing also when only its row_step is wanted)
  const bool want_lin = c == 0 && lin_w != nullptr;
  int done = 0;
  if (AGE) {
    done = (int)a.age.state[3] + a.age.step_off;
    if (a.age.tab) cur_s[threadIdx.x] = a.age.tab[(done & 1) * xdfm::kDriftG + threadIdx.x];
    __syncthreads();
  }

  float s_acc[VEC], q_acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s_acc[v] = q_acc[v] = 0.f;
  float lin_acc = 0.f;
  int err = 0;

  if (lane_on) {
    // Up to PU rows per lane group are handled together: their CSR bounds, then their first
    // tokens, then the table rows of those tokens are each requested in one go, so a sample
    // costs three memory round trips instead of three per row.  Further tokens of a multi-hot
    // row follow in column order (the sums stay bit-identical to the sequential form).
    // Under ageing a row is four loads, not one: the CSR bounds and tokens of PU rows are still requested
    // together, their table rows RB at a time (2 rows in flight per lane group keep the kernel at ~110
    // registers; 4 take 161 registers and 32 us instead of 26 on the bench workload).
    constexpr int PU = 4;
    constexpr int RB = AGE ? 2 : 4;
    for (int f0 = g; f0 < F; f0 += PU * RPP) {
      int beg[PU], end[PU];
      int64_t id0[PU];
      float w0[PU];
      bool has[PU], ok0[PU];
      // The CSR bounds and, speculatively, the token a single-hot batch keeps at position r (one
      // token per row: row_ptr is the identity) are requested together: two memory round trips
      // instead of three when the guess holds, which it does for every row of Criteo-shaped data.
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int f = f0 + u * RPP;
        const int64_t r = b * F + (f < F ? f : F - 1);
        beg[u] = row_ptr[r];
        end[u] = row_ptr[r + 1];
        const int64_t guess = r < nnz ? r : 0;
        id0[u] = nnz > 0 ? ids[guess] : 0;  // a batch without tokens has no array to guess in
        w0[u] = nnz > 0 ? weights[guess] : 0.f;
        if (f >= F) end[u] = beg[u];  // no such row: behaves as empty, nothing is stored
      }
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        has[u] = end[u] > beg[u];
        const int f = f0 + u * RPP;
        const int64_t r = b * F + (f < F ? f : F - 1);
        if (has[u] && beg[u] != r) {  // not where the guess looked
          id0[u] = ids[beg[u]];
          w0[u] = weights[beg[u]];
        }
        ok0[u] = has[u] && (uint64_t)id0[u] < (uint64_t)N;
      }
#pragma unroll
      for (int u0 = 0; u0 < PU; u0 += RB) {
        if (f0 + u0 * RPP >= F) break;  // (wave-uniform per lane group; lanes past F idle anyway)
        RowReq<VEC> rq[RB];
#pragma unroll
        for (int v = 0; v < RB; ++v)
          row_request<VEC, AGE>(table, lin_w, TS, LS, a.age, ok0[u0 + v] ? id0[u0 + v] : 0, c, want_lin, rq[v]);
#pragma unroll
        for (int v = 0; v < RB; ++v) {
          const int u = u0 + v;
          const int f = f0 + u * RPP;
          if (f >= F) continue;
          const int64_t r = b * F + f;
          float acc[VEC];
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
          if (has[u]) {
            if (ok0[u]) {
              row_resolve<VEC, AGE>(a.age, cur_s, done, want_lin, rq[v]);
#pragma unroll
              for (int k = 0; k < VEC; ++k) {
                const float t = w0[u] * rq[v].r.p[k];  // one rounding (TF: embeddings *= weights)
                acc[k] = t;
                q_acc[k] += t * t;
              }
              if (want_lin) lin_acc += w0[u] * rq[v].l.p[0];
            } else {
              err |= XDFM_DEVERR_ID_RANGE;
            }
          }
          bool first = !(has[u] && ok0[u]);  // no valid token summed yet
          for (int k = beg[u] + 1; k < end[u]; ++k) {
            const int64_t id = ids[k];
            const float w = weights[k];
            if ((uint64_t)id >= (uint64_t)N) {
              err |= XDFM_DEVERR_ID_RANGE;
              continue;
            }
            RowReq<VEC> q;
            row_request<VEC, AGE>(table, lin_w, TS, LS, a.age, id, c, want_lin, q);
            row_resolve<VEC, AGE>(a.age, cur_s, done, want_lin, q);
#pragma unroll
            for (int kk = 0; kk < VEC; ++kk) {
              const float t = w * q.r.p[kk];
              acc[kk] = first ? t : acc[kk] + t;  // segment_sum in column order
              q_acc[kk] += t * t;
            }
            first = false;
            if (want_lin) lin_acc += w * q.l.p[0];
          }
          store_vec<VEC>(emb_out + r * D + c * VEC, acc);
#pragma unroll
          for (int k = 0; k < VEC; ++k) s_acc[k] += acc[k];
        }
      }
    }
  }
  if (err) atomicOr(a.err_flag, err);
  if (!want_fm && !lin_out) return;  // uniform over the block

#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    red_s[wave][lane][v] = lane_on ? s_acc[v] : 0.f;
    red_q[wave][lane][v] = lane_on ? q_acc[v] : 0.f;
  }
  red_l[wave][lane] = (lane_on && c == 0) ? lin_acc : 0.f;
  __syncthreads();
  const bool valid = b < B;  // no early exit: every wave takes part in the barriers below
  // lanes 0..LPR-1: sum their column chunk over the row groups, in group order
  float part = 0.f;
  if (valid && lane < LPR) {
    float s[VEC], q[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = q[v] = 0.f;
    for (int gg = 0; gg < RPP; ++gg) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        s[v] += red_s[wave][gg * LPR + lane][v];
        q[v] += red_q[wave][gg * LPR + lane][v];
      }
    }
    if (fm_sum) store_vec<VEC>(fm_sum + b * D + lane * VEC, s);
#pragma unroll
    for (int v = 0; v < VEC; ++v) part += s[v] * s[v] - q[v];
  }
  __syncthreads();
  red_s[wave][lane][0] = part;
  __syncthreads();
  if (valid && lane == 0) {
    if (fm_out) {
      float tot = 0.f;
      for (int cc = 0; cc < LPR; ++cc) tot += red_s[wave][cc][0];
      fm_out[b] = 0.5f * tot;
    }
    if (lin_out) {
      float tot = 0.f;
      for (int gg = 0; gg < RPP; ++gg) tot += red_l[wave][gg * LPR];
      lin_out[b] = tot;
    }
  }
}

// ---------------------------------------------------------------- backward
struct BwdArgs {
  float *table, *slot_m, *slot_v, *lin_w, *lin_m, *lin_v, *grad_table, *grad_lin;
  int64_t N, B, nnz;
  int64_t TS, LS;  // row strides (floats) of table / slots and of lin_w / its slots
  int64_t GS, GLS;  // strides of the gradient rows / scalars this call writes (modes 1, 2)
  int64_t RS, RLS;  // strides of the d_emb rows / g_logit scalars it reads
  int D, F;
  const int32_t *sorted_ids, *sorted_tok, *tok_row;
  const float *weights, *d_emb, *d_emb2, *g_logit, *fm_sum;
  int use_fm;
  OptDev opt;
  AgeDev age;
  float* pieces;  // head[nchunks][D], tail[nchunks][D], then head / tail [nchunks] of the linear and of the FM scalar
};

// A row about to be finished: requested early (with the gather of its gradient), completed when the
// segment's sum is known.  The gradient of a segment (one id) is
//     sum_j (w_j d_emb[row_j] + gw_j fm_sum[b_j])  -  E[id] * sum_j gw_j w_j,       gw_j = g[b_j] w_j
// (the FM term of every token of a segment has the same table row E[id] as its factor: it is taken
// out of the sum, so no token reads the table -- the row is read once, where it is updated, and under
// ageing it is the aged row, the one the forward pass used).
template <int VEC>
struct RowState {
  RowVals<VEC> rv;
  RowVals<1> rl;
  int last;
};

template <int VEC, int MODE, bool AGE>
__device__ __forceinline__ void finish_request(const BwdArgs& a, int key, int c, RowState<VEC>& s) {
  s.last = 0;
  if (MODE == 0) {
    const int64_t off = (int64_t)key * a.TS + c * VEC;
    const int64_t lo = (int64_t)key * a.LS;
    row_load<VEC>(a.opt.kind, a.table + off, a.slot_m + off, a.slot_v + off, s.rv);
    if (AGE && a.age.packed_tail) {
      const float4 t = *reinterpret_cast<const float4*>(a.lin_w + lo);
      s.rl.p[0] = t.x;
      s.rl.m[0] = t.y;
      s.rl.v[0] = t.z;
      s.last = __float_as_int(t.w);
    } else {
      if (a.lin_w && c == 0) row_load<1>(a.opt.kind, a.lin_w + lo, a.lin_m + lo, a.lin_v + lo, s.rl);
      if (AGE) s.last = a.age.row_step[(int64_t)key * a.age.SS];
    }
  } else if (a.use_fm) {
    load_vec<VEC>(a.table + (int64_t)key * a.TS + c * VEC, s.rv.p);
  }
}

template <int VEC, int MODE, bool AGE>
__device__ __forceinline__ void finish_complete(const BwdArgs& a, const float4* cur_s, int done, int key, int c,
                                                RowState<VEC>& s, float (&g)[VEC], float lsum, float fsum) {
  if (MODE == 0) {
    const bool lin = a.lin_w != nullptr && c == 0;
    if (AGE) {
      const int G = done - s.last;
      if (age_needed(a.age.kind, s.last, G)) {
        const float4 cf = drift_coeffs(a.age, cur_s, s.last, G);
        age_vals<VEC>(a.age, cf, s.rv, G);
        if (lin) age_vals<1>(a.age, cf, s.rl, G);
      }
    }
    if (a.use_fm) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) g[k] = g[k] - s.rv.p[k] * fsum;
    }
    const int64_t off = (int64_t)key * a.TS + c * VEC;
    const int64_t lo = (int64_t)key * a.LS;
    row_update<VEC>(a.opt, s.rv, g);
    row_store<VEC>(a.opt.kind, a.table + off, a.slot_m + off, a.slot_v + off, s.rv);
    if (lin) {
      const float g1[1] = {lsum};
      row_update<1>(a.opt, s.rl, g1);
    }
    if (AGE && a.age.packed_tail) {
      if (c == 0)
        *reinterpret_cast<float4*>(a.lin_w + lo) =
            make_float4(s.rl.p[0], s.rl.m[0], s.rl.v[0], __int_as_float(done + 1));
    } else {
      if (lin) row_store<1>(a.opt.kind, a.lin_w + lo, a.lin_m + lo, a.lin_v + lo, s.rl);
      if (AGE && c == 0) a.age.row_step[(int64_t)key * a.age.SS] = done + 1;
    }
  } else {
    if (a.use_fm) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) g[k] = g[k] - s.rv.p[k] * fsum;
    }
    const int64_t off = (int64_t)key * a.GS + c * VEC;
    store_vec<VEC>(a.grad_table + off, g);
    if (a.grad_lin && c == 0) a.grad_lin[(int64_t)key * a.GLS] = lsum;
  }
}

// Segment sum over the id-sorted token list, one wave per chunk of 64 sorted positions.
//  * lane = position: loads key / token / row / weight once, and runs a 6-step segmented
//    scan of the linear-part scalar g*w and of the FM scalar g*w*w.
//  * the D-wide contributions are handled R = 64/LPR positions per pass (LPR = D/VEC lanes
//    own one row, VEC floats each): gather, then a log2(R)-step segmented inclusive scan
//    across the lane groups with cross-lane shuffles; a carry links the passes.
//  * a segment that lies inside the chunk is finished on the spot (optimizer fused).  The
//    partial sum of a segment that comes in from the previous chunk goes to head[chunk], of
//    one that starts here and runs on into the next chunk to tail[chunk]; the second kernel
//    adds tail[c0] + head[c0+1] + ... in chunk order for each such segment.
// Every sum has one fixed association: bit-identical from run to run, no float atomics.
constexpr int kChunk = 64;

template <int VEC>
__device__ __forceinline__ void shfl_vec(const float (&v)[VEC], int src, float (&o)[VEC]) {
#pragma unroll
  for (int k = 0; k < VEC; ++k) o[k] = __shfl(v[k], src);
}

struct Pieces {
  float *head_vec, *tail_vec, *head_lin, *tail_lin, *head_fm, *tail_fm;
};
__device__ __forceinline__ Pieces pieces_of(const BwdArgs& a, int64_t nchunks) {
  Pieces p;
  p.head_vec = a.pieces;
  p.tail_vec = p.head_vec + nchunks * a.D;
  p.head_lin = p.tail_vec + nchunks * a.D;
  p.tail_lin = p.head_lin + nchunks;
  p.head_fm = p.tail_lin + nchunks;
  p.tail_fm = p.head_fm + nchunks;
  return p;
}

template <int VEC, int MODE, bool AGE>
__global__ __launch_bounds__(256) void embed_bwd_kernel(BwdArgs a) {
  xdfm::Stamp stamp_(MODE == 0 ? xdfm::kStampEmbedBwd : xdfm::kStampEmbedScatter);
  __shared__ float4 cur_s[AGE ? xdfm::kDriftG : 1];
  int done = 0;
  if (AGE) {
    done = (int)a.age.state[3] + a.age.step_off;
    if (a.age.tab) cur_s[threadIdx.x] = a.age.tab[(done & 1) * xdfm::kDriftG + threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t p0 = chunk * kChunk;
  if (p0 >= a.nnz) return;  // wave-uniform; no barrier below
  const int LPR = a.D / VEC;
  const int R = 64 / LPR;
  const int g = lane / LPR;
  const int c = lane - g * LPR;
  const int nvalid = (int)((a.nnz - p0 < kChunk) ? (a.nnz - p0) : kChunk);
  const bool use_lin = a.lin_w != nullptr;
  const bool need_g = a.use_fm || use_lin;

  // ---- per-position metadata
  const bool pv = lane < nvalid;
  const int64_t p = p0 + (pv ? lane : 0);
  const int key = pv ? a.sorted_ids[p] : -1;
  const int tok = a.sorted_tok ? a.sorted_tok[p] : (int)p;  // null: tok_row / weights come sorted
  const int row = a.tok_row[tok];
  const float w = pv ? a.weights[tok] : 0.f;
  float gw = 0.f;
  if (need_g) gw = pv ? a.g_logit[(int64_t)(row / a.F) * a.RLS] * w : 0.f;
  const int prev_key = (p0 > 0) ? a.sorted_ids[p0 - 1] : -1;
  const int next_key = (p0 + kChunk < a.nnz) ? a.sorted_ids[p0 + kChunk] : -2;
  const int first_key = __shfl(key, 0);
  const int last_key = __shfl(key, nvalid - 1);
  const bool cont_prev = prev_key == first_key;
  const bool cont_next = next_key == last_key;
  const int key_after = __shfl_down(key, 1);  // key of position lane+1 (garbage for lane 63)

  // ---- scalars: segmented inclusive scans of g*w (linear part) and g*w*w (FM) over the 64 positions
  float lsum = gw, fsum = gw * w;
  if (need_g) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(lsum, d);
      const float of = __shfl_up(fsum, d);
      const int ko = __shfl_up(key, d);
      if (lane >= d && ko == key) {
        lsum += o;
        fsum += of;
      }
    }
  }

  const int64_t nchunks = (a.nnz + kChunk - 1) / kChunk;
  const Pieces pc = pieces_of(a, nchunks);

  float carry[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) carry[k] = 0.f;
  int carry_key = -3;
  const int npass = (nvalid + R - 1) / R;
  for (int t = 0; t < npass; ++t) {
    const int q = t * R + g;                 // position of this lane group
    const bool on = g < R && q < nvalid;
    const int qs = on ? q : 0;
    const int key_raw = __shfl(key, qs);  // every lane takes part in the shuffle
    const int key_q = on ? key_raw : -1;
    const int row_q = __shfl(row, qs);
    const float w_q = __shfl(w, qs);
    const float gw_q = __shfl(gw, qs);
    const int keyn_q = __shfl(key_after, qs);
    const float lsum_q = __shfl(lsum, qs);
    const float fsum_q = __shfl(fsum, qs);
    // what becomes of this position's running sum is known from the keys alone: a row that will
    // be finished in this pass is requested now, together with the gather of its gradient
    const bool chunk_end = q == nvalid - 1;
    const bool seg_end = on && (chunk_end || keyn_q != key_q);
    const bool to_head = cont_prev && key_q == first_key;
    const bool to_tail = chunk_end && cont_next;
    const bool finish = seg_end && !to_head && !to_tail;
    RowState<VEC> rs;
    if (finish) finish_request<VEC, MODE, AGE>(a, key_q, c, rs);
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.f;
    if (on) {
      float de[VEC];
      const int64_t ds = a.RS;
      load_vec<VEC>(a.d_emb + (int64_t)row_q * ds + c * VEC, de);
      if (a.d_emb2) {  // a second producer's gradient (the DNN tower's), summed on the fly
        float de2[VEC];
        load_vec<VEC>(a.d_emb2 + (int64_t)row_q * ds + c * VEC, de2);
#pragma unroll
        for (int k = 0; k < VEC; ++k) de[k] += de2[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[k] = w_q * de[k];
      if (a.use_fm) {
        float sfm[VEC];
        load_vec<VEC>(a.fm_sum + (int64_t)(row_q / a.F) * a.D + c * VEC, sfm);
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = fmaf(gw_q, sfm[k], v[k]);
      }
      if (g == 0 && carry_key == key_q) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] += carry[k];
      }
    }
    // segmented inclusive scan across the lane groups
    for (int d = 1; d < R; d <<= 1) {
      float o[VEC];
      const int src = lane - d * LPR;
      shfl_vec<VEC>(v, src < 0 ? lane : src, o);
      const int ko = __shfl(key_q, src < 0 ? lane : src);
      if (on && g >= d && ko == key_q) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] += o[k];
      }
    }
    // carry for the next pass = the last group's running sum
    {
      const int src = (R - 1) * LPR + c;
      shfl_vec<VEC>(v, src < 64 ? src : lane, carry);
      carry_key = __shfl(key_q, (R - 1) * LPR);
    }
    if (seg_end) {
      if (to_head) {
        store_vec<VEC>(pc.head_vec + chunk * a.D + c * VEC, v);
        if (c == 0) {
          pc.head_lin[chunk] = lsum_q;
          pc.head_fm[chunk] = fsum_q;
        }
      } else if (to_tail) {
        store_vec<VEC>(pc.tail_vec + chunk * a.D + c * VEC, v);
        if (c == 0) {
          pc.tail_lin[chunk] = lsum_q;
          pc.tail_fm[chunk] = fsum_q;
        }
      } else {
        finish_complete<VEC, MODE, AGE>(a, cur_s, done, key_q, c, rs, v, lsum_q, fsum_q);
      }
    }
  }
}

// One wave per chunk boundary.  The wave whose boundary is the first one a segment crosses
// adds the segment's pieces in chunk order and finishes the row.
template <int VEC, int MODE, bool AGE>
__global__ __launch_bounds__(256) void embed_bwd_long_kernel(BwdArgs a) {
  xdfm::Stamp stamp_(xdfm::kStampEmbedBwdLong);
  __shared__ float4 cur_s[AGE ? xdfm::kDriftG : 1];
  int done = 0;
  if (AGE) {
    done = (int)a.age.state[3] + a.age.step_off;
    if (a.age.tab) cur_s[threadIdx.x] = a.age.tab[(done & 1) * xdfm::kDriftG + threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t c1 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) + 1;  // boundary between c1-1 and c1
  const int64_t nchunks = (a.nnz + kChunk - 1) / kChunk;
  if (c1 >= nchunks) return;
  const int key = a.sorted_ids[c1 * kChunk];
  if (a.sorted_ids[c1 * kChunk - 1] != key) return;                           // no segment crosses here
  if (c1 >= 2 && a.sorted_ids[(c1 - 1) * kChunk - 1] == key) return;          // an earlier boundary owns it
  const int LPR = a.D / VEC;
  const int R = 64 / LPR;
  const int g = lane / LPR;
  const int c = lane - g * LPR;
  RowState<VEC> rs;
  if (g == 0) finish_request<VEC, MODE, AGE>(a, key, c, rs);
  // last chunk of the segment: first chunk ce >= c1 whose successor does not start with key
  int64_t ce = c1;
  for (;;) {
    const int64_t cc = ce + lane;
    const bool cont = (cc + 1 < nchunks) && a.sorted_ids[(cc + 1) * kChunk] == key;
    const unsigned long long m = __ballot(!cont);
    if (m) {
      ce += __ffsll((long long)m) - 1;
      break;
    }
    ce += 64;
  }
  const Pieces pc = pieces_of(a, nchunks);
  // group g adds pieces g, g+R, ... (piece 0 = tail[c1-1], piece k = head[c1+k-1]) in order
  const int64_t npieces = ce - c1 + 2;
  float v[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) v[k] = 0.f;
  if (g < R) {
    for (int64_t k = g; k < npieces; k += R) {
      float pv[VEC];
      const float* src = (k == 0) ? pc.tail_vec + (c1 - 1) * a.D : pc.head_vec + (c1 + k - 1) * a.D;
      load_vec<VEC>(src + c * VEC, pv);
#pragma unroll
      for (int kx = 0; kx < VEC; ++kx) v[kx] += pv[kx];
    }
  }
  // add the groups in a fixed tree
  for (int d = 1; d < R; d <<= 1) {
    float o[VEC];
    const int src = lane + d * LPR;
    shfl_vec<VEC>(v, src < 64 ? src : lane, o);
    if (g + d < R && (g & (2 * d - 1)) == 0) {
#pragma unroll
      for (int kx = 0; kx < VEC; ++kx) v[kx] += o[kx];
    }
  }
  float lsum = 0.f, fsum = 0.f;
  if (a.lin_w != nullptr || a.use_fm) {
    for (int64_t k = lane; k < npieces; k += 64) {
      lsum += (k == 0) ? pc.tail_lin[c1 - 1] : pc.head_lin[c1 + k - 1];
      fsum += (k == 0) ? pc.tail_fm[c1 - 1] : pc.head_fm[c1 + k - 1];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lsum += __shfl_down(lsum, d);
      fsum += __shfl_down(fsum, d);
    }
    lsum = __shfl(lsum, 0);
    fsum = __shfl(fsum, 0);
  }
  if (g == 0) finish_complete<VEC, MODE, AGE>(a, cur_s, done, key, c, rs, v, lsum, fsum);
}

__global__ void plan_gather_kernel(const int32_t* __restrict__ sorted_tok, const int32_t* __restrict__ tok_row,
                                   const float* __restrict__ weights, int64_t nnz,
                                   int32_t* __restrict__ row_sorted, float* __restrict__ w_sorted) {
  xdfm::Stamp stamp_(xdfm::kStampPlanGather);
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  const int tok = sorted_tok[p];
  row_sorted[p] = tok_row[tok];
  w_sorted[p] = weights[tok];
}

__global__ void clear_rows_kernel(float* __restrict__ grad_table, float* __restrict__ grad_lin,
                                  int D, const int32_t* __restrict__ sorted_ids, int64_t nnz) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / D;
  const int d = (int)(t - i * D);
  if (i >= nnz) return;
  const int id = sorted_ids[i];
  if (i > 0 && sorted_ids[i - 1] == id) return;
  if (grad_table) grad_table[(int64_t)id * D + d] = 0.f;
  if (grad_lin && d == 0) grad_lin[id] = 0.f;
}

int pick_vec(int D) { return (D % 4 == 0) ? 4 : (D % 2 == 0 ? 2 : 1); }

int make_opt(const xdfm_opt_args* opt, OptDev* o) {
  XDFM_REQUIRE(opt, "optimizer arguments are null");
  XDFM_REQUIRE(opt->kind >= XDFM_OPT_SGD && opt->kind <= XDFM_OPT_FTRL,
               "unknown optimizer kind %d", opt->kind);
  o->kind = opt->kind;
  o->lr = opt->lr;
  o->eps = opt->eps;
  o->one_minus_b1 = 1.f - opt->beta1;
  o->one_minus_b2 = 1.f - opt->beta2;
  // TF ApplyAdam: alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t), all in fp32
  o->state = opt->opt_state;
  o->alpha = (opt->kind == XDFM_OPT_ADAM)
                 ? opt->lr * sqrtf(1.f - opt->beta2_power) / (1.f - opt->beta1_power)
                 : 0.f;
  return XDFM_OK;
}

// Device view of xdfm_row_age.  lin_w / LS: the linear weights the call itself addresses (null: none);
// need_slots: the call reads the slots through the struct (the gathers; the row update has them as
// arguments of its own).
int make_age(const xdfm_row_age* age, const float* lin_w, int64_t LS, bool need_slots, AgeDev* a) {
  memset(a, 0, sizeof(*a));
  if (!age) return XDFM_OK;
  XDFM_REQUIRE(age->row_step && age->opt && age->opt->opt_state,
               "row ageing needs row_step and the optimizer arguments with opt_state");
  const xdfm_opt_args* opt = age->opt;
  XDFM_REQUIRE(opt->kind >= XDFM_OPT_SGD && opt->kind <= XDFM_OPT_FTRL, "unknown optimizer kind %d", opt->kind);
  XDFM_REQUIRE(opt->kind != XDFM_OPT_ADAM || age->drift_tab, "row ageing under Adam needs the drift table");
  XDFM_REQUIRE(age->step_offset >= -1 && age->step_offset <= 0, "row ageing: step_offset must be 0 or -1");
  const bool um = opt->kind != XDFM_OPT_SGD, uv = xdfm::opt_needs_v(opt->kind);
  if (need_slots) {
    XDFM_REQUIRE((!um || age->slot_m) && (!uv || age->slot_v), "row ageing: table slot missing");
    XDFM_REQUIRE(!lin_w || ((!um || age->lin_m) && (!uv || age->lin_v)), "row ageing: lin slot missing");
  }
  a->row_step = age->row_step;
  a->SS = age->step_stride ? age->step_stride : 1;
  a->slot_m = age->slot_m;
  a->slot_v = age->slot_v;
  a->lin_m = age->lin_m;
  a->lin_v = age->lin_v;
  a->tab = opt->kind == XDFM_OPT_ADAM ? reinterpret_cast<const float4*>(age->drift_tab) : nullptr;
  a->state = opt->opt_state;
  a->step_off = age->step_offset;
  a->kind = opt->kind;
  a->lr = opt->lr;
  a->ebar = opt->eps * exp2f(-(float)xdfm::kDriftC * 0.5f * log2f(opt->beta2));
  a->lg_b1 = opt->beta1 > 0.f ? log2f(opt->beta1) : 0.f;
  a->lg_b2 = opt->beta2 > 0.f ? log2f(opt->beta2) : 0.f;
  return XDFM_OK;
}

// one aligned 16-byte group [lin w | lin m | lin v | row_step] per row (the record layout of the engine)
bool tail_is_packed(const AgeDev& a, const float* lin_w, const float* lin_m, const float* lin_v, int64_t LS) {
  return a.row_step && lin_w && a.kind != XDFM_OPT_SGD && xdfm::opt_needs_v(a.kind) && lin_m == lin_w + 1 &&
         lin_v == lin_w + 2 && reinterpret_cast<const float*>(a.row_step) == lin_w + 3 && a.SS == LS &&
         (LS % 4) == 0 && (reinterpret_cast<uintptr_t>(lin_w) % 16) == 0;
}

}  // namespace

extern "C" int xdfm_build_row_ptr(const int64_t* dnn_feat_indices, int64_t nnz, int64_t n_rows,
                                  int32_t* row_ptr, int32_t* tok_row, int32_t* err_flag,
                                  void* stream) {
  XDFM_REQUIRE(row_ptr && err_flag, "xdfm_build_row_ptr: null output");
  XDFM_REQUIRE(nnz >= 0 && n_rows >= 0 && nnz < ((int64_t)1 << 31) && n_rows < ((int64_t)1 << 31) - 1,
               "xdfm_build_row_ptr: sizes out of range (nnz=%lld rows=%lld)", (long long)nnz,
               (long long)n_rows);
  XDFM_REQUIRE(nnz == 0 || (dnn_feat_indices && tok_row), "xdfm_build_row_ptr: null input");
  const int threads = 256;
  const unsigned blocks = (unsigned)((nnz + 1 + threads - 1) / threads);
  XDFM_LAUNCH("build_row_ptr_kernel", stream, build_row_ptr_kernel, dim3(blocks), dim3(threads), 0,
                     dnn_feat_indices, nnz, n_rows, row_ptr, tok_row, err_flag);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// row strides in floats; 0 = the dense layouts table[N][D], lin_w[N]
static bool strides_ok(int64_t* ts, int64_t* ls, int D, int vec) {
  if (*ts == 0) *ts = D;
  if (*ls == 0) *ls = 1;
  return *ts >= D && *ls >= 1 && (*ts % vec) == 0;
}

extern "C" int xdfm_embed_fwd(const float* table, const float* lin_w, int64_t table_stride,
                              int64_t lin_stride, int64_t N, int32_t D,
                              int32_t F, int64_t B, const int32_t* row_ptr, const int64_t* ids,
                              const float* weights, int64_t nnz, float* emb_out, float* lin_out,
                              float* fm_out, float* fm_sum, int32_t* err_flag, const xdfm_row_age* age,
                              void* stream) {
  XDFM_REQUIRE(table && row_ptr && emb_out && err_flag, "xdfm_embed_fwd: null argument");
  XDFM_REQUIRE(nnz == 0 || (ids && weights), "xdfm_embed_fwd: null token arrays");
  XDFM_REQUIRE(N > 0 && D > 0 && F > 0 && B >= 0, "xdfm_embed_fwd: bad sizes");
  XDFM_REQUIRE(!lin_out || lin_w, "xdfm_embed_fwd: lin_out needs lin_w");
  const int vec = pick_vec(D);
  XDFM_REQUIRE(D / vec <= kWave, "xdfm_embed_fwd: dim %d not supported (needs D/%d <= 64)", D, vec);
  XDFM_REQUIRE(strides_ok(&table_stride, &lin_stride, D, vec), "xdfm_embed_fwd: bad row strides");
  if (B == 0) return XDFM_OK;
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.table = table;
  a.lin_w = lin_out ? lin_w : nullptr;
  a.TS = table_stride;
  a.LS = lin_stride;
  a.N = N;
  a.B = B;
  a.nnz = nnz;
  a.D = D;
  a.F = F;
  a.row_ptr = row_ptr;
  a.ids = ids;
  a.weights = weights;
  a.emb_out = emb_out;
  a.lin_out = lin_out;
  a.fm_out = fm_out;
  a.fm_sum = fm_sum;
  a.err_flag = err_flag;
  int rc = make_age(age, a.lin_w, lin_stride, true, &a.age);
  if (rc) return rc;
  a.age.packed_tail = tail_is_packed(a.age, a.lin_w, a.age.lin_m, a.age.lin_v, lin_stride) ? 1 : 0;
  const unsigned blocks = (unsigned)((B + 3) / 4);
#define LAUNCH(V, A) \
  XDFM_LAUNCH("embed_fwd_kernel<" #V ", " #A ">", stream, (embed_fwd_kernel<V, A>), dim3(blocks), dim3(256), 0, a)
  if (a.age.row_step) {
    if (vec == 4) LAUNCH(4, true); else if (vec == 2) LAUNCH(2, true); else LAUNCH(1, true);
  } else {
    if (vec == 4) LAUNCH(4, false); else if (vec == 2) LAUNCH(2, false); else LAUNCH(1, false);
  }
#undef LAUNCH
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// plain row gather (owner side of the row-sharded table): D/VEC lanes per row
struct GatherArgs {
  const float *table, *lin_w;
  int64_t TS, LS, N, n, OS, OLS;
  int D;
  const int64_t* rows;
  float *out, *lin_out;
  int32_t* err_flag;
  AgeDev age;
};

template <int VEC, bool AGE>
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a) {
  xdfm::Stamp stamp_(xdfm::kStampGatherRows);
  __shared__ float4 cur_s[AGE ? xdfm::kDriftG : 1];
  int done = 0;
  if (AGE) {
    done = (int)a.age.state[3] + a.age.step_off;
    if (a.age.tab) cur_s[threadIdx.x] = a.age.tab[(done & 1) * xdfm::kDriftG + threadIdx.x];
    __syncthreads();
  }
  const int LPR = a.D / VEC;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = t / LPR;
  const int c = (int)(t - r * LPR);
  if (r >= a.n) return;
  int64_t id = a.rows[r];
  if ((uint64_t)id >= (uint64_t)a.N) {
    atomicOr(a.err_flag, XDFM_DEVERR_ID_RANGE);
    id = 0;
  }
  const bool want_lin = c == 0 && a.lin_out != nullptr;
  RowReq<VEC> q;
  row_request<VEC, AGE>(a.table, a.lin_w, a.TS, a.LS, a.age, id, c, want_lin, q);
  row_resolve<VEC, AGE>(a.age, cur_s, done, want_lin, q);
  store_vec<VEC>(a.out + r * a.OS + c * VEC, q.r.p);
  if (want_lin) a.lin_out[r * a.OLS] = q.l.p[0];
}

extern "C" int xdfm_embed_gather(const float* table, const float* lin_w, int64_t table_stride,
                                 int64_t lin_stride, int64_t N, int32_t D,
                                 const int64_t* rows, int64_t n, float* out, float* lin_out,
                                 int64_t out_stride, int32_t* err_flag, const xdfm_row_age* age, void* stream) {
  XDFM_REQUIRE(table && out && err_flag, "xdfm_embed_gather: null argument");
  XDFM_REQUIRE(N > 0 && D > 0 && n >= 0, "xdfm_embed_gather: bad sizes");
  XDFM_REQUIRE(!lin_out || lin_w, "xdfm_embed_gather: lin_out needs lin_w");
  if (n == 0) return XDFM_OK;
  XDFM_REQUIRE(rows, "xdfm_embed_gather: null rows");
  const int vec = pick_vec(D);
  XDFM_REQUIRE(D / vec <= kWave, "xdfm_embed_gather: dim %d not supported", D);
  XDFM_REQUIRE(strides_ok(&table_stride, &lin_stride, D, vec), "xdfm_embed_gather: bad row strides");
  // out_stride: floats between consecutive output rows AND between consecutive lin_out entries
  // (one buffer of [row | linear weight | pad] records goes through one all-to-all); 0 = dense
  XDFM_REQUIRE(out_stride == 0 || (out_stride >= D && out_stride % vec == 0), "xdfm_embed_gather: bad out_stride");
  GatherArgs a;
  memset(&a, 0, sizeof(a));
  a.table = table;
  a.lin_w = lin_out ? lin_w : nullptr;
  a.TS = table_stride;
  a.LS = lin_stride;
  a.N = N;
  a.n = n;
  a.OS = out_stride ? out_stride : D;
  a.OLS = out_stride ? out_stride : 1;
  a.D = D;
  a.rows = rows;
  a.out = out;
  a.lin_out = lin_out;
  a.err_flag = err_flag;
  int rc = make_age(age, a.lin_w, lin_stride, true, &a.age);
  if (rc) return rc;
  a.age.packed_tail = tail_is_packed(a.age, a.lin_w, a.age.lin_m, a.age.lin_v, lin_stride) ? 1 : 0;
  const int64_t threads = n * (D / vec);
  const unsigned blocks = (unsigned)((threads + 255) / 256);
#define LAUNCH(V, A) \
  XDFM_LAUNCH("gather_rows_kernel<" #V ", " #A ">", stream, (gather_rows_kernel<V, A>), dim3(blocks), dim3(256), 0, a)
  if (a.age.row_step) {
    if (vec == 4) LAUNCH(4, true); else if (vec == 2) LAUNCH(2, true); else LAUNCH(1, true);
  } else {
    if (vec == 4) LAUNCH(4, false); else if (vec == 2) LAUNCH(2, false); else LAUNCH(1, false);
  }
#undef LAUNCH
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_embed_bwd(int32_t mode, float* table, float* slot_m, float* slot_v,
                              float* lin_w, float* lin_m, float* lin_v, int64_t table_stride,
                              int64_t lin_stride, float* grad_table,
                              float* grad_lin, int64_t N, int32_t D, int32_t F, int64_t B,
                              const int32_t* sorted_ids, const int32_t* sorted_tok,
                              const int32_t* tok_row, const float* weights, int64_t nnz,
                              const float* d_emb, const float* g_logit, const float* fm_sum,
                              int32_t use_fm, const xdfm_opt_args* opt, float* pieces,
                              const float* d_emb2, int64_t grad_stride, const xdfm_row_age* age,
                              void* stream) {
  XDFM_REQUIRE(mode >= 0 && mode <= 2, "xdfm_embed_bwd: mode must be 0, 1 or 2");
  XDFM_REQUIRE(N > 0 && D > 0 && F > 0 && B >= 0 && nnz >= 0, "xdfm_embed_bwd: bad sizes");
  if (nnz == 0) return XDFM_OK;
  XDFM_REQUIRE(sorted_ids && tok_row && weights && d_emb && pieces,
               "xdfm_embed_bwd: null argument");
  XDFM_REQUIRE(!use_fm || (g_logit && fm_sum && table), "xdfm_embed_bwd: FM needs g_logit, fm_sum, table");
  XDFM_REQUIRE(!age || mode == 0, "xdfm_embed_bwd: rows are aged where they are updated (mode 0)");
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  if (mode == 0) {
    XDFM_REQUIRE(table, "xdfm_embed_bwd: sparse_rows mode needs the table");
    int rc = make_opt(opt, &a.opt);
    if (rc) return rc;
    XDFM_REQUIRE(opt->l1 == 0.f && opt->l2 == 0.f && opt->clip_norm == 0.f,
                 "xdfm_embed_bwd: sparse_rows mode cannot apply l1/l2/clip (their gradient is dense); "
                 "use mode 1 + xdfm_dense_update");
    XDFM_REQUIRE(opt->kind == XDFM_OPT_SGD || slot_m, "xdfm_embed_bwd: optimizer slot missing");
    XDFM_REQUIRE(!xdfm::opt_needs_v(opt->kind) || slot_v, "xdfm_embed_bwd: this optimizer needs slot_v");
    if (lin_w) {
      XDFM_REQUIRE(g_logit, "xdfm_embed_bwd: linear part needs g_logit");
      XDFM_REQUIRE(opt->kind == XDFM_OPT_SGD || lin_m, "xdfm_embed_bwd: lin slot missing");
      XDFM_REQUIRE(!xdfm::opt_needs_v(opt->kind) || lin_v, "xdfm_embed_bwd: lin slot_v missing");
    }
    XDFM_REQUIRE(!age || (age->opt && age->opt->kind == opt->kind), "xdfm_embed_bwd: age->opt must name the same optimizer");
  } else {
    XDFM_REQUIRE(grad_table, "xdfm_embed_bwd: scatter mode needs grad_table");
    XDFM_REQUIRE(!grad_lin || g_logit, "xdfm_embed_bwd: linear part needs g_logit");
  }
  a.table = table;
  a.slot_m = slot_m;
  a.slot_v = slot_v;
  a.lin_w = (mode == 0) ? lin_w : grad_lin;  // "linear part on" marker in both modes
  a.lin_m = lin_m;
  a.lin_v = lin_v;
  a.grad_table = grad_table;
  a.grad_lin = grad_lin;
  a.N = N;
  a.B = B;
  a.nnz = nnz;
  a.D = D;
  a.F = F;
  a.sorted_ids = sorted_ids;
  a.sorted_tok = sorted_tok;
  a.tok_row = tok_row;
  a.weights = weights;
  a.d_emb = d_emb;
  a.d_emb2 = d_emb2;
  a.g_logit = g_logit;
  a.fm_sum = fm_sum;
  a.use_fm = use_fm;
  a.pieces = pieces;

  const int vec = pick_vec(D);
  const int lpr = D / vec;
  XDFM_REQUIRE(lpr <= 64, "xdfm_embed_bwd: dim %d not supported (needs D/%d <= 64)", D, vec);
  XDFM_REQUIRE(strides_ok(&table_stride, &lin_stride, D, vec), "xdfm_embed_bwd: bad row strides");
  a.TS = table_stride;
  a.LS = lin_stride;
  XDFM_REQUIRE(grad_stride == 0 || (grad_stride >= D && grad_stride % vec == 0), "xdfm_embed_bwd: bad grad_stride");
  // mode 0 reads records; mode 1 reads dense rows and writes records; mode 2 reads records and writes dense
  a.RS = (mode != 1 && grad_stride) ? grad_stride : D;
  a.RLS = (mode != 1 && grad_stride) ? grad_stride : 1;
  a.GS = (mode == 1 && grad_stride) ? grad_stride : D;
  a.GLS = (mode == 1 && grad_stride) ? grad_stride : 1;
  int rc = make_age(mode == 0 ? age : nullptr, lin_w, lin_stride, false, &a.age);
  if (rc) return rc;
  a.age.packed_tail = (mode == 0 && tail_is_packed(a.age, lin_w, lin_m, lin_v, lin_stride)) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nchunks = (nnz + kChunk - 1) / kChunk;
  const unsigned blocks = (unsigned)((nchunks + 3) / 4);
  const unsigned blocks_long = (unsigned)((nchunks + 3) / 4);  // boundaries 1 .. nchunks-1
#define LAUNCH(V, M, A)                                                                                          \
  do {                                                                                                           \
    XDFM_LAUNCH("embed_bwd_kernel<" #V ", " #M ", " #A ">", st, (embed_bwd_kernel<V, M, A>), dim3(blocks),      \
                dim3(256), 0, a);                                                                                \
    XDFM_LAUNCH("embed_bwd_long_kernel<" #V ", " #M ", " #A ">", st, (embed_bwd_long_kernel<V, M, A>),          \
                dim3(blocks_long), dim3(256), 0, a);                                                             \
  } while (0)
  if (mode == 0 && a.age.row_step) {
    if (vec == 4) LAUNCH(4, 0, true); else if (vec == 2) LAUNCH(2, 0, true); else LAUNCH(1, 0, true);
  } else if (mode == 0) {
    if (vec == 4) LAUNCH(4, 0, false); else if (vec == 2) LAUNCH(2, 0, false); else LAUNCH(1, 0, false);
  } else {
    if (vec == 4) LAUNCH(4, 1, false); else if (vec == 2) LAUNCH(2, 1, false); else LAUNCH(1, 1, false);
  }
#undef LAUNCH
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// ---------------------------------------------------------------- aged rows written back
// One lane group (LPR = D/VEC lanes) per sorted position, groups never straddle a wave; only the
// first position of an id's run works.
struct CatchArgs {
  float *table, *slot_m, *slot_v, *lin_w, *lin_m, *lin_v;
  int64_t TS, LS, N, n;
  int D;
  const int32_t* sorted_ids;  // null: every row 0..N-1
  AgeDev age;
};

template <int VEC>
__global__ __launch_bounds__(256) void embed_catchup_kernel(CatchArgs a) {
  __shared__ float4 cur_s[xdfm::kDriftG];
  const int done = (int)a.age.state[3] + a.age.step_off;
  if (a.age.tab) cur_s[threadIdx.x] = a.age.tab[(done & 1) * xdfm::kDriftG + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int LPR = a.D / VEC;
  const int R = 64 / LPR;
  const int g = lane / LPR;
  const int c = lane - g * LPR;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t q = wave * R + g;
  if (g >= R || q >= a.n) return;
  int64_t id = q;
  if (a.sorted_ids) {
    id = a.sorted_ids[q];
    if (q > 0 && a.sorted_ids[q - 1] == id) return;
  }
  const int last = a.age.row_step[id * a.age.SS];
  const int G = done - last;
  if (age_needed(a.age.kind, last, G)) {
    const float4 cf = drift_coeffs(a.age, cur_s, last, G);
    const int64_t off = id * a.TS + c * VEC;
    RowVals<VEC> rv;
    row_load<VEC>(a.age.kind, a.table + off, a.slot_m + off, a.slot_v + off, rv);
    age_vals<VEC>(a.age, cf, rv, G);
    row_store<VEC>(a.age.kind, a.table + off, a.slot_m + off, a.slot_v + off, rv);
    if (a.lin_w && c == 0) {
      const int64_t lo = id * a.LS;
      RowVals<1> rl;
      row_load<1>(a.age.kind, a.lin_w + lo, a.lin_m + lo, a.lin_v + lo, rl);
      age_vals<1>(a.age, cf, rl, G);
      row_store<1>(a.age.kind, a.lin_w + lo, a.lin_m + lo, a.lin_v + lo, rl);
    }
  }
  // written last, by a lane whose own read of it has been consumed above
  if (c == 0 && G > 0) a.age.row_step[id * a.age.SS] = done;
}

extern "C" int xdfm_embed_clear_rows(float* grad_table, float* grad_lin, int32_t D,
                                     const int32_t* sorted_ids, int64_t nnz, void* stream) {
  XDFM_REQUIRE(D > 0 && nnz >= 0, "xdfm_embed_clear_rows: bad sizes");
  if (nnz == 0) return XDFM_OK;
  XDFM_REQUIRE(sorted_ids, "xdfm_embed_clear_rows: null sorted_ids");
  const int64_t total = nnz * D;
  XDFM_LAUNCH("clear_rows_kernel", stream, clear_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, grad_table, grad_lin, D, sorted_ids, nnz);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_embed_catchup(float* table, float* slot_m, float* slot_v, float* lin_w, float* lin_m,
                                  float* lin_v, int64_t table_stride, int64_t lin_stride, int64_t N, int32_t D,
                                  const int32_t* sorted_ids, int64_t n, const xdfm_row_age* age, void* stream) {
  XDFM_REQUIRE(table && age, "xdfm_embed_catchup: null argument");
  XDFM_REQUIRE(N > 0 && D > 0 && n >= 0, "xdfm_embed_catchup: bad sizes");
  CatchArgs a;
  memset(&a, 0, sizeof(a));
  int rc = make_age(age, lin_w, lin_stride, false, &a.age);
  if (rc) return rc;
  const int kind = a.age.kind;
  XDFM_REQUIRE(kind == XDFM_OPT_SGD || slot_m, "xdfm_embed_catchup: optimizer slot missing");
  XDFM_REQUIRE(!xdfm::opt_needs_v(kind) || slot_v, "xdfm_embed_catchup: this optimizer needs slot_v");
  XDFM_REQUIRE(!lin_w || ((kind == XDFM_OPT_SGD || lin_m) && (!xdfm::opt_needs_v(kind) || lin_v)),
               "xdfm_embed_catchup: lin slot missing");
  if (!sorted_ids) n = N;
  if (n == 0) return XDFM_OK;
  const int vec = pick_vec(D);
  XDFM_REQUIRE(D / vec <= kWave, "xdfm_embed_catchup: dim %d not supported", D);
  XDFM_REQUIRE(strides_ok(&table_stride, &lin_stride, D, vec), "xdfm_embed_catchup: bad row strides");
  a.table = table;
  a.slot_m = slot_m;
  a.slot_v = slot_v;
  a.lin_w = lin_w;
  a.lin_m = lin_m;
  a.lin_v = lin_v;
  a.TS = table_stride;
  a.LS = lin_stride;
  a.N = N;
  a.n = n;
  a.D = D;
  a.sorted_ids = sorted_ids;
  const int R = kWave / (D / vec);
  const int64_t waves = (n + R - 1) / R;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
#define LAUNCH(V) XDFM_LAUNCH("embed_catchup_kernel<" #V ">", stream, embed_catchup_kernel<V>, dim3(blocks), dim3(256), 0, a)
  if (vec == 4)
    LAUNCH(4);
  else if (vec == 2)
    LAUNCH(2);
  else
    LAUNCH(1);
#undef LAUNCH
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_embed_plan_gather(const int32_t* sorted_tok, const int32_t* tok_row, const float* weights,
                                      int64_t nnz, int32_t* row_sorted, float* w_sorted, void* stream) {
  XDFM_REQUIRE(nnz >= 0, "xdfm_embed_plan_gather: bad size");
  if (nnz == 0) return XDFM_OK;
  XDFM_REQUIRE(sorted_tok && tok_row && weights && row_sorted && w_sorted, "xdfm_embed_plan_gather: null argument");
  XDFM_LAUNCH("plan_gather_kernel", stream, plan_gather_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0,
              sorted_tok, tok_row, weights, nnz, row_sorted, w_sorted);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

XDFM_STAMP_UNIT(stamp_unit_embed)
