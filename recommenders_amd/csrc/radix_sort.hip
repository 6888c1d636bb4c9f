// Stable LSD radix sort of (feature id, token index) pairs for gfx950: the ordering step of the
// deterministic sorted-key segment sum (xdfm_embed_plan).  TensorFlow has no counterpart (its
// gradient of embedding_lookup_sparse scatters unordered); the sort exists so that every row's
// gradient is summed in token order, bit-identical from run to run.
//
// Sized to the key width: ceil(log2 N) bits are split evenly into ceil(bits / 9) digits (2^25 rows:
// 9 + 8 + 8), two launches per digit and nothing else:
//   sort_hist_kernel    per-tile digit counts -> table[tile][512]
//   sort_scatter_kernel each block sums the table's columns for its own global offsets (tiles before
//                       it, all tiles), ranks its keys with wave ballots (no atomics, stable), orders
//                       them by digit in LDS and writes runs of consecutive addresses.
// The first pass reads the int64 ids directly (clamped into [0, N)) and numbers the tokens; the last
// pass writes sorted_ids / sorted_tok.  HBM-bound in principle (8 bytes per pair per pass, read and
// written), in practice launch/latency-bound at one batch (319 488 pairs = 2.5 MB).
#include <hip/hip_runtime.h>

#include "sort_plan.h"
#include "xdfm_common.h"
#include "stamp.h"

namespace {

constexpr int kThreads = 512;          // 8 waves
constexpr int kWaves = kThreads / 64;
constexpr int kPerThread = 4;
constexpr int kSub = kThreads * kPerThread;  // keys ranked per block iteration
constexpr int kBins = xdfm::kSortBins;        // digits of at most 9 bits
constexpr int kMaxTiles = xdfm::kSortMaxTiles;  // table rows: a scatter block reads one column entry per tile

template <bool FIRST>
__device__ __forceinline__ uint32_t load_key(const void* src, int64_t i, int64_t N) {
  if (FIRST) {
    int64_t id = static_cast<const int64_t*>(src)[i];
    id = id < 0 ? 0 : (id >= N ? N - 1 : id);  // out-of-range ids are flagged by the forward pass
    return (uint32_t)id;
  }
  return static_cast<const uint32_t*>(src)[i];
}

// exclusive prefix sum of one value per thread over the 512 threads of a block (two barriers)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) before += (w < wave) ? wave_tot[w] : 0u;
  __syncthreads();
  return incl - v + before;
}

template <bool FIRST>
__global__ __launch_bounds__(kThreads) void sort_hist_kernel(const void* __restrict__ keys, int64_t n, int64_t N,
                                                             int shift, int bits, int64_t tile,
                                                             uint32_t* __restrict__ table) {
  xdfm::Stamp stamp_(xdfm::kStampSort);
  __shared__ uint32_t h[kBins];
  const int tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const uint32_t dmask = (1u << bits) - 1u;
  const int64_t beg = (int64_t)blockIdx.x * tile;
  const int64_t end = beg + tile < n ? beg + tile : n;
  for (int64_t i0 = beg; i0 < end; i0 += 4 * kThreads) {
    uint32_t d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // four loads in flight per lane
      const int64_t i = i0 + u * kThreads + tid;
      d[u] = i < end ? ((load_key<FIRST>(keys, i, N) >> shift) & dmask) : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // after the first pass equal keys sit side by side (the 13 numeric fields of Criteo-shaped
      // data are 13 ids x B tokens): a wave whose lanes all hold one digit adds once
      const uint32_t d0 = __builtin_amdgcn_readfirstlane(d[u]);
      if (__all(d[u] == d0)) {
        if ((tid & 63) == 0 && d0 != 0xffffffffu) atomicAdd(&h[d0], 64u);
      } else if (d[u] != 0xffffffffu) {
        atomicAdd(&h[d[u]], 1u);
      }
    }
  }
  __syncthreads();
  table[(int64_t)blockIdx.x * kBins + tid] = h[tid];
}

template <bool FIRST>
__global__ __launch_bounds__(kThreads) void sort_scatter_kernel(
    const void* __restrict__ src_keys, const uint32_t* __restrict__ src_vals, uint32_t* __restrict__ dst_keys,
    uint32_t* __restrict__ dst_vals, int64_t n, int64_t N, int shift, int bits, int64_t tile, int nt,
    const uint32_t* __restrict__ table) {
  xdfm::Stamp stamp_(xdfm::kStampSort);
  __shared__ uint32_t whist[kWaves][kBins];  // per wave: keys of each digit seen so far
  __shared__ uint32_t gbase[kBins];          // where this block's next key of each digit goes
  __shared__ uint32_t tstart[kBins];         // start of each digit inside the ordered sub-tile
  __shared__ uint32_t skey[kSub];
  __shared__ uint32_t sval[kSub];
  __shared__ uint32_t wave_tot[kWaves];
  __shared__ uint32_t colsum[2][4][kBins];   // [tiles before | all tiles][row group][digit]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t dmask = (1u << bits) - 1u;
  const int t = blockIdx.x;
  const int64_t tbeg = (int64_t)t * tile;
  const int64_t tend = tbeg + tile < n ? tbeg + tile : n;
  uint32_t key[kPerThread], val[kPerThread], rk[kPerThread];
  // wave w owns positions [256 w, 256 w + 256) of a sub-tile, 64 at a time, in order
  auto load_sub = [&](int64_t sb) {
    const int sub_n = (int)(tend - sb < kSub ? tend - sb : kSub);
#pragma unroll
    for (int r = 0; r < kPerThread; ++r) {
      const int li = wave * (64 * kPerThread) + r * 64 + lane;
      const bool act = li < sub_n;
      key[r] = act ? load_key<FIRST>(src_keys, sb + li, N) : 0u;
      val[r] = FIRST ? (uint32_t)(sb + li) : (act ? src_vals[sb + li] : 0u);
    }
  };
  load_sub(tbeg);  // requested ahead of the table's columns: the two latencies overlap
  {
    // keys of each digit in the tiles before this one, and in all tiles: the table's column sums.
    // 128 lanes cover a row of 512 counts with 16-byte loads, 4 row groups, 13 loads in flight.
    const int q = tid & 127, rg = tid >> 7;
    uint32_t before[4] = {0, 0, 0, 0}, total[4] = {0, 0, 0, 0};
    constexpr int U = 13;
    const uint4* rows = reinterpret_cast<const uint4*>(table);
    for (int r0 = rg; r0 < nt; r0 += 4 * U) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = r0 + 4 * u;
        v[u] = row < nt ? rows[(int64_t)row * (kBins / 4) + q] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool bf = r0 + 4 * u < t;
        before[0] += bf ? v[u].x : 0u;
        before[1] += bf ? v[u].y : 0u;
        before[2] += bf ? v[u].z : 0u;
        before[3] += bf ? v[u].w : 0u;
        total[0] += v[u].x;
        total[1] += v[u].y;
        total[2] += v[u].z;
        total[3] += v[u].w;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      colsum[0][rg][4 * q + k] = before[k];
      colsum[1][rg][4 * q + k] = total[k];
    }
    __syncthreads();
    uint32_t bsum = 0, tsum = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bsum += colsum[0][g][tid];
      tsum += colsum[1][g][tid];
    }
    gbase[tid] = block_excl_scan(tsum, wave_tot) + bsum;
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int64_t sb = tbeg; sb < tend; sb += kSub) {
    const int sub_n = (int)(tend - sb < kSub ? tend - sb : kSub);
    if (sb != tbeg) load_sub(sb);
#pragma unroll
    for (int w = 0; w < kWaves; ++w) whist[w][tid] = 0;
    __syncthreads();
    volatile uint32_t* wh = whist[wave];
#pragma unroll
    for (int r = 0; r < kPerThread; ++r) {
      const int li = wave * (64 * kPerThread) + r * 64 + lane;
      const bool act = li < sub_n;
      const uint32_t d = (key[r] >> shift) & dmask;
      unsigned long long peers = __ballot(act);  // lanes holding the same digit
      for (int b = 0; b < bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit && act);
        peers &= bit ? bal : ~bal;
      }
      const uint32_t rank = (uint32_t)__popcll(peers & lt);
      const uint32_t pre = act ? wh[d] : 0u;
      rk[r] = pre + rank;
      __builtin_amdgcn_wave_barrier();
      if (act && rank == 0) wh[d] = pre + (uint32_t)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // thread = digit: counts of the waves -> exclusive over waves; start of the digit in the sub-tile
    uint32_t cnt = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t x = whist[w][tid];
      whist[w][tid] = cnt;
      cnt += x;
    }
    tstart[tid] = block_excl_scan(cnt, wave_tot);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kPerThread; ++r) {
      const int li = wave * (64 * kPerThread) + r * 64 + lane;
      if (li < sub_n) {
        const uint32_t d = (key[r] >> shift) & dmask;
        const uint32_t lp = tstart[d] + whist[wave][d] + rk[r];
        skey[lp] = key[r];
        sval[lp] = val[r];
      }
    }
    __syncthreads();
    // ordered by digit now: neighbouring lanes write neighbouring addresses inside a digit's run
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      const int i = tid + k * kThreads;
      if (i < sub_n) {
        const uint32_t kk = skey[i];
        const uint32_t d = (kk >> shift) & dmask;
        const int64_t g = (int64_t)gbase[d] + (uint32_t)(i - (int)tstart[d]);
        dst_keys[g] = kk;
        dst_vals[g] = sval[i];
      }
    }
    __syncthreads();
    gbase[tid] += cnt;  // read again only after the next iteration's barriers
  }
}

inline int key_bits(int64_t N) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < N) ++bits;
  return bits;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int64_t tile_of(int64_t n) {
  const int64_t subs = (n + kSub - 1) / kSub;
  const int64_t per_tile = (subs + kMaxTiles - 1) / kMaxTiles;
  return (per_tile < 1 ? 1 : per_tile) * kSub;
}

}  // namespace

namespace xdfm {

int plan_sort(const int64_t* ids, int64_t n, int64_t N, int32_t* sorted_ids, int32_t* sorted_tok, void* ws,
              size_t ws_bytes, hipStream_t st) {
  XDFM_REQUIRE(n > 0 && N > 0 && N <= ((int64_t)1 << 31) && n < ((int64_t)1 << 31), "plan_sort: bad sizes");
  XDFM_REQUIRE(ids && sorted_ids && sorted_tok && ws, "plan_sort: null argument");
  XDFM_REQUIRE(((uintptr_t)ws & 255) == 0, "plan_sort: workspace must be 256-byte aligned");
  if (ws_bytes < plan_sort_workspace_bytes(n))
    return fail(XDFM_E_WORKSPACE, "xdfm_embed_plan: workspace %zu < %zu", ws_bytes, plan_sort_workspace_bytes(n));
  const size_t seg = align_up((size_t)n * sizeof(uint32_t), 256);
  char* base = static_cast<char*>(ws);
  uint32_t* kbuf[2] = {(uint32_t*)base, (uint32_t*)(base + 2 * seg)};
  uint32_t* vbuf[2] = {(uint32_t*)(base + seg), (uint32_t*)(base + 3 * seg)};
  uint32_t* table = (uint32_t*)(base + 4 * seg);
  const int bits = key_bits(N);
  const int npass = (bits + 8) / 9;
  const int64_t tile = tile_of(n);
  const int nt = (int)((n + tile - 1) / tile);
  const void* src_k = ids;
  const uint32_t* src_v = nullptr;
  int shift = 0;
  for (int p = 0; p < npass; ++p) {
    const int pbits = bits / npass + (p < bits % npass ? 1 : 0);
    const bool last = p == npass - 1;
    uint32_t* dst_k = last ? (uint32_t*)sorted_ids : kbuf[p & 1];
    uint32_t* dst_v = last ? (uint32_t*)sorted_tok : vbuf[p & 1];
    if (p == 0) {
      XDFM_LAUNCH("sort_hist_kernel", st, sort_hist_kernel<true>, dim3(nt), dim3(kThreads), 0, src_k, n, N, shift,
                  pbits, tile, table);
      XDFM_LAUNCH("sort_scatter_kernel", st, sort_scatter_kernel<true>, dim3(nt), dim3(kThreads), 0, src_k, src_v,
                  dst_k, dst_v, n, N, shift, pbits, tile, nt, table);
    } else {
      XDFM_LAUNCH("sort_hist_kernel", st, sort_hist_kernel<false>, dim3(nt), dim3(kThreads), 0, src_k, n, N, shift,
                  pbits, tile, table);
      XDFM_LAUNCH("sort_scatter_kernel", st, sort_scatter_kernel<false>, dim3(nt), dim3(kThreads), 0, src_k, src_v,
                  dst_k, dst_v, n, N, shift, pbits, tile, nt, table);
    }
    src_k = dst_k;
    src_v = dst_v;
    shift += pbits;
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

// Generic form over 32-bit keys with a 32-bit payload each (the evaluation's order-preserving score keys with
// the label as payload): `bits` low key bits are sorted on, in digits of at most 9 bits.  Same kernels, same
// workspace layout (plan_sort_workspace_bytes(n)); keys_in / vals_in are left untouched.
int sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, int64_t n, int bits, uint32_t* keys_out,
                   uint32_t* vals_out, void* ws, size_t ws_bytes, hipStream_t st) {
  XDFM_REQUIRE(n > 0 && n < ((int64_t)1 << 31) && bits >= 1 && bits <= 32, "sort_pairs_u32: bad sizes");
  XDFM_REQUIRE(keys_in && vals_in && keys_out && vals_out && ws, "sort_pairs_u32: null argument");
  XDFM_REQUIRE(((uintptr_t)ws & 255) == 0, "sort_pairs_u32: workspace must be 256-byte aligned");
  if (ws_bytes < plan_sort_workspace_bytes(n))
    return fail(XDFM_E_WORKSPACE, "sort_pairs_u32: workspace %zu < %zu", ws_bytes, plan_sort_workspace_bytes(n));
  const size_t seg = align_up((size_t)n * sizeof(uint32_t), 256);
  char* base = static_cast<char*>(ws);
  uint32_t* kbuf[2] = {(uint32_t*)base, (uint32_t*)(base + 2 * seg)};
  uint32_t* vbuf[2] = {(uint32_t*)(base + seg), (uint32_t*)(base + 3 * seg)};
  uint32_t* table = (uint32_t*)(base + 4 * seg);
  const int npass = (bits + 8) / 9;
  const int64_t tile = tile_of(n);
  const int nt = (int)((n + tile - 1) / tile);
  const uint32_t* src_k = keys_in;
  const uint32_t* src_v = vals_in;
  int shift = 0;
  for (int p = 0; p < npass; ++p) {
    const int pbits = bits / npass + (p < bits % npass ? 1 : 0);
    const bool last = p == npass - 1;
    uint32_t* dst_k = last ? keys_out : kbuf[p & 1];
    uint32_t* dst_v = last ? vals_out : vbuf[p & 1];
    XDFM_LAUNCH("sort_hist_kernel", st, sort_hist_kernel<false>, dim3(nt), dim3(kThreads), 0, (const void*)src_k, n,
                (int64_t)0, shift, pbits, tile, table);
    XDFM_LAUNCH("sort_scatter_kernel", st, sort_scatter_kernel<false>, dim3(nt), dim3(kThreads), 0, (const void*)src_k,
                src_v, dst_k, dst_v, n, (int64_t)0, shift, pbits, tile, nt, table);
    src_k = dst_k;
    src_v = dst_v;
    shift += pbits;
  }
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

}  // namespace xdfm

extern "C" int xdfm_embed_plan_workspace_bytes(int64_t nnz, int64_t N, size_t* bytes) {
  XDFM_REQUIRE(bytes && nnz >= 0 && N > 0, "xdfm_embed_plan_workspace_bytes: bad argument");
  *bytes = xdfm::plan_sort_workspace_bytes(nnz);
  return XDFM_OK;
}

extern "C" int xdfm_embed_plan(const int64_t* ids, int64_t nnz, int64_t N, int32_t* sorted_ids,
                               int32_t* sorted_tok, void* plan_ws, size_t plan_ws_bytes, void* stream) {
  XDFM_REQUIRE(nnz >= 0 && N > 0 && N <= ((int64_t)1 << 31), "xdfm_embed_plan: bad sizes");
  if (nnz == 0) return XDFM_OK;
  return xdfm::plan_sort(ids, nnz, N, sorted_ids, sorted_tok, plan_ws, plan_ws_bytes, (hipStream_t)stream);
}

XDFM_STAMP_UNIT(stamp_unit_sort)
