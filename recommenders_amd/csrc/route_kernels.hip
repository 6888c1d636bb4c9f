// Routing of a batch's tokens to the ranks that own their table rows (row-sharded table, owner =
// id mod G, local row = id div G).  No counterpart in the reference (one device,
// one table).  Index work only, bit-exact by construction:
//   xdfm_route_keys : key = owner * rows_per_rank + local_row  -- one stable sort of these keys
//                     (xdfm_embed_plan) then groups the tokens by owner, orders them by row inside an
//                     owner and puts duplicates side by side
//   xdfm_route_uniq : run heads of the sorted keys -> the per-source de-duplicated row list each
//                     owner is asked for (1/3 of the tokens on Criteo-shaped data), the token ->
//                     slot inverse map the local forward reads its rows through, the slot of every
//                     sorted position (segment key of the per-slot gradient pre-reduction) and the
//                     first slot of every owner (exchange split sizes).
#include <hip/hip_runtime.h>

#include "xdfm_common.h"
#include "stamp.h"

namespace {

__global__ void route_keys_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t N, int32_t G, int64_t rpr,
                                  int64_t* __restrict__ keys) {
  xdfm::Stamp stamp_(xdfm::kStampRoute);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t id = ids[i];
  id = id < 0 ? 0 : (id >= N ? N - 1 : id);  // out-of-range ids are flagged by the forward pass
  keys[i] = (id % G) * rpr + id / G;
}

constexpr int kUT = 256;
constexpr int kUPer = 4;
constexpr int kUTile = kUT * kUPer;

__device__ __forceinline__ bool is_head(const int32_t* __restrict__ sk, int64_t i) {
  return i == 0 || sk[i] != sk[i - 1];
}

__global__ __launch_bounds__(kUT) void route_count_kernel(const int32_t* __restrict__ sk, int64_t n,
                                                          int32_t* __restrict__ block_heads) {
  xdfm::Stamp stamp_(xdfm::kStampRoute);
  __shared__ int wsum[kUT / 64];
  const int64_t base = (int64_t)blockIdx.x * kUTile;
  int c = 0;
#pragma unroll
  for (int u = 0; u < kUPer; ++u) {
    const int64_t i = base + u * kUT + threadIdx.x;
    c += (i < n && is_head(sk, i)) ? 1 : 0;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kUT / 64; ++w) t += wsum[w];
    block_heads[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kUT) void route_write_kernel(const int32_t* __restrict__ sk, const int32_t* __restrict__ st,
                                                          int64_t n, int64_t rpr, int32_t G,
                                                          const int32_t* __restrict__ block_heads,
                                                          int32_t* __restrict__ slot_sorted, int64_t* __restrict__ inverse,
                                                          int64_t* __restrict__ uniq_rows, int32_t* __restrict__ first_slot) {
  xdfm::Stamp stamp_(xdfm::kStampRoute);
  __shared__ int red[kUT];
  __shared__ int wtot[kUT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // heads in the blocks before this one
  int before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += kUT) before += block_heads[b];
  red[tid] = before;
  __syncthreads();
  for (int s = kUT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const int block_base = red[0];
  __syncthreads();
  // thread owns kUPer consecutive positions
  const int64_t p0 = (int64_t)blockIdx.x * kUTile + (int64_t)tid * kUPer;
  int flag[kUPer], mine = 0;
#pragma unroll
  for (int u = 0; u < kUPer; ++u) {
    flag[u] = (p0 + u < n && is_head(sk, p0 + u)) ? 1 : 0;
    mine += flag[u];
  }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int wbefore = 0;
  for (int w = 0; w < wave; ++w) wbefore += wtot[w];
  int slot = block_base + wbefore + incl - mine - 1;  // slot of the last head at or before p0 - 1
#pragma unroll
  for (int u = 0; u < kUPer; ++u) {
    const int64_t p = p0 + u;
    if (p >= n) break;
    slot += flag[u];
    slot_sorted[p] = slot;
    inverse[st[p]] = slot;
    if (flag[u]) {
      const int64_t key = (uint32_t)sk[p];
      uniq_rows[slot] = key % rpr;
      const int owner = (int)(key / rpr);
      if (p == 0 || (int)((int64_t)(uint32_t)sk[p - 1] / rpr) != owner) first_slot[owner] = slot;
    }
    if (p == n - 1) first_slot[G] = slot + 1;  // the number of distinct rows
  }
}

// Fixed-capacity form of a request: owner g's distinct rows in slots [g S, g S + count_g) of its block of
// S = cap + 1 slots, pad_row (a scratch row the owner keeps past its real ones) in the unused ones, so that every
// exchange of the step moves G x S records whatever the batch holds and needs no sizes.  Slots are renumbered
// accordingly (slot' = owner S + position inside the owner).  The last slot of every block is a header: the low
// 32 bits name pad_row like an unused slot, the high 32 bits carry this rank's LARGEST per-owner count, so the
// message that takes the rows to their owners also tells every owner whether this source's batch fits its
// capacity (xdfm_route_decode).  counts[g] = count_g, counts[G] = their maximum: a batch whose maximum exceeds cap
// does not fit (nothing is written out of bounds for it, its padded arrays are not to be used).
__global__ __launch_bounds__(256) void route_pad_kernel(const int32_t* __restrict__ first_slot,
                                                        const int64_t* __restrict__ uniq_rows,
                                                        const int32_t* __restrict__ slot_sorted,
                                                        const int64_t* __restrict__ inverse, int64_t n, int32_t G,
                                                        int64_t cap, int64_t pad_row, int64_t* __restrict__ rows_padded,
                                                        int32_t* __restrict__ slot_sorted_p,
                                                        int64_t* __restrict__ inverse_p, int32_t* __restrict__ counts) {
  xdfm::Stamp stamp_(xdfm::kStampRoute);
  __shared__ int start[65];
  __shared__ int mx_s;
  if (threadIdx.x == 0) {
    // an owner without a row (-1) starts where the next one does; no token at all: everything 0
    int nxt = first_slot[G] < 0 ? 0 : first_slot[G];
    start[G] = nxt;
    for (int g = G - 1; g >= 0; --g) {
      const int f = first_slot[g];
      if (f >= 0 && first_slot[G] >= 0) nxt = f;
      start[g] = nxt;
    }
    int mx = 0;
    for (int g = 0; g < G; ++g) {
      const int c = start[g + 1] - start[g];
      mx = c > mx ? c : mx;
    }
    mx_s = mx;
  }
  __syncthreads();
  const int64_t S = cap + 1;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    for (int g = 0; g < G; ++g) counts[g] = start[g + 1] - start[g];
    counts[G] = mx_s;
  }
  if (t < (int64_t)G * S) {
    const int g = (int)(t / S);
    const int64_t k = t - (int64_t)g * S;
    int64_t v = pad_row;
    if (k == cap)
      v = pad_row | ((int64_t)mx_s << 32);
    else if (k < start[g + 1] - start[g])
      v = uniq_rows[start[g] + k];
    rows_padded[t] = v;
  }
  if (t < n) {
    auto renumber = [&](int slot) -> int64_t {
      int g = 0;
      while (g + 1 < G && slot >= start[g + 1]) ++g;
      const int64_t k = slot - start[g];
      return k < cap ? (int64_t)g * S + k : 0;  // (k >= cap: the batch does not fit, see above)
    };
    slot_sorted_p[t] = (int32_t)renumber(slot_sorted[t]);
    inverse_p[t] = renumber((int)inverse[t]);
  }
}

// owner side: the G blocks that arrived -> plain row indices (headers become the scratch row) and every
// source's largest per-owner count
__global__ void route_decode_kernel(const int64_t* __restrict__ raw, int32_t G, int64_t cap,
                                    int64_t* __restrict__ rows, int32_t* __restrict__ src_max) {
  xdfm::Stamp stamp_(xdfm::kStampRoute);
  const int64_t S = cap + 1;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)G * S) return;
  const int64_t v = raw[t];
  rows[t] = v & 0xffffffffll;
  if (t % S == cap) src_max[t / S] = (int32_t)(v >> 32);
}

}  // namespace

extern "C" int xdfm_route_pad(const int32_t* first_slot, const int64_t* uniq_rows, const int32_t* slot_sorted,
                              const int64_t* inverse, int64_t n, int32_t G, int64_t cap, int64_t pad_row,
                              int64_t* rows_padded, int32_t* slot_sorted_p, int64_t* inverse_p, int32_t* counts,
                              void* stream) {
  XDFM_REQUIRE(n >= 0 && G > 0 && G <= 64 && cap > 0 && (int64_t)G * (cap + 1) < ((int64_t)1 << 31),
               "xdfm_route_pad: bad sizes (at most 64 owners)");
  XDFM_REQUIRE(pad_row >= 0 && pad_row < ((int64_t)1 << 31), "xdfm_route_pad: pad_row out of range");
  XDFM_REQUIRE(first_slot && rows_padded && counts, "xdfm_route_pad: null argument");
  XDFM_REQUIRE(n == 0 || (uniq_rows && slot_sorted && inverse && slot_sorted_p && inverse_p),
               "xdfm_route_pad: null argument");
  const int64_t slots = (int64_t)G * (cap + 1);
  const int64_t threads = n > slots ? n : slots;
  XDFM_LAUNCH("route_pad_kernel", stream, route_pad_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
              first_slot, uniq_rows, slot_sorted, inverse, n, G, cap, pad_row, rows_padded, slot_sorted_p, inverse_p,
              counts);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_route_decode(const int64_t* raw, int32_t G, int64_t cap, int64_t* rows, int32_t* src_max,
                                 void* stream) {
  XDFM_REQUIRE(raw && rows && src_max && G > 0 && cap > 0, "xdfm_route_decode: bad argument");
  const int64_t slots = (int64_t)G * (cap + 1);
  XDFM_LAUNCH("route_decode_kernel", stream, route_decode_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, raw,
              G, cap, rows, src_max);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_route_keys(const int64_t* ids, int64_t n, int64_t N, int32_t G, int64_t* keys, void* stream) {
  XDFM_REQUIRE(n >= 0 && N > 0 && G > 0, "xdfm_route_keys: bad sizes");
  if (n == 0) return XDFM_OK;
  XDFM_REQUIRE(ids && keys, "xdfm_route_keys: null argument");
  const int64_t rpr = (N + G - 1) / G;
  XDFM_REQUIRE(rpr * G <= ((int64_t)1 << 31), "xdfm_route_keys: FEATURE_COUNT too large for 32-bit route keys");
  XDFM_LAUNCH("route_keys_kernel", stream, route_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ids, n, N, G,
              rpr, keys);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_route_uniq(const int32_t* sorted_keys, const int32_t* sorted_tok, int64_t n, int64_t N, int32_t G,
                               int32_t* slot_sorted, int64_t* inverse, int64_t* uniq_rows, int32_t* first_slot,
                               int32_t* block_scratch, void* stream) {
  XDFM_REQUIRE(n >= 0 && N > 0 && G > 0, "xdfm_route_uniq: bad sizes");
  XDFM_REQUIRE(first_slot, "xdfm_route_uniq: null first_slot");
  hipStream_t st = (hipStream_t)stream;
  // owners without a row keep -1; the caller fills them forward from first_slot[G]
  XDFM_HIP(hipMemsetAsync(first_slot, 0xFF, sizeof(int32_t) * (size_t)(G + 1), st));
  if (n == 0) return XDFM_OK;
  XDFM_REQUIRE(sorted_keys && sorted_tok && slot_sorted && inverse && uniq_rows && block_scratch,
               "xdfm_route_uniq: null argument");
  const int64_t rpr = (N + G - 1) / G;
  const unsigned blocks = (unsigned)((n + kUTile - 1) / kUTile);
  XDFM_LAUNCH("route_count_kernel", st, route_count_kernel, dim3(blocks), dim3(kUT), 0, sorted_keys, n, block_scratch);
  XDFM_LAUNCH("route_write_kernel", st, route_write_kernel, dim3(blocks), dim3(kUT), 0, sorted_keys, sorted_tok, n, rpr, G,
              block_scratch, slot_sorted, inverse, uniq_rows, first_slot);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}

extern "C" int xdfm_route_scratch_ints(int64_t n, size_t* ints) {
  XDFM_REQUIRE(ints && n >= 0, "xdfm_route_scratch_ints: bad argument");
  *ints = (size_t)((n + kUTile - 1) / kUTile) + 1;
  return XDFM_OK;
}

XDFM_STAMP_UNIT(stamp_unit_route)
