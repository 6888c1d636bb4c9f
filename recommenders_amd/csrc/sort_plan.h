// Stable LSD radix sort of a batch's tokens by feature id (internal interface shared by
// xdfm_embed_plan and its no-allocation variant; csrc/radix_sort.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace xdfm {

constexpr int kSortBins = 512;      // digits of at most 9 bits
constexpr int kSortMaxTiles = 512;  // rows of the per-tile digit table

// bytes of scratch plan_sort needs for n tokens: two ping-pong (key, token) buffer pairs and the
// digit table
constexpr size_t plan_sort_workspace_bytes(int64_t n) {
  return 4 * (((size_t)(n > 0 ? n : 1) * sizeof(uint32_t) + 255) / 256 * 256) +
         (size_t)kSortMaxTiles * kSortBins * sizeof(uint32_t);
}

// sorted_ids[n] = the ids (clamped into [0, N)) in ascending order, sorted_tok[n] = the token index
// each came from; tokens with equal ids keep their order.  ws: plan_sort_workspace_bytes(n) bytes,
// 256-byte aligned; nothing in it has to be initialised.  Enqueues 2 launches per digit pass
// (ceil(log2 N) bits in digits of <= 9 bits).
int plan_sort(const int64_t* ids, int64_t n, int64_t N, int32_t* sorted_ids, int32_t* sorted_tok,
              void* ws, size_t ws_bytes, hipStream_t stream);

// The same sort over 32-bit keys with a 32-bit payload each (`bits` low key bits, stable); same workspace.
int sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, int64_t n, int bits, uint32_t* keys_out,
                   uint32_t* vals_out, void* ws, size_t ws_bytes, hipStream_t stream);

}  // namespace xdfm
