// Device helpers shared by the CIN kernel files: staging of X tiles into LDS.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace xdfm_cin {
namespace {

constexpr int kPad = 16;  // LDS row stride = rows + 16 floats: the four k-groups of a wave hit disjoint banks

// x / d for 0 <= x < 2^21 and 1 <= d, with inv = 1.0f / d: three VALU ops instead of the ~20
// of an integer division (the index arithmetic below shares the SIMD lanes with the MFMAs).
// (x + 0.5) / d is at least 0.5 / d away from an integer and the float error is below
// x / d * 2^-22, so the truncation is exact.
__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// stage src[b0 .. b0+nb)[R][D] transposed into lds[r * Tp + bl*D + d]; rows R..Rpad are zero.
// Four loads per thread are in flight before the first LDS write (the tile is read once, so
// this loop runs at memory latency unless it keeps several requests outstanding).
__device__ __forceinline__ void stage_cols(const float* __restrict__ src, int64_t b0, int nb, int R,
                                           int Rpad, int D, float* lds, int Tp) {
  const float* base = src + b0 * R * D;
  const int nt = blockDim.x;
  const float invR = 1.0f / (float)R;
  if ((D & 3) == 0) {
    const int D4 = D >> 2;
    const float invD4 = 1.0f / (float)D4;
    const int total4 = nb * R * D4;
    const float4* base4 = reinterpret_cast<const float4*>(base);
    for (int q0 = threadIdx.x; q0 < total4; q0 += 4 * nt) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * nt;
        v[u] = base4[q < total4 ? q : total4 - 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * nt;
        if (q < total4) {
          const int row = fdiv(q, invD4);
          const int d4 = q - row * D4;
          const int bl = fdiv(row, invR);
          const int r = row - bl * R;
          *reinterpret_cast<float4*>(lds + r * Tp + bl * D + 4 * d4) = v[u];
        }
      }
    }
  } else {
    const int total = nb * R * D;
    const float invD = 1.0f / (float)D;
    for (int q0 = threadIdx.x; q0 < total; q0 += 4 * nt) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * nt;
        v[u] = base[q < total ? q : total - 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u * nt;
        if (q < total) {
          const int row = fdiv(q, invD);
          const int d = q - row * D;
          const int bl = fdiv(row, invR);
          const int r = row - bl * R;
          lds[r * Tp + bl * D + d] = v[u];
        }
      }
    }
  }
  // padding rows (feature index >= R) and padding columns (row >= nb*D) must read as zero
  const int cols = nb * D;
  for (int idx = threadIdx.x; idx < (Rpad - R) * Tp; idx += nt) lds[R * Tp + idx] = 0.f;
  const int padc = Tp - cols;
  for (int idx = threadIdx.x; idx < R * padc; idx += nt) {
    const int r = idx / padc;
    lds[r * Tp + cols + (idx - r * padc)] = 0.f;
  }
}

}  // namespace
}  // namespace xdfm_cin
