// Shared by the bf16 matrix-core CIN kernels (cin_bf16_kernels.hip, cin_bf16_dw_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "xdfm_common.h"

namespace xdfm_bf16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// row of a 32x32 result tile held in register i of a lane in half `half` (lane >> 5)
__device__ __forceinline__ int tile_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }

constexpr int kMT = 2;           // 32-column m tiles per wave
constexpr int kWaves = 4;        // waves per block
constexpr int kMB = kWaves * kMT * 32;  // m per block

struct Dims {
  int64_t B, cols;  // cols = B * D
  int F, H, D, L;   // L = the channels these kernels run (the layer's hidden ones)
  int masked;
  int Fp, G;        // f padded to a multiple of G; G filters' images form one LDS chunk
};


inline int make_dims(int64_t B, int F, int H, int D, int L, int masked, int nks, Dims* c) {
  XDFM_REQUIRE(B >= 0 && F > 0 && H > 0 && D > 0 && L > 0, "xdfm_cin_bf16: bad sizes");
  XDFM_REQUIRE(!masked || F == H, "xdfm_cin_bf16: the masked layer has xk == x0");
  c->B = B;
  c->cols = B * D;
  c->F = F;
  c->H = H;
  c->D = D;
  c->L = L;
  c->masked = masked ? 1 : 0;
  // f per staged chunk: 8 KB of filter image per barrier; 16 KB (two f) for the widest filters, where
  // a barrier per 16 products measured 6 % slower, while two blocks still fit a CU (x0 factors Fp KB +
  // two chunks, at most 80 KB per block).  Narrower filters measured slower with 16 KB.
  c->G = nks > 0 ? (8 / nks > 0 ? 8 / nks : 1) : 1;
  c->Fp = (F + c->G - 1) / c->G * c->G;
  if (nks == 8 && (size_t)((F + 1) / 2 * 2) * 1024 + (size_t)2 * 2 * nks * 1024 <= 80 * 1024) {
    c->G = 2;
    c->Fp = (F + 1) / 2 * 2;
  }
  return XDFM_OK;
}

}  // namespace xdfm_bf16
