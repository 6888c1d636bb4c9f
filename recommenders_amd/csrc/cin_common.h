// Shared between the legacy VALU CIN kernels (cin_kernels.hip) and the MFMA ones.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/xdfm.h"

namespace xdfm_cin {

struct CinDims {
  int64_t B;
  int F, H, D, L, n_hidden, masked;
  int ldw;  // floats between consecutive rows of the filter / its gradient (0 = L): a layer may run on
            // a leading column range of a wider filter (its hidden channels)
};
__host__ __device__ inline int cin_ldw(const CinDims& c) { return c.ldw ? c.ldw : c.L; }

// shapes the MFMA path handles (else the generic VALU kernels run)
bool mfma_supported(int64_t B, int F, int H, int D, int L);
size_t fwd_mfma_workspace_floats(int64_t B, int F, int H, int D, int L);
int fwd_mfma(const CinDims& c, const float* x0, const float* xk, const float* w, float* xnext,
             float* pooled, int pool_stride, int pool_off, float* ws, hipStream_t st);
size_t bwd_mfma_workspace_floats(int64_t B, int F, int H, int D, int L, int masked);
int bwd_mfma(const CinDims& c, const float* x0, const float* xk, const float* w, const float* packed_dx,
             const float* d_xnext, const float* d_pooled, int pool_stride, int pool_off, float* d_xk,
             float* d_x0, float* d_w, float* ws, int phases, hipStream_t st, const float* add0, const float* add1);

// filter images packed ahead of the layer calls (xdfm_cin_pack*)
void pack_floats(int64_t B, int F, int H, int D, int L, int masked, size_t* fwd_floats, size_t* bwd_floats);
int pack_all(const xdfm_cin_pack_job* jobs, int n_jobs, int64_t B, int D, hipStream_t st);

}  // namespace xdfm_cin
