// Activations of `_activate` (base_model.py:339-353) and their derivatives through the saved output.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/xdfm.h"

namespace {

__device__ __forceinline__ float activate(float v, int act) {
  switch (act) {
    case XDFM_ACT_RELU: return v > 0.f ? v : 0.f;
    case XDFM_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case XDFM_ACT_TANH: return tanhf(v);
    case XDFM_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

// derivative expressed through the saved output y = act(z)
__device__ __forceinline__ float act_grad(float y, int act) {
  switch (act) {
    case XDFM_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case XDFM_ACT_SIGMOID: return y * (1.f - y);
    case XDFM_ACT_TANH: return 1.f - y * y;
    case XDFM_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    default: return 1.f;
  }
}

}  // namespace
