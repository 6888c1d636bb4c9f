// Microbenchmark: sustained fp32 rates on gfx950 for the candidate CIN inner loops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ITERS 4096

__global__ __launch_bounds__(256) void k_fma_sgpr(float* out, const float* w) {
  float acc[20];
  for (int n = 0; n < 20; ++n) acc[n] = 0.f;
  float z = threadIdx.x * 1e-3f;
  for (int it = 0; it < ITERS; ++it) {
    const float* wr = w + (it & 63) * 20;
#pragma unroll
    for (int n = 0; n < 20; ++n) acc[n] = fmaf(z, wr[n], acc[n]);
    z += 1e-6f;
  }
  float s = 0;
  for (int n = 0; n < 20; ++n) s += acc[n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma_vgpr(float* out, const float* w) {
  float acc[20], wv[20];
  for (int n = 0; n < 20; ++n) { acc[n] = 0.f; wv[n] = w[n + threadIdx.x]; }
  float z = threadIdx.x * 1e-3f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int n = 0; n < 20; ++n) acc[n] = fmaf(z, wv[n], acc[n]);
    z += 1e-6f;
  }
  float s = 0;
  for (int n = 0; n < 20; ++n) s += acc[n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pkfma(float* out, const float* w) {
  f32x2 acc[10], wv[10];
  for (int n = 0; n < 10; ++n) { acc[n] = {0.f, 0.f}; wv[n] = {w[2 * n + threadIdx.x], w[2 * n + 1 + threadIdx.x]}; }
  f32x2 z = {threadIdx.x * 1e-3f, threadIdx.x * 1e-3f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int n = 0; n < 10; ++n) acc[n] = __builtin_elementwise_fma(z, wv[n], acc[n]);
    z += 1e-6f;
  }
  float s = 0;
  for (int n = 0; n < 10; ++n) s += acc[n][0] + acc[n][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mfma4(float* out, const float* w) {
  f32x4 acc[5];
  float wv[5];
  for (int n = 0; n < 5; ++n) { acc[n] = {0, 0, 0, 0}; wv[n] = w[n * 4 + (threadIdx.x & 3)]; }
  float z = threadIdx.x * 1e-3f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int n = 0; n < 5; ++n) acc[n] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[n], z, acc[n], 0, 0, 0);
    z += 1e-6f;
  }
  float s = 0;
  for (int n = 0; n < 5; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mfma16(float* out, const float* w) {
  f32x4 acc[4];
  float wv = w[threadIdx.x & 63];
  for (int n = 0; n < 4; ++n) acc[n] = {0, 0, 0, 0};
  float z = threadIdx.x * 1e-3f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(z, wv, acc[n], 0, 0, 0);
    z += 1e-6f;
  }
  float s = 0;
  for (int n = 0; n < 4; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mfma32(float* out, const float* w) {
  f32x16 acc[2];
  float wv = w[threadIdx.x & 63];
  for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0;
  float z = threadIdx.x * 1e-3f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, wv, acc[n], 0, 0, 0);
    z += 1e-6f;
  }
  float s = 0;
  for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, double flops_per_thread, int blocks, float* out, float* w) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, w);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, w);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double tf = flops_per_thread * 256.0 * blocks * 5 / (ms * 1e-3) / 1e12;
  printf("%-12s blocks=%5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms / 5, tf);
}

int main() {
  float *out, *w;
  hipMalloc(&out, 256 * 8192 * 4); hipMalloc(&w, 1 << 20);
  std::vector<float> h(1 << 18, 0.001f);
  hipMemcpy(w, h.data(), 1 << 20, hipMemcpyHostToDevice);
  for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
    run("fma_sgpr", k_fma_sgpr, 2.0 * 20 * ITERS, blocks, out, w);
    run("fma_vgpr", k_fma_vgpr, 2.0 * 20 * ITERS, blocks, out, w);
    run("pk_fma", k_pkfma, 2.0 * 20 * ITERS, blocks, out, w);
    run("mfma4x4x1", k_mfma4, 5.0 * 512 / 64 * ITERS, blocks, out, w);
    run("mfma16x16x4", k_mfma16, 4.0 * 2048 / 64 * ITERS, blocks, out, w);
    run("mfma32x32x2", k_mfma32, 2.0 * 4096 / 64 * ITERS, blocks, out, w);
  }
  return 0;
}
