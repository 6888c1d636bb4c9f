// Microbenchmark: how much VALU / LDS / SALU work can ride along with fp32 MFMAs on one SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ub_mfma_mix.hip -o tools/ub_mfma_mix && tools/ub_mfma_mix
// Each wave runs ITERS trips of 4 independent v_mfma_f32_16x16x4_f32 chains; between the MFMAs
// the trip carries NV independent VALU ops and NL ds_read_b32 per MFMA.  Reported: fraction of
// the 32-cycle-per-MFMA issue rate, for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2048

template <int NV, int NL, bool DEP>
__global__ __launch_bounds__(256) void k_mix(float* out, int stride) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f;
  float v[8];
  for (int n = 0; n < 8; ++n) v[n] = threadIdx.x + n;
  float b[4] = {1.f, 2.f, 3.f, 4.f};
  int addr = threadIdx.x & 63;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float bn = b[n];
      if (NL > 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) b[n] = lds[(addr + (it + n + l) * stride) & 4095];  // used next trip
      }
      acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bn, acc[n], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        if (DEP)
          v[0] = fmaf(v[0], 1.0001f, 0.5f);  // one serial chain
        else
          v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int n = 0; n < 4; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  for (int n = 0; n < 8; ++n) s += v[n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + b[0] + b[1] + b[2] + b[3];
}

template <int NV, int NL, bool DEP>
void run(const char* name, float* out) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double clk = p.clockRate * 1e3;  // Hz
  printf("%-28s", name);
  for (int wps = 1; wps <= 4; wps *= 2) {
    dim3 grid(cus * wps);  // 256-thread blocks: wps waves per SIMD
    hipLaunchKernelGGL((k_mix<NV, NL, DEP>), grid, dim3(256), 0, 0, out, 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<NV, NL, DEP>), grid, dim3(256), 0, 0, out, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)ITERS * 4 * wps;
    const double cyc = ms * 1e-3 * clk;
    printf("  %dw/SIMD: %5.1f cyc/MFMA (%3.0f%%)", wps, cyc / mfma_per_simd, 3200.0 * mfma_per_simd / cyc);
  }
  printf("\n");
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 64 * sizeof(float));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("CUs %d clock %.0f MHz\n", p.multiProcessorCount, p.clockRate / 1e3);
  run<0, 0, false>("mfma only", out);
  run<2, 0, false>("+2 valu/mfma", out);
  run<4, 0, false>("+4 valu/mfma", out);
  run<6, 0, false>("+6 valu/mfma", out);
  run<8, 0, false>("+8 valu/mfma", out);
  run<12, 0, false>("+12 valu/mfma", out);
  run<4, 0, true>("+4 dependent valu/mfma", out);
  run<0, 1, false>("+1 lds/mfma", out);
  run<0, 2, false>("+2 lds/mfma", out);
  run<4, 1, false>("+4 valu +1 lds", out);
  run<6, 2, false>("+6 valu +2 lds", out);
  return 0;
}
