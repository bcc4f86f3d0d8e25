// MFMA issue-rate probe: how many ns does a v_mfma_f32_16x16x4_f32 take per SIMD, as a function of kernel length?
// (short kernels vs long ones: clock ramp / power management; waves per SIMD: issue interleaving)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
  }
  float s = 0;
  for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD: blocks of 4 waves, wps blocks per CU
    for (int iters : {64, 256, 1024, 8192, 65536}) {
      const int grid = 256 * wps;
      k<<<grid, 256>>>(out, iters, 1.f, 2.f);
      hipDeviceSynchronize();
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<<<grid, 256>>>(out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double mfma_per_simd = (double)iters * 4 * wps;
      printf("waves/SIMD %d  iters %6d  %.1f us  %.2f ns per MFMA per SIMD (32 cycles at 2.4 GHz = 13.3 ns)\n", wps, iters,
             best * 1e3, best * 1e6 / mfma_per_simd);
    }
  }
  return 0;
}
