// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950: 256 workgroups of 1024 threads (4 waves per SIMD), each lane
// running ITER rounds of 8 independent accumulators.  Prints ns per wave-instruction per SIMD slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;
__global__ __launch_bounds__(1024) void k_fma(float* out, float a, float b) {
  float c[8];
  for (int i = 0; i < 8; ++i) c[i] = threadIdx.x * 1e-9f + i;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c[i];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
__global__ __launch_bounds__(1024) void k_pk(float* out, float a, float b) {
  f32x2 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x2{threadIdx.x * 1e-9f + i, 1.f};
  f32x2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(av), "v"(bv));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c[i].x + c[i].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (which == 0) k_fma<<<256, 1024>>>(out, 0.5f, 0.25f); else k_pk<<<256, 1024>>>(out, 0.5f, 0.25f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per SIMD: 4 waves x ITER x 8 instructions
      printf("%s: %.3f ms, %.2f ns per wave-instruction on a SIMD\n", which ? "v_pk_fma_f32" : "v_fma_f32", ms, ms * 1e6 / (4.0 * ITER * 8));
    }
  }
  return 0;
}
