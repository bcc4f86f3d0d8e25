// Which element of the 16x16 result does lane l, register i of v_mfma_f64_16x16x4f64 hold?  (and: operand layout check)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
  const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
  // assumed operand layout (as f32 16x16x4): A lane -> (row r, k = kq), B lane -> (k = kq, col r)
  f64x4 rows = {0, 0, 0, 0}, cols = {0, 0, 0, 0};
  rows = __builtin_amdgcn_mfma_f64_16x16x4f64(kq == 0 ? (double)r : 0.0, kq == 0 ? 1.0 : 0.0, rows, 0, 0, 0);   // C[m][n] = m
  cols = __builtin_amdgcn_mfma_f64_16x16x4f64(kq == 0 ? 1.0 : 0.0, kq == 0 ? (double)r : 0.0, cols, 0, 0, 0);   // C[m][n] = n
  for (int i = 0; i < 4; ++i) { out[(lane * 4 + i) * 2] = rows[i]; out[(lane * 4 + i) * 2 + 1] = cols[i]; }
}
int main() {
  double* d; hipMalloc(&d, 64 * 4 * 2 * 8);
  k<<<1, 64>>>(d);
  double h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int lane : {0, 1, 15, 16, 17, 32, 48, 63})
    for (int i = 0; i < 4; ++i) printf("lane %2d reg %d -> row %2.0f col %2.0f\n", lane, i, h[(lane * 4 + i) * 2], h[(lane * 4 + i) * 2 + 1]);
  return 0;
}
