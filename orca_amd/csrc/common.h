// Internal helpers shared by the HIP translation units of liborca_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/orca_hip.h"

#define ORCA_API extern "C" __attribute__((visibility("default")))

void orca_set_error(const char* fmt, ...);

#define ORCA_HIP_TRY(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      orca_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ORCA_EHIP;                                                                 \
    }                                                                                   \
  } while (0)

#define ORCA_REQUIRE(cond, ...)                                                         \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      orca_set_error(__VA_ARGS__);                                                      \
      return ORCA_EINVAL;                                                               \
    }                                                                                   \
  } while (0)

#define ORCA_LAUNCH_CHECK() ORCA_HIP_TRY(hipGetLastError())

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- packed weight format -------------------------------------------------------------------
// A matrix W with N output rows and Kd reduction columns is stored as NT = ceil(N/16) n-tiles x
// KB = ceil(Kd/16) k-blocks; each (nt, kb) block is 64 lanes x float4 = 1 KiB, lane l holding
// W[nt*16 + (l&15)][kb*16 + 4*(l>>4) + s], s = 0..3 (zero outside the matrix).  One wave-wide
// 16-byte load therefore fetches the B operands of four v_mfma_f32_16x16x4_f32 steps, fully
// coalesced.  The A operand uses the same k permutation (lane l: A[l&15][kb*16 + 4*(l>>4) + s]).
__host__ __device__ static inline int pw_tiles(int n) { return (n + 15) / 16; }
__host__ __device__ static inline int64_t pw_floats(int N, int Kd) { return (int64_t)pw_tiles(N) * pw_tiles(Kd) * 256; }

// One wave accumulates up to NT 16x16 output tiles of C = A(16 x 16*KB) * PW.
// a_row: LDS pointer to this lane's A row (row l&15) already offset by 4*(l>>4) floats.
// pw: packed matrix base; tile[t] the n-tile ids; nt <= NT of them are live (wave-uniform).
template <int NT>
__device__ __forceinline__ void wave_gemm16(const float* a_row, const f32x4* __restrict__ pw, int KB,
                                            const int (&tile)[NT], int nt, f32x4 (&acc)[NT], int lane) {
  const f32x4* bp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bp[t] = pw + (int64_t)(t < nt ? tile[t] : tile[0]) * KB * 64 + lane;
  f32x4 bcur[NT], bnext[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nt) bcur[t] = bp[t][0];
  for (int kb = 0; kb < KB; ++kb) {
    f32x4 a = *reinterpret_cast<const f32x4*>(a_row + kb * 16);
    if (kb + 1 < KB) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) bnext[t] = bp[t][(kb + 1) * 64];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bcur[t][s], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) bcur[t] = bnext[t];
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// generic strided pack kernel launcher (pack.hip)
int orca_pack_matrix(float* dst, int NT_total, int KB, int n0, int n_cnt, int Kd, const float* src,
                     int64_t sn, int64_t sk, hipStream_t stream);
