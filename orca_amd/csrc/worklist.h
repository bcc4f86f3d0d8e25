// Distinct-id work-list in two passes without same-address atomics (hub ids repeat thousands of times
// in a power-law batch, and atomics on one address serialise at L2):
//   claim : every position p stores slot[id] = epoch + p with a plain store -- some one value survives;
//   append: the position whose value survived is the id's unique representative; representatives are
//           appended with one atomicAdd per 1024-thread workgroup.
// Shared by k_worklist_append (state.hip) and the append role of the native Path C step (engine.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ int64_t worklist_id(const int32_t* ids32, int64_t n32, const int64_t* ids64, int64_t i) {
  return i < n32 ? (int64_t)ids32[i] : ids64[i - n32];
}
// one 1024-thread workgroup takes candidates [1024 * block, 1024 * block + 1024) of [ids32 (n32) ; ids64 (n64)]
__device__ __forceinline__ void worklist_append_block(const int32_t* __restrict__ ids32, int64_t n32,
                                                      const int64_t* __restrict__ ids64, int64_t n64,
                                                      const uint8_t* __restrict__ flags, const int64_t* __restrict__ slot,
                                                      int64_t epoch, int64_t n_nodes, int32_t* worklist, int32_t* count,
                                                      int64_t block, int* wave_n, int* block_base) {
  // one atomicAdd per 1024-thread workgroup: the counter is a single address, and per-wave atomics on it
  // serialise at L2 (704 of them cost more than the rest of the kernel)
  const int64_t i = block * 1024 + threadIdx.x;
  bool take = false;
  int64_t id = -1;
  if (i < n32 + n64) {
    id = worklist_id(ids32, n32, ids64, i);
    take = id >= 0 && id < n_nodes && (!flags || flags[id]) && slot[id] == epoch + i;
  }
  const unsigned long long m = __ballot(take);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_n[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) { int c = wave_n[w]; wave_n[w] = tot; tot += c; }
    *block_base = tot ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  if (take) worklist[*block_base + wave_n[wave] + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)id;
}
