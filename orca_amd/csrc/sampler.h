// Wave-level most-recent-K temporal neighbour sampler (device code shared by csr.hip and engine.hip).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// Sampler: one 64-lane wavefront per (node, t) query.
//   1. 64-ary lower_bound on the node's float64 timestamps: every round the 64 lanes probe 64
//      evenly spaced positions, a ballot counts the probes that are < t, and the range shrinks
//      64x (5 dependent rounds cover 2^30 entries instead of 30 for a scalar binary search).
//   2. lanes 0..K-1 gather the K entries ending at the bound (contiguous in HBM) and write the
//      right-aligned, zero-padded outputs plus the fused Δt.
// Algorithmic bytes per query: 16 (indptr pair) + 8*ceil(log2(deg+1)) (probes) + 16K read + 16K written.
// Narrows [lo, hi] with 64-ary rounds until at most `window` candidates remain.
// invariant: answer in [lo, hi]; elements < lo are < t, element hi (if any) is >= t
__device__ __forceinline__ void wave_narrow(const double* __restrict__ ts, int64_t& lo, int64_t& hi, double t, int lane,
                                            int window) {
  while (hi - lo > window) {
    int64_t n = hi - lo;
    int64_t step = (n + 63) >> 6;
    int64_t p = lo + (int64_t)(lane + 1) * step - 1;
    bool inb = p < hi;
    bool less = inb ? (ts[p] < t) : false;
    unsigned long long m = __ballot(less);
    int c = __popcll(m);  // probes are monotone: the first c lanes are true
    int64_t nlo = lo + (int64_t)c * step;
    int64_t nhi = nlo + step - 1;   // position of the first probe that is >= t
    if (nhi > hi) nhi = hi;
    lo = nlo < hi ? nlo : hi;
    hi = nhi;
  }
}

// Fence index (csr.hip builds it with the CSR): level 1 holds the LAST timestamp of every aligned block of
// ORCA_FENCE_B0 = 32 entries of ts, level k+1 the last entry of every aligned block of 64 entries of level k.  A
// 64-ary round of wave_narrow touches 64 different cache lines (and, in a hub's row, 64 different pages of a
// gigabyte-sized array) to use 8 bytes of each: 11x the algorithmic bytes of the sampler.  With the
// fences a round reads 64 CONTIGUOUS doubles instead.  Blocks are aligned to the global array, not to rows: for a row
// [lo, hi) the fences with index in R_1 = [lo/32, hi/32) are entries of that row (fence j is entry 32j+31), so they
// are sorted, and "the first fence >= t" names the block that holds the bound (clipped to the row; no such fence:
// the bound is in the row's tail).  R_{k+1} = [s_k/64, e_k/64) in the same way.  The search walks down from the
// first level whose range fits one wave; every level is one round trip, the result is a window of <= 32 entries.
__device__ __forceinline__ void fence_narrow(const double* __restrict__ ts, const orca_fences& F, int64_t& lo, int64_t& hi,
                                             double t, int lane, int window) {
  if (hi - lo <= window) return;
  if (!F.lvl[0]) { wave_narrow(ts, lo, hi, t, lane, window); return; }
  // R_{k+1} = [lo >> (5 + 6k), hi >> (5 + 6k)): chained floor divisions are shifts (ORCA_FENCE_B0 = 32, then 64 a level)
  static_assert(ORCA_FENCE_B0 == 32, "the shifts below assume 32-entry blocks");
  int top = 0;
#pragma unroll
  for (int k = 1; k < ORCA_FENCE_LEVELS; ++k)
    if (top == k - 1 && (hi >> (6 * k - 1)) - (lo >> (6 * k - 1)) > 64 && F.lvl[k]) top = k;
  int64_t a = lo >> (5 + 6 * top), b = hi >> (5 + 6 * top);
  if (b - a > 64) {       // (a row that needs more than ORCA_FENCE_LEVELS levels: not below 2^35 entries)
    const int64_t end = b;
    wave_narrow(F.lvl[top], a, b, t, lane, 63);
    if (b < end) ++b;     // fence b is >= t: it stays in the range, so that "none found" keeps meaning "the tail"
  }
#pragma unroll
  for (int k = ORCA_FENCE_LEVELS - 1; k >= 0; --k) {
    if (k > top) continue;
    const int64_t p = a + lane;
    const bool less = p < b ? (F.lvl[k][p] < t) : false;
    const int64_t j = a + __popcll(__ballot(less));          // first fence of [a, b) that is >= t, or b
    const bool found = j < b;
    if (k > 0) {
      const int64_t sk = lo >> (6 * k - 1), ek = hi >> (6 * k - 1);       // R_k
      a = j * 64 > sk ? j * 64 : sk;
      b = found && j * 64 + 64 < ek ? j * 64 + 64 : ek;
    } else {
      lo = j * ORCA_FENCE_B0 > lo ? j * ORCA_FENCE_B0 : lo;
      hi = found && j * ORCA_FENCE_B0 + ORCA_FENCE_B0 < hi ? j * ORCA_FENCE_B0 + ORCA_FENCE_B0 : hi;
    }
  }
  if (hi - lo > window) wave_narrow(ts, lo, hi, t, lane, window);  // (window < 32: K > 32)
}

__device__ __forceinline__ int64_t wave_lower_bound(const double* __restrict__ ts, const orca_fences& F, int64_t lo,
                                                    int64_t hi, double t, int lane) {
  fence_narrow(ts, F, lo, hi, t, lane, 64);
  int64_t p = lo + lane;
  bool less = (p < hi) ? (ts[p] < t) : false;
  return lo + __popcll(__ballot(less));
}

// One wavefront answers one query (v, t): lower_bound on v's timestamps, then lanes 0..K-1 write row `q` of the
// right-aligned, zero-padded outputs plus the fused delta-t.  With claim.slot set the wave also runs the first
// pass of the distinct-id work-list over [neighbours (Q*K positions) ; roots (Q positions)] (state.hip).
__device__ __forceinline__ void sample_recent_wave(const int64_t* __restrict__ indptr, const int32_t* __restrict__ nbr,
                                                   const int32_t* __restrict__ eid, const double* __restrict__ ts,
                                                   const orca_fences& F, int64_t n_rows, int64_t v, double t, int64_t q, int64_t Q, int K,
                                                   int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid,
                                                   float* __restrict__ o_et, float* __restrict__ o_dt, int* err_flag,
                                                   const orca_claim& claim, int lane) {
  if (claim.slot) {
    if (q == 0 && lane == 0) *claim.count = 0;
    if (lane == 0 && v >= 0 && v < claim.n_nodes && (!claim.flags || claim.flags[v])) claim.slot[v] = claim.epoch + Q * K + q;
  }
  if (K <= 32) {
    // Last round and gather in ONE memory round trip: the search stops at a window of <= 64 - K candidates, and
    // the 64 lanes load timestamps, neighbours and edge ids of the entries [lo - K, lo - K + 64), which contain
    // both the candidates and the K entries before any bound among them.  The bound is a ballot away and slot j
    // of the answer is a shuffle from lane (bound - lo) + j.  Entries before the row start load as zeros: they
    // are exactly the padding of a short history.
    int64_t row_lo = 0, lo = 0, hi = 0;
    const bool ok = v >= 0 && v < n_rows;
    if (ok) {
      row_lo = lo = indptr[v];
      hi = indptr[v + 1];
      fence_narrow(ts, F, lo, hi, t, lane, 64 - K);
    } else if (err_flag && lane == 0) {
      *err_flag = 1;
    }
    const int64_t p = lo - K + lane;
    const bool have = ok && p >= row_lo && p < hi;
    const double tv = have ? ts[p] : 0.0;
    int32_t n = have ? nbr[p] : 0;
    int32_t e = have ? eid[p] : 0;
    const int c = __popcll(__ballot(have && p >= lo && tv < t));   // bound = lo + c
    float et = (float)tv;
    const int from = (c + lane) & 63;                              // lane j < K reads entry (lo + c) - K + j
    n = __shfl(n, from, 64);
    e = __shfl(e, from, 64);
    et = __shfl(et, from, 64);
    if (lane < K) {
      const int64_t o = q * K + lane;
      o_nbr[o] = n;
      o_eid[o] = e;
      o_et[o] = et;
      if (o_dt) o_dt[o] = (float)(t - (double)et);   // embedding_module.py:130-131
      if (claim.slot && n >= 0 && n < claim.n_nodes && (!claim.flags || claim.flags[n])) claim.slot[n] = claim.epoch + o;
    }
    return;
  }
  int64_t lo = 0, i = 0;
  if (v >= 0 && v < n_rows) {
    lo = indptr[v];
    int64_t hi = indptr[v + 1];
    i = wave_lower_bound(ts, F, lo, hi, t, lane);
  } else if (err_flag && lane == 0) {
    *err_flag = 1;
  }
  for (int j = lane; j < K; j += 64) {
    int64_t idx = i - K + j;
    int32_t n = 0, e = 0;
    float et = 0.0f;
    if (idx >= lo) {
      n = nbr[idx];
      e = eid[idx];
      et = (float)ts[idx];
    }
    int64_t o = q * K + j;
    o_nbr[o] = n;
    o_eid[o] = e;
    o_et[o] = et;
    if (o_dt) o_dt[o] = (float)(t - (double)et);   // embedding_module.py:130-131
    if (claim.slot && n >= 0 && n < claim.n_nodes && (!claim.flags || claim.flags[n])) claim.slot[n] = claim.epoch + o;
  }
}

// ---------------------------------------------------------------------------------------------
// Uniform sampling (not in the reference; defined here, pinned against the oracle's restatement).
// For a query with i > 0 entries before t, slot j draws r_j = mix(seed, key, j) mod i, key = the query's index in
// the caller's numbering (the GLOBAL root index in the step driver, so that a rank's slice draws what a single
// rank would); the K draws are then ordered by position (== by (ts, position)) so the output is time-ordered
// like the recent sampler's.  i == 0 gives an all-padding row.  K <= 64.
__host__ __device__ static inline uint64_t orca_mix64(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ void sample_uniform_wave(const int64_t* __restrict__ indptr, const int32_t* __restrict__ nbr,
                                                    const int32_t* __restrict__ eid, const double* __restrict__ ts,
                                                    const orca_fences& F, int64_t n_rows, int64_t v, double t, int64_t q, int64_t Q, int K,
                                                    uint64_t seed, int64_t key, int32_t* __restrict__ o_nbr,
                                                    int32_t* __restrict__ o_eid, float* __restrict__ o_et,
                                                    float* __restrict__ o_dt, int* err_flag, const orca_claim& claim,
                                                    int lane) {
  if (claim.slot) {
    if (q == 0 && lane == 0) *claim.count = 0;
    if (lane == 0 && v >= 0 && v < claim.n_nodes && (!claim.flags || claim.flags[v])) claim.slot[v] = claim.epoch + Q * K + q;
  }
  int64_t lo = 0, cnt = 0;
  if (v >= 0 && v < n_rows) {
    lo = indptr[v];
    cnt = wave_lower_bound(ts, F, lo, indptr[v + 1], t, lane) - lo;
  } else if (err_flag && lane == 0) {
    *err_flag = 1;
  }
  // lane j owns draw j; rank it among the K draws (ties broken by slot) to sort
  int64_t r = -1;
  if (lane < K && cnt > 0) r = (int64_t)(orca_mix64(seed ^ orca_mix64((uint64_t)key * 64u + (uint64_t)lane)) % (uint64_t)cnt);
  int rank = 0;
  for (int j = 0; j < K; ++j) {
    int64_t rj = __shfl(r, j, 64);
    rank += (rj < r) || (rj == r && j < lane);
  }
  if (lane < K) {
    int32_t n = 0, e = 0;
    float et = 0.0f;
    if (cnt > 0) {
      n = nbr[lo + r];
      e = eid[lo + r];
      et = (float)ts[lo + r];
    } else {
      rank = lane;
    }
    int64_t o = q * K + rank;
    o_nbr[o] = n;
    o_eid[o] = e;
    o_et[o] = et;
    if (o_dt) o_dt[o] = (float)(t - (double)et);
    if (claim.slot && n >= 0 && n < claim.n_nodes && (!claim.flags || claim.flags[n])) claim.slot[n] = claim.epoch + o;
  }
}
