// Offline embedding-cache planner ON THE DEVICE: read_batches + MRD of the reference's utils/cache.py:19-166, the
// same plan as the host restatement in plan.hip (orca_mrd_plan_host), entry for entry.
//
// Pass 1 (every batch at once, HBM-bound sorts): the targets [src;dst] and the sampled neighbours of all training
// interactions become 64-bit keys (batch << 32 | id); a radix sort groups them per batch, a binary search drops the
// in-batch neighbours (cache.py:58-67), and the first key of every run -- one (id, batch, kind) event per unique
// target / unique out-of-batch neighbour -- is sorted by (id, batch) into the per-node occurrence lists
// (cache.py:80-87).
// Pass 2 (sequential over the batches: the cache after batch b decides what batch b+1 reuses): ONE workgroup walks
// the batches; per batch it gathers the candidates {uncached out-of-batch neighbours, cached ids, targets}
// (cache.py:121-127), looks up each candidate's next use as a neighbour (:129-146), sorts (distance, id) keys in
// registers (bitonic, up to 16384 candidates) and keeps the `budget` nearest (:150-157).  Equal distances are ordered by node
// id, as in plan.hip (the reference's unstable argsort leaves that order undefined: SURVEY H11).
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"

namespace {
constexpr int64_t MAX_DISTANCE = 100000000;      // cache.py:95
constexpr int PLAN_THREADS = 1024;
constexpr int PLAN_KEYS = 16384;                  // candidates of one batch the LDS sort holds (128 KiB)
constexpr uint64_t SENT = ~0ull;

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// key of target i of [src ; dst]: (batch << 32) | id
__global__ void k_target_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int64_t bs,
                              int64_t n_nodes, uint64_t* __restrict__ keys, int* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * E) return;
  const int64_t e = i < E ? i : i - E;
  const int64_t v = i < E ? src[e] : dst[e];
  if (v < 0 || v >= n_nodes) { *err = 1; keys[i] = SENT; return; }
  keys[i] = ((uint64_t)(e / bs) << 32) | (uint64_t)v;
}

// key of sampled neighbour j (side-major, [E, K] per side), or SENT for padding and in-batch neighbours
__global__ void k_out_keys(const int32_t* __restrict__ nbr_src, const int32_t* __restrict__ nbr_dst, int64_t E, int K,
                           int64_t bs, int64_t n_nodes, const uint64_t* __restrict__ tk, int64_t n_tk,
                           uint64_t* __restrict__ keys, int* __restrict__ err) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = E * K;
  if (j >= 2 * per) return;
  const int64_t r = j < per ? j : j - per;
  const int64_t v = (j < per ? nbr_src : nbr_dst)[r];
  uint64_t key = SENT;
  if (v != 0) {                                                       // cache.py:51
    if (v < 0 || v >= n_nodes) {
      *err = 1;
    } else {
      key = ((uint64_t)((r / K) / bs) << 32) | (uint64_t)v;
      const int64_t p = lower_bound_u64(tk, n_tk, key);
      if (p < n_tk && tk[p] == key) key = SENT;                        // a target of its own batch (:58-67)
    }
  }
  keys[j] = key;
}

// one event per run of equal (batch, id) keys: (id << 33) | (batch << 1) | kind  (kind 0 target, 1 neighbour)
__global__ void k_events(const uint64_t* __restrict__ keys, int64_t n, int kind, uint64_t* __restrict__ ev) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  ev[i] = (k != SENT && (i == 0 || keys[i - 1] != k)) ? (((k & 0xffffffffull) << 33) | ((k >> 32) << 1) | (uint64_t)kind) : SENT;
}

// ptr[b] = first position of a key >= (b << shift), b = 0 .. n_ptr-1
__global__ void k_pointers(const uint64_t* __restrict__ keys, int64_t n, int shift, int64_t n_ptr, int64_t* __restrict__ ptr) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_ptr) return;
  ptr[b] = lower_bound_u64(keys, n, (uint64_t)b << shift);
}

// ---- the (distance, id) sort of a batch's candidates: bitonic, ascending, keys in REGISTERS -----------------------------
// Element i of the P2 = 1024 R keys lives in register i / 1024 of thread i % 1024.  A compare-exchange at distance j then
// is, for j >= 1024, between two registers of one thread; for j < 64, between two lanes of one wave (two 32-bit
// shuffles); and only for 64 <= j < 1024 between two waves: through LDS, two barriers.  Of the 66 stages of a 2048-key
// sort that is 1 + 51 + 14 -- the same sort kept in LDS with a barrier per stage took 29 of a batch's 44 us, this one
// takes 22 of 36: what is left is the vector instructions of 2048 x 66 compare-exchanges on ONE CU's four SIMDs.
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int j) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, j, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), j, 64);
  return ((uint64_t)hi << 32) | lo;
}
template <int R, int JR>
__device__ __forceinline__ void bitonic_in_thread(uint64_t (&a)[R], int tid, int k) {
  if constexpr (JR < R) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((r & JR) == 0) {
        const bool asc = ((tid + 1024 * r) & k) == 0;
        const uint64_t x = a[r], y = a[r | JR];
        const bool sw = asc ? x > y : x < y;
        a[r] = sw ? y : x;
        a[r | JR] = sw ? x : y;
      }
  }
}
// n2: a power of two, 1024 <= n2 <= 1024 R: the keys from n2 on are all SENT (padding) and stay where they are
template <int R>
__device__ __forceinline__ void bitonic_regs(uint64_t (&a)[R], uint64_t* lds, int tid, int n2) {
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 1024) {
        switch (j >> 10) {      // (uniform)
          case 1: bitonic_in_thread<R, 1>(a, tid, k); break;
          case 2: bitonic_in_thread<R, 2>(a, tid, k); break;
          case 4: bitonic_in_thread<R, 4>(a, tid, k); break;
          default: bitonic_in_thread<R, 8>(a, tid, k); break;
        }
      } else if (j >= 64) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (1024 * r < n2) lds[tid + 1024 * r] = a[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (1024 * r >= n2) continue;      // (uniform)
          const int i = tid + 1024 * r;
          const uint64_t c = lds[i ^ j];
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          a[r] = keep_min ? (a[r] < c ? a[r] : c) : (a[r] > c ? a[r] : c);
        }
        __syncthreads();
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (1024 * r >= n2) continue;      // (uniform)
          const int i = tid + 1024 * r;
          const uint64_t c = shfl_xor_u64(a[r], j);
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          a[r] = keep_min ? (a[r] < c ? a[r] : c) : (a[r] > c ? a[r] : c);
        }
      }
    }
}

struct PlanState {          // carried from one launch of k_mrd_batches to the next
  int32_t cached_n;
  int32_t overflow;         // a batch had more candidates than PLAN_KEYS
  int32_t grow;             // a batch had more candidates than the launch's instantiation sorts (1024 R): the walk stopped
  int32_t pad_;
  long long next_b;         // ... in front of this batch
  long long n_reuse, n_recompute, total_rd;
};

template <int R>
__global__ __launch_bounds__(PLAN_THREADS) void k_mrd_batches(
    int64_t b0, int64_t b1, int64_t budget, const uint64_t* __restrict__ tk, const int64_t* __restrict__ tptr,
    const uint64_t* __restrict__ ok, const int64_t* __restrict__ optr, const uint64_t* __restrict__ ev,
    const int64_t* __restrict__ occ_ptr, uint8_t* __restrict__ cache_flag, int32_t* __restrict__ time_flag,
    int32_t* __restrict__ pos, int32_t* __restrict__ fresh, int32_t* __restrict__ mark, int32_t* __restrict__ cached,
    PlanState* __restrict__ st, int64_t* __restrict__ plan_out, int64_t* __restrict__ plan_len) {
  extern __shared__ uint64_t keys[];                 // PLAN_KEYS
  __shared__ int n_cand;
  const int tid = threadIdx.x;
  long long reuse_all = 0, recompute_all = 0, rd_all = 0;
  int cached_n = st->cached_n;
  if (st->overflow) return;
  auto add = [&](uint32_t v) {
    const int idx = atomicAdd(&n_cand, 1);
    if (idx < PLAN_KEYS) keys[idx] = v;
  };
  for (int64_t b = b0; b < b1; ++b) {
    long long reuse = 0, recompute = 0, rd = 0;      // of this batch: counted once it has been sorted (see `grow`)
    if (tid == 0) n_cand = 0;
    __syncthreads();
    // out-of-batch neighbours of batch b (with multiplicity): reuse / recompute counts, uncached ones are candidates.
    // One workgroup walks the batches, so what a batch costs is its chains of dependent loads: the loop is taken four
    // positions per thread at a time, every load of a level requested before the first is looked at (clamped positions,
    // masked results: no branch around a load).
    {
      const int64_t o_lo = optr[b], o_hi = optr[b + 1];
      for (int64_t i0 = o_lo + tid; i0 < o_hi; i0 += 4 * PLAN_THREADS) {
        uint64_t k[4], kp[4];
        bool in[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t i = i0 + (int64_t)r * PLAN_THREADS;
          in[r] = i < o_hi;
          const int64_t ic = in[r] ? i : o_lo;
          k[r] = ok[ic];
          kp[r] = ok[ic > o_lo ? ic - 1 : o_lo];
        }
        uint8_t cf[4];
        int32_t tf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cf[r] = cache_flag[(uint32_t)k[r]];
          tf[r] = time_flag[(uint32_t)k[r]];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!in[r]) continue;
          const int64_t i = i0 + (int64_t)r * PLAN_THREADS;
          const uint32_t v = (uint32_t)k[r];
          if (cf[r]) { ++reuse; rd += b - tf[r]; }                         // cache.py:113-119
          else {
            ++recompute;
            if (i == o_lo || kp[r] != k[r]) { fresh[v] = (int32_t)b; add(v); }
          }
        }
      }
    }
    // unique targets of batch b: new_computed (:124-125)
    for (int64_t i = tptr[b] + tid; i < tptr[b + 1]; i += PLAN_THREADS) {
      const uint64_t k = tk[i];
      if (i == tptr[b] || tk[i - 1] != k) {
        const uint32_t v = (uint32_t)k;
        mark[v] = (int32_t)b;
        fresh[v] = (int32_t)b;
        add(v);
      }
    }
    __syncthreads();
    // the ids cached after the previous batch, unless they are targets of this one (already candidates); a cached id
    // cannot be among the uncached neighbours
    for (int i = tid; i < cached_n; i += PLAN_THREADS) {
      const int32_t v = cached[i];
      if (mark[v] != (int32_t)b) add((uint32_t)v);
    }
    __syncthreads();
    const int C = n_cand;
    if (C > PLAN_KEYS) {                 // uniform: every thread sees the same count
      if (tid == 0) st->overflow = 1;
      break;
    }
    if (C > 1024 * R) {                  // more candidates than this instantiation sorts: the host re-launches from batch b
      if (tid == 0) { st->grow = 1; st->next_b = b; }      // with the next size (nothing of batch b has been changed that
      break;                                                // its second walk does not set again)
    }
    // next use as a neighbour (:129-146); key = (distance << 32) | id.  The thread's R candidates at a time, level by level
    // as above: the occurrence-list bounds and the cursor, then the event under the cursor; a cursor that has to move on
    // (the id has events in batches up to b that this walk has not passed yet) finishes in the loop below.
    uint64_t key[R];
    constexpr int G = R < 4 ? R : 4;
#pragma unroll
    for (int g = 0; g < R; g += G) {
      uint32_t v[G];
      bool in[G];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        const int i = tid + 1024 * (g + r);
        in[r] = i < C;
        v[r] = in[r] ? (uint32_t)keys[i] : 0u;
      }
      int64_t o0[G], o1[G], p[G];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        o0[r] = occ_ptr[v[r]];
        o1[r] = occ_ptr[v[r] + 1];
        p[r] = pos[v[r]];
      }
      uint64_t e[G];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        p[r] += o0[r];
        e[r] = ev[p[r] < o1[r] ? p[r] : (o1[r] > o0[r] ? o1[r] - 1 : 0)];
      }
#pragma unroll
      for (int r = 0; r < G; ++r) {
        key[g + r] = SENT;
        if (!in[r]) continue;
        while (p[r] < o1[r] && (int64_t)((e[r] >> 1) & 0xffffffffull) <= b) {
          ++p[r];
          if (p[r] < o1[r]) e[r] = ev[p[r]];
        }
        pos[v[r]] = (int32_t)(p[r] - o0[r]);
        const uint64_t d = (p[r] == o1[r] || (e[r] & 1ull) == 0) ? (uint64_t)(MAX_DISTANCE + 1)
                                                                 : (uint64_t)((int64_t)((e[r] >> 1) & 0xffffffffull) - b);
        key[g + r] = (d << 32) | v[r];
      }
    }
    __syncthreads();                     // (every candidate id has been read: the sort's exchanges reuse `keys`)
    int n2 = 1024;
    while (n2 < C) n2 <<= 1;
    bitonic_regs<R>(key, keys, tid, n2);     // ascending (:150-152)
    const int n_keep = (int64_t)C < budget ? C : (int)budget;
    for (int i = tid; i < cached_n; i += PLAN_THREADS) cache_flag[cached[i]] = 0;    // :155-156
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = tid + 1024 * r;
      if (i >= n_keep) continue;
      const uint32_t v = (uint32_t)key[r];
      plan_out[b * budget + i] = (int64_t)v;
      cache_flag[v] = 1;
      cached[i] = (int32_t)v;
      if (fresh[v] == (int32_t)b) time_flag[v] = (int32_t)b;            // :160-162
    }
    if (tid == 0) plan_len[b] = n_keep;
    cached_n = n_keep;
    reuse_all += reuse; recompute_all += recompute; rd_all += rd;
    __syncthreads();
  }
  if (tid == 0) st->cached_n = cached_n;
  atomicAdd((unsigned long long*)&st->n_reuse, (unsigned long long)reuse_all);
  atomicAdd((unsigned long long*)&st->n_recompute, (unsigned long long)recompute_all);
  atomicAdd((unsigned long long*)&st->total_rd, (unsigned long long)rd_all);
}

struct Buffers {            // everything the planner allocates, freed on every exit path
  void* p[16] = {};
  int n = 0;
  template <typename T> hipError_t alloc(T** out, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) p[n++] = q;
    *out = (T*)q;
    return e;
  }
  ~Buffers() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};

int sort_keys(uint64_t* in, uint64_t* out, size_t n, int end_bit, hipStream_t stream) {
  size_t tb = 0;
  void* tmp = nullptr;
  ORCA_HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, in, out, n, 0, end_bit, stream));
  ORCA_HIP_TRY(hipMalloc(&tmp, tb ? tb : 1));
  hipError_t e = rocprim::radix_sort_keys(tmp, tb, in, out, n, 0, end_bit, stream);
  hipError_t e2 = hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  ORCA_HIP_TRY(e);
  ORCA_HIP_TRY(e2);
  return ORCA_OK;
}
}  // namespace

ORCA_API int orca_mrd_plan_device(int64_t E, int64_t bs, int K, const int64_t* src, const int64_t* dst,
                                  const int32_t* nbr_src, const int32_t* nbr_dst, int64_t n_nodes, int64_t budget,
                                  int64_t* plan_out, int64_t* plan_len, int64_t* stats_host, double* avg_reuse_distance_host,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ORCA_REQUIRE(E >= 0 && bs > 0 && K >= 1 && n_nodes > 0 && n_nodes < (1LL << 31) && budget >= 0 && budget <= PLAN_KEYS,
               "orca_mrd_plan_device: bad sizes (budget at most %d)", PLAN_KEYS);
  if (stats_host) stats_host[0] = stats_host[1] = 0;
  if (avg_reuse_distance_host) *avg_reuse_distance_host = __builtin_nan("");
  if (E == 0) return ORCA_OK;
  ORCA_REQUIRE(src && dst && nbr_src && nbr_dst && plan_out && plan_len, "orca_mrd_plan_device: null pointer");
  const int64_t num_batch = (E + bs - 1) / bs;
  ORCA_REQUIRE(num_batch < (1LL << 31), "orca_mrd_plan_device: too many batches");
  const int64_t n_t = 2 * E, n_o = 2 * E * K, n_e = n_t + n_o;
  Buffers buf;
  uint64_t *a = nullptr, *b = nullptr, *tk = nullptr, *ok = nullptr;
  int64_t *tptr = nullptr, *optr = nullptr, *occ_ptr = nullptr;
  int* err = nullptr;
  ORCA_HIP_TRY(buf.alloc(&a, (size_t)n_e));
  ORCA_HIP_TRY(buf.alloc(&b, (size_t)n_e));
  ORCA_HIP_TRY(buf.alloc(&tk, (size_t)n_t));
  ORCA_HIP_TRY(buf.alloc(&ok, (size_t)n_o));
  ORCA_HIP_TRY(buf.alloc(&tptr, (size_t)num_batch + 1));
  ORCA_HIP_TRY(buf.alloc(&optr, (size_t)num_batch + 1));
  ORCA_HIP_TRY(buf.alloc(&occ_ptr, (size_t)n_nodes + 1));
  ORCA_HIP_TRY(buf.alloc(&err, 1));
  ORCA_HIP_TRY(hipMemsetAsync(err, 0, sizeof(int), stream));
  const int B = 256;
  int bbits = 1;
  while ((1LL << bbits) < num_batch) ++bbits;
  int rc;
  // ---- pass 1 ----
  k_target_keys<<<(unsigned)ceil_div64(n_t, B), B, 0, stream>>>(src, dst, E, bs, n_nodes, a, err);
  if ((rc = sort_keys(a, tk, (size_t)n_t, 64, stream)) != ORCA_OK) return rc;      // (SENT of a bad id sorts last)
  int h_err = 0;
  ORCA_HIP_TRY(hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost));
  if (h_err) { orca_set_error("orca_mrd_plan_device: node id out of range"); return ORCA_ERANGE; }
  k_out_keys<<<(unsigned)ceil_div64(n_o, B), B, 0, stream>>>(nbr_src, nbr_dst, E, K, bs, n_nodes, tk, n_t, a, err);
  if ((rc = sort_keys(a, ok, (size_t)n_o, 64, stream)) != ORCA_OK) return rc;
  ORCA_HIP_TRY(hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost));
  if (h_err) { orca_set_error("orca_mrd_plan_device: neighbour id out of range"); return ORCA_ERANGE; }
  k_pointers<<<(unsigned)ceil_div64(num_batch + 1, B), B, 0, stream>>>(tk, n_t, 32, num_batch + 1, tptr);
  k_pointers<<<(unsigned)ceil_div64(num_batch + 1, B), B, 0, stream>>>(ok, n_o, 32, num_batch + 1, optr);   // SENT keys lie past every batch
  k_events<<<(unsigned)ceil_div64(n_t, B), B, 0, stream>>>(tk, n_t, 0, a);
  k_events<<<(unsigned)ceil_div64(n_o, B), B, 0, stream>>>(ok, n_o, 1, a + n_t);
  if ((rc = sort_keys(a, b, (size_t)n_e, 64, stream)) != ORCA_OK) return rc;
  uint64_t* ev = b;
  k_pointers<<<(unsigned)ceil_div64(n_nodes + 1, B), B, 0, stream>>>(ev, n_e, 33, n_nodes + 1, occ_ptr);      // SENT events lie past every id
  ORCA_LAUNCH_CHECK();
  // ---- pass 2 ----
  uint8_t* cache_flag = nullptr;
  int32_t *time_flag = nullptr, *pos = nullptr, *fresh = nullptr, *mark = nullptr, *cached = nullptr;
  PlanState* st = nullptr;
  ORCA_HIP_TRY(buf.alloc(&cache_flag, (size_t)n_nodes));
  ORCA_HIP_TRY(buf.alloc(&time_flag, (size_t)n_nodes));
  ORCA_HIP_TRY(buf.alloc(&pos, (size_t)n_nodes));
  ORCA_HIP_TRY(buf.alloc(&fresh, (size_t)n_nodes));
  ORCA_HIP_TRY(buf.alloc(&mark, (size_t)n_nodes));
  ORCA_HIP_TRY(buf.alloc(&cached, (size_t)budget));
  ORCA_HIP_TRY(buf.alloc(&st, 1));
  ORCA_HIP_TRY(hipMemsetAsync(cache_flag, 0, (size_t)n_nodes, stream));
  ORCA_HIP_TRY(hipMemsetAsync(time_flag, 0, (size_t)n_nodes * sizeof(int32_t), stream));
  ORCA_HIP_TRY(hipMemsetAsync(pos, 0, (size_t)n_nodes * sizeof(int32_t), stream));
  ORCA_HIP_TRY(hipMemsetAsync(fresh, 0xff, (size_t)n_nodes * sizeof(int32_t), stream));
  ORCA_HIP_TRY(hipMemsetAsync(mark, 0xff, (size_t)n_nodes * sizeof(int32_t), stream));
  ORCA_HIP_TRY(hipMemsetAsync(st, 0, sizeof(PlanState), stream));
  static bool attr_set[64] = {};
  const size_t lds = (size_t)PLAN_KEYS * sizeof(uint64_t);
  if (orca_first_use_on_device(attr_set)) {
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_mrd_batches<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_mrd_batches<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_mrd_batches<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_mrd_batches<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_mrd_batches<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // The sort holds 1024 R keys in registers; R follows the batches: a launch that meets a batch with more candidates
  // stops in front of it, and the walk goes on from there with the next size (it never shrinks: a trace's batches are alike).
  const int64_t chunk = 1024;            // batches per launch: a launch stays in the tens of milliseconds
  int R = 1;
  while (1024 * R < budget && R < 16) R *= 2;      // (the cached ids alone are candidates)
  for (int64_t b0 = 0; b0 < num_batch;) {
    const int64_t b1 = b0 + chunk < num_batch ? b0 + chunk : num_batch;
#define PLAN_LAUNCH(RR) k_mrd_batches<RR><<<1, PLAN_THREADS, lds, stream>>>(b0, b1, budget, tk, tptr, ok, optr, ev, occ_ptr, cache_flag, \
                                                                        time_flag, pos, fresh, mark, cached, st, plan_out, plan_len)
    if (R == 1) PLAN_LAUNCH(1); else if (R == 2) PLAN_LAUNCH(2); else if (R == 4) PLAN_LAUNCH(4); else if (R == 8) PLAN_LAUNCH(8);
    else PLAN_LAUNCH(16);
#undef PLAN_LAUNCH
    ORCA_LAUNCH_CHECK();
    PlanState hs{};
    ORCA_HIP_TRY(hipMemcpyAsync(&hs, st, sizeof(PlanState), hipMemcpyDeviceToHost, stream));
    ORCA_HIP_TRY(hipStreamSynchronize(stream));
    if (hs.overflow) break;
    if (hs.grow) {
      ORCA_REQUIRE(R < 16, "orca_mrd_plan_device: a batch asks for more than %d sort keys", PLAN_KEYS);
      R *= 2;
      b0 = hs.next_b;
      ORCA_HIP_TRY(hipMemsetAsync(&st->grow, 0, sizeof(int32_t), stream));
    } else {
      b0 = b1;
    }
  }
  ORCA_LAUNCH_CHECK();
  PlanState h{};
  ORCA_HIP_TRY(hipMemcpyAsync(&h, st, sizeof(PlanState), hipMemcpyDeviceToHost, stream));
  ORCA_HIP_TRY(hipStreamSynchronize(stream));
  ORCA_REQUIRE(!h.overflow, "orca_mrd_plan_device: a batch has more than %d candidate ids (budget + uncached neighbours + "
               "targets); use orca_mrd_plan_host for batches this large", PLAN_KEYS);
  if (stats_host) { stats_host[0] = h.n_reuse; stats_host[1] = h.n_recompute; }
  if (avg_reuse_distance_host && h.n_reuse) *avg_reuse_distance_host = (double)h.total_rd / (double)h.n_reuse;
  return ORCA_OK;
}
