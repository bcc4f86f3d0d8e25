// Device-resident node-state bookkeeping: everything the reference does with NumPy on the host
// between its GPU ops (np.unique, boolean flag arrays, last-occurrence dedupe, index_put / index_select).
// All of it is HBM-bound integer / row-copy work: one wave per row, coalesced 4-byte lanes.
#include "common.h"

namespace {
constexpr int D = ORCA_D;

__global__ void k_time_encode(const float* __restrict__ t, int64_t n, const float* __restrict__ w,
                              const float* __restrict__ b, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  int c = (int)(i % D);
  out[i] = orca_cosf(orca_time_arg(t[i / D], w[c], b[c]));   // model/time_encoding.py:26-27
}

__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_fill_u8(uint8_t* p, int64_t n, uint8_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

#include "worklist.h"

__global__ void k_worklist_claim(const int32_t* __restrict__ ids32, int64_t n32, const int64_t* __restrict__ ids64,
                                 int64_t n64, const uint8_t* __restrict__ flags, int64_t* slot, int64_t epoch,
                                 int64_t n_nodes, int32_t* count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *count = 0;            // the append pass (next kernel on the stream) counts from zero
  if (i >= n32 + n64) return;
  int64_t id = worklist_id(ids32, n32, ids64, i);
  if (id < 0 || id >= n_nodes || (flags && !flags[id])) return;
  slot[id] = epoch + i;
}
__global__ __launch_bounds__(1024) void k_worklist_append(const int32_t* __restrict__ ids32, int64_t n32,
                                                           const int64_t* __restrict__ ids64, int64_t n64,
                                                           const uint8_t* __restrict__ flags,
                                                           const int64_t* __restrict__ slot, int64_t epoch,
                                                           int64_t n_nodes, int32_t* worklist, int32_t* count) {
  __shared__ int wave_n[16];
  __shared__ int block_base;
  worklist_append_block(ids32, n32, ids64, n64, flags, slot, epoch, n_nodes, worklist, count, blockIdx.x, wave_n, &block_base);
}

__global__ void k_set_flags(uint8_t* flags, const int64_t* ids64, const int32_t* ids32, int64_t n,
                            const int32_t* n_dev, uint8_t v, int64_t n_nodes) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev && i >= *n_dev) return;
  if (i >= n) return;
  int64_t id = ids64 ? ids64[i] : (int64_t)ids32[i];
  if (id >= 0 && id < n_nodes) flags[id] = v;       // ids outside the table are skipped, as everywhere in this file
}

// last-occurrence election: slot[id] = max over positions of (epoch + p)
__global__ void k_elect(const int64_t* __restrict__ ids, int64_t M, int64_t* slot, int64_t epoch, int64_t n_nodes) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  int64_t id = ids[p];
  if (id < 0 || id >= n_nodes) return;
  atomicMax(reinterpret_cast<unsigned long long*>(&slot[id]), (unsigned long long)(epoch + p));
}
__global__ void k_elect2(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t bs, int64_t* slot,
                         int64_t epoch, int64_t n_nodes) {   // ids = [a ; b]
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= 2 * bs) return;
  int64_t id = p < bs ? a[p] : b[p - bs];
  if (id < 0 || id >= n_nodes) return;
  atomicMax(reinterpret_cast<unsigned long long*>(&slot[id]), (unsigned long long)(epoch + p));
}
__global__ void k_winner(const int64_t* __restrict__ ids, int64_t M, const int64_t* __restrict__ slot, int64_t epoch,
                         int64_t n_nodes, uint8_t* winner) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  int64_t id = ids[p];
  winner[p] = (id >= 0 && id < n_nodes && slot[id] == epoch + p) ? 1 : 0;
}

__global__ void k_segment_mean(const int64_t* __restrict__ ids, int64_t M, int64_t Dm, const float* __restrict__ msgs,
                               const uint8_t* __restrict__ winner, float* __restrict__ out) {
  const int64_t p = blockIdx.x;
  if (!winner[p]) return;
  const int64_t id = ids[p];
  int cnt = 0;
  for (int64_t i = 0; i <= p; ++i) cnt += (ids[i] == id);
  const float inv = 1.0f / (float)cnt;
  for (int64_t c = threadIdx.x; c < Dm; c += blockDim.x) {
    float s = 0.0f;
    for (int64_t i = 0; i <= p; ++i)
      if (ids[i] == id) s += msgs[i * Dm + c];   // arrival order, as torch.mean over the stacked rows
    out[p * Dm + c] = s * inv;
  }
}

// model/tgn_model.py:172-191 + modules/memory.py:25-28, one wave per position of [src;dst]
__global__ __launch_bounds__(256) void k_message_store(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                       const int64_t* __restrict__ eidx, const double* __restrict__ t,
                                                       int64_t bs, const float* __restrict__ memory,
                                                       const float* __restrict__ last_update,
                                                       const float* __restrict__ edge_feats, int d_e,
                                                       const float* __restrict__ tw, const float* __restrict__ tb,
                                                       const int64_t* __restrict__ slot, int64_t epoch,
                                                       const uint8_t* __restrict__ winner,
                                                       int64_t n_nodes, float* __restrict__ messages,
                                                       float* __restrict__ msg_ts, uint8_t* __restrict__ pending) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= 2 * bs) return;
  const int64_t e = p < bs ? p : p - bs;
  const int64_t id = p < bs ? src[e] : dst[e];
  const int64_t partner = p < bs ? dst[e] : src[e];
  if (id < 0 || id >= n_nodes || (winner ? !winner[p] : slot[id] != epoch + p)) return;   // not the last occurrence of id
  message_store_wave(id, partner, eidx[e], t[e], memory, last_update, edge_feats, d_e, tw, tb, messages, msg_ts, pending, lane);
}

// memory_updater.update_memory (:25-39) when GRU(message, memory) of every pending id already sits in the
// overlay: one wave per position of the id list copies the row and the message timestamp.  Duplicate ids
// copy the same values (benign).  The pending flags are left alone: the caller stores new messages for the
// same ids next, which sets them again (tgn_model.py:127-141).
__global__ __launch_bounds__(256) void k_persist_overlay(const int64_t* __restrict__ ids, int64_t n,
                                                         const uint8_t* __restrict__ pending,
                                                         const float* __restrict__ overlay, float* __restrict__ memory,
                                                         float* __restrict__ last_update,
                                                         const float* __restrict__ msg_ts, int64_t n_nodes,
                                                         bool by_position) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= n) return;
  const int64_t id = ids[p];
  if (id < 0 || id >= n_nodes || !pending[id]) return;
  const float* row = overlay + (by_position ? p : id) * D;   // a table indexed by id, or rows gathered per position
  for (int c = lane; c < D; c += 64) memory[id * D + c] = row[c];
  if (lane == 0) last_update[id] = msg_ts[id];
}

__global__ __launch_bounds__(256) void k_push_rows(float* __restrict__ emb, const float* __restrict__ x,
                                                   const int64_t* __restrict__ ids, int64_t n,
                                                   const int64_t* __restrict__ slot, int64_t epoch,
                                                   const uint8_t* __restrict__ winner, int64_t n_nodes) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= n) return;
  const int64_t id = ids[p];
  if (id < 0 || id >= n_nodes || (winner ? !winner[p] : slot[id] != epoch + p)) return;
  for (int c = lane; c < D; c += 64) emb[id * D + c] = x[p * D + c];
}

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ table, int64_t n_rows, int64_t width,
                                                     const int64_t* __restrict__ idx64,
                                                     const int32_t* __restrict__ idx32, int64_t n,
                                                     float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= n) return;
  const int64_t r = idx64 ? idx64[p] : (int64_t)idx32[p];
  const bool ok = r >= 0 && r < n_rows;       // a row index outside the table reads as zeros
  for (int64_t c = lane; c < width; c += 64) out[p * width + c] = ok ? table[r * width + c] : 0.f;
}

// pos[list[i]] = i for the first min(n, *n_dev) entries of a list of distinct ids
__global__ void k_index_positions(const int32_t* __restrict__ list, int64_t n, const int32_t* __restrict__ n_dev,
                                  int32_t* __restrict__ pos, int64_t n_nodes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t m = n_dev ? (*n_dev < n ? *n_dev : n) : n;
  if (i < m && list[i] >= 0 && list[i] < n_nodes) pos[list[i]] = (int32_t)i;
}
// dst[pos[id]] += rows[p] for every position p whose id carries a flag.  Several positions share an id -- in a
// power-law batch a hub id thousands of them -- and fp32 atomics on one row serialise at L2 (95 us for 41 k rows), so a
// workgroup first GROUPS its SCAT_CHUNK positions by destination row (a hash table numbers the chunk's distinct rows, a
// counting sort over those numbers lists the positions row by row) and then sums each group in REGISTERS: the 16 waves
// split the sorted list at group boundaries, a wave loads its positions' rows together (one memory round trip), walks
// them in order and issues one global atomic row add whenever the destination changes -- a hub row receives ONE add per
// chunk instead of one per occurrence.  (The first version summed into LDS accumulator rows with ds_add_f32: 19 of its
// 34 us were those LDS atomics, 9 the global ones, 6 everything else.)
// CHUNK = positions per workgroup: sized to the list (4095 positions in chunks of 256 are 16 workgroups on a 256-CU chip).
template <int SCAT_CHUNK>
__global__ __launch_bounds__(1024) void k_scatter_add_flagged(const float* rows, const int64_t* idx64,
                                                              const int32_t* idx32, int64_t n,
                                                              const uint8_t* __restrict__ flags,
                                                              const int32_t* __restrict__ pos, int64_t n_nodes,
                                                              float* __restrict__ dst, int blocks1 = 0x7fffffff,
                                                              const float* rows2 = nullptr, const int64_t* idx64_2 = nullptr,
                                                              const int32_t* idx32_2 = nullptr, int64_t n2 = 0) {
  constexpr int SCAT_HASH = SCAT_CHUNK <= 256 ? 512 : 2 * SCAT_CHUNK;
  static_assert(SCAT_CHUNK <= 512 && SCAT_HASH <= 1024 && SCAT_CHUNK % 16 == 0, "a thread per position and per table slot, a sparse table");
  constexpr int PER = SCAT_CHUNK / 16;          // sorted positions per wave (at most)
  __shared__ int keys[SCAT_HASH], vals[SCAT_HASH];
  __shared__ int cnt[SCAT_CHUNK], start[SCAT_CHUNK];      // per distinct row of the chunk: positions, first place in the list
  __shared__ int lpos[SCAT_CHUNK], lkey[SCAT_CHUNK], lrow[SCAT_CHUNK];   // the sorted list: position in the chunk, destination row, its number
  __shared__ int n_rows, n_valid;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int64_t blk = blockIdx.x;
  if (blk >= blocks1) {     // a second list in the same launch (orca_scatter_add_flagged_pair): workgroup-uniform
    blk -= blocks1;
    rows = rows2; idx64 = idx64_2; idx32 = idx32_2; n = n2;
  }
  const int64_t base = blk * SCAT_CHUNK;
  for (int i = t; i < SCAT_HASH; i += 1024) keys[i] = -1;
  if (t < SCAT_CHUNK) cnt[t] = 0;
  if (t == 0) { n_rows = 0; n_valid = 0; }
  __syncthreads();
  // the key of this thread's position: the row of dst it adds to, or -1
  int key = -1, h = 0;
  if (t < SCAT_CHUNK) {
    const int64_t p = base + t;
    if (p < n) {
      const int64_t id = idx64 ? idx64[p] : (int64_t)idx32[p];
      if (id >= 0 && id < n_nodes && flags[id]) key = pos[id];
    }
    if (key >= 0) {
      h = (int)(((uint32_t)key * 2654435761u) >> 20) & (SCAT_HASH - 1);
      for (;;) {      // at most SCAT_CHUNK keys in SCAT_HASH slots: an empty slot always turns up
        const int old = atomicCAS(&keys[h], -1, key);
        if (old == -1 || old == key) break;
        h = (h + 1) & (SCAT_HASH - 1);
      }
    }
  }
  __syncthreads();
  if (t < SCAT_HASH && keys[t] >= 0) vals[t] = atomicAdd(&n_rows, 1);     // number the distinct rows
  __syncthreads();
  int row = -1, rank = 0;
  if (key >= 0) {
    row = vals[h];
    rank = atomicAdd(&cnt[row], 1);
  }
  __syncthreads();
  if (t < SCAT_CHUNK) {           // exclusive prefix of the counts (at most SCAT_CHUNK short LDS reads per thread)
    int s0 = 0;
    const int nr = n_rows;
#pragma unroll 16
    for (int i = 0; i < SCAT_CHUNK; ++i) s0 += (i < t && i < nr) ? cnt[i] : 0;     // (fixed trip count: the reads pipeline)
    start[t] = s0;
    if (t == nr - 1) n_valid = s0 + cnt[t];
  }
  __syncthreads();
  if (key >= 0) {
    const int at = start[row] + rank;
    lpos[at] = t;
    lkey[at] = key;
    lrow[at] = row;
  }
  __syncthreads();
  // The sorted list in 16 nearly even pieces, cut at group boundaries: a wave owns every group that STARTS in its
  // piece and follows the last one to its end, so a destination row is added to ONCE per workgroup (a hub row flushed
  // by every wave it spans serialises at L2 again).  A wave's positions are loaded together, PER at a time.
  const int nv = n_valid;
  if (nv == 0) return;
  const int per = (nv + 15) / 16;
  auto cut = [&](int i) {       // first group start at or after list place i
    if (i >= nv) return nv;
    const int r = lrow[i], s0 = start[r];
    return s0 == i ? i : s0 + cnt[r];
  };
  const int b = __builtin_amdgcn_readfirstlane(cut(wave * per)), e = __builtin_amdgcn_readfirstlane(cut((wave + 1) * per));
  if (b >= e) return;
  const int c1 = lane + 64;
  const bool hi = c1 < D;
  float sa = 0.f, sb = 0.f;
  int cur = __builtin_amdgcn_readfirstlane(lkey[b]);
  for (int i0 = b; i0 < e; i0 += PER) {
    float va[PER], vb[PER];
    int kk[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = i0 + u < e ? i0 + u : e - 1;          // clamped, not skipped (wave-uniform)
      kk[u] = __builtin_amdgcn_readfirstlane(i0 + u < e ? lkey[i] : -1);
      const float* src = rows + (base + __builtin_amdgcn_readfirstlane(lpos[i])) * D;
      va[u] = src[lane];
      vb[u] = src[hi ? c1 : 0];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (kk[u] < 0) break;                               // wave-uniform: past this wave's groups
      if (kk[u] != cur) {
        float* d = dst + (int64_t)cur * D;
        atomicAdd(d + lane, sa);
        if (hi) atomicAdd(d + c1, sb);
        sa = sb = 0.f;
        cur = kk[u];
      }
      sa += va[u];
      sb += vb[u];
    }
  }
  float* d = dst + (int64_t)cur * D;
  atomicAdd(d + lane, sa);
  if (hi) atomicAdd(d + c1, sb);
}

__global__ __launch_bounds__(256) void k_scatter_rows(float* __restrict__ table, int64_t n_rows, int64_t width,
                                                      const int64_t* __restrict__ ids, int64_t n,
                                                      const float* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= n) return;
  const int64_t r = ids[p];
  if (r < 0 || r >= n_rows) return;
  for (int64_t c = lane; c < width; c += 64) table[r * width + c] = x[p * width + c];
}

// ---- find_uncached_out_of_batch_neighbors (modules/embedding_module.py:200-212) on the device -------------------
// np.unique(neighbors) minus the batch's own nodes minus the cached ids, ascending, then WITHOUT its first element
// (the reference drops it on the assumption that it is the padding id 0: SURVEY H8).  Three small kernels: the
// batch's nodes stamp slot[id] = epoch; every neighbour that is not stamped and not flagged sets its bit in a
// [n_nodes] bitmap; one workgroup scans the bitmap in ascending order (popcount prefix over its words), writes the
// ids past the first, clears the words it read and leaves the count.
__global__ void k_stamp_ids(const int64_t* __restrict__ ids, int64_t n, int64_t* __restrict__ slot, int64_t epoch,
                            int64_t n_nodes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = ids[i];
  if (v >= 0 && v < n_nodes) slot[v] = epoch;
}
__global__ void k_mark_uncached(const int32_t* __restrict__ nbr, int64_t n, const int64_t* __restrict__ slot, int64_t epoch,
                                const uint8_t* __restrict__ cache_flag, int64_t n_nodes, uint32_t* __restrict__ bitmap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = nbr[i];
  if (v < 0 || v >= n_nodes || slot[v] == epoch || cache_flag[v]) return;
  // test before set: padded slots all name node 0 and hubs repeat -- thousands of atomics on one word serialise at L2
  // (26 us for 6000 neighbours); a stale read only costs a redundant atomic
  const uint32_t bit = 1u << (v & 31);
  if (__hip_atomic_load(&bitmap[v >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return;
  atomicOr(&bitmap[v >> 5], bit);
}
// The ordered scan of the bitmap, two launches over up to SCAN_WGS workgroups: each thread owns SCAN_PER consecutive
// words (thread order = id order), all loaded before the first is looked at (one workgroup walking 35 words per thread
// in a loop paid a memory round trip per word: 37 us at 1.1 M nodes).  Pass 1 leaves each workgroup's population count,
// pass 2 ranks every set bit (workgroup base + wave prefix), writes the ids past the first, clears the words.
constexpr int SCAN_PER = 16, SCAN_WGS = 256;
// the workgroup counts between the two passes: one array per GPU (calls that overlap on one GPU -- different streams --
// would share it; every caller here issues its steps on one stream)
__device__ int32_t g_scan_totals[SCAN_WGS];
__device__ __forceinline__ void scan_load(const uint32_t* __restrict__ bitmap, int64_t n_words, int64_t w0, uint32_t (&w)[SCAN_PER]) {
#pragma unroll
  for (int i = 0; i < SCAN_PER; ++i) {
    const uint32_t v = bitmap[w0 + i < n_words ? w0 + i : n_words - 1];     // clamped, never skipped
    w[i] = w0 + i < n_words ? v : 0u;
  }
}
__global__ __launch_bounds__(1024) void k_scan_count(const uint32_t* __restrict__ bitmap, int64_t n_words, int64_t per_wg) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t lo = (int64_t)blockIdx.x * per_wg, hi = lo + per_wg < n_words ? lo + per_wg : n_words;
  int mine = 0;
  for (int64_t base = lo; base < hi; base += 1024 * SCAN_PER) {     // (one trip unless the table has > 134 M rows)
    uint32_t w[SCAN_PER];
    scan_load(bitmap, hi, base + (int64_t)tid * SCAN_PER, w);
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) mine += __popc(w[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if (lane == 0) wave_tot[wave] = mine;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int i = 0; i < 16; ++i) t += wave_tot[i];
    g_scan_totals[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(1024) void k_scan_write(uint32_t* __restrict__ bitmap, int64_t n_words, int64_t per_wg, int n_wg,
                                                     int64_t* __restrict__ out, int64_t cap, int32_t* __restrict__ count) {
  __shared__ int64_t wave_tot[16];
  __shared__ int64_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t lo = (int64_t)blockIdx.x * per_wg, hi = lo + per_wg < n_words ? lo + per_wg : n_words;
  int64_t before = 0, total = 0;                       // set bits of the workgroups before this one / of all
  for (int i = lane; i < n_wg; i += 64) {
    const int t = g_scan_totals[i];
    total += t;
    if (i < (int)blockIdx.x) before += t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); total += __shfl_xor(total, o, 64); }
  if (tid == 0) carry = before;
  for (int64_t base = lo; base < hi; base += 1024 * SCAN_PER) {
    uint32_t w[SCAN_PER];
    const int64_t w0 = base + (int64_t)tid * SCAN_PER;
    scan_load(bitmap, hi, w0, w);
    int64_t mine = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) mine += __popc(w[i]);
    int64_t incl = mine;                                   // inclusive prefix over the wave
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    __syncthreads();                                       // (carry of the previous trip is in place)
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int64_t wbase = carry, trip = 0;
    for (int i = 0; i < 16; ++i) { if (i < wave) wbase += wave_tot[i]; trip += wave_tot[i]; }
    int64_t k = wbase + incl - mine;                       // rank of this thread's first id
    if (mine) {
#pragma unroll
      for (int i = 0; i < SCAN_PER; ++i) {
        uint32_t bits = w[i];
        if (!bits) continue;
        bitmap[w0 + i] = 0;                                // the bitmap is clean again for the next call
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          if (k >= 1 && k - 1 < cap) out[k - 1] = (w0 + i) * 32 + b;   // [1:] drops the smallest id
          ++k;
        }
      }
    }
    __syncthreads();
    if (tid == 0) carry += trip;
  }
  if (blockIdx.x == 0 && tid == 0) *count = (int32_t)(total > 0 ? total - 1 : 0);
}

inline unsigned blocks(int64_t n, int per) { return (unsigned)ceil_div64(n, per); }
}  // namespace

ORCA_API int orca_find_uncached(const int32_t* nbr, int64_t n_nbr, const int64_t* batch_nodes, int64_t n_batch,
                                const uint8_t* cache_flag, int64_t* slot, int64_t epoch, int64_t n_nodes, uint32_t* bitmap,
                                int64_t* out, int64_t cap, int32_t* count, void* stream) {
  ORCA_REQUIRE(n_nbr >= 0 && n_batch >= 0 && n_nodes > 0 && cap >= 0 && cache_flag && slot && bitmap && count && (cap == 0 || out),
               "orca_find_uncached: bad arguments");
  ORCA_REQUIRE((n_nbr == 0 || nbr) && (n_batch == 0 || batch_nodes), "orca_find_uncached: null id array");
  hipStream_t st = (hipStream_t)stream;
  if (n_batch) k_stamp_ids<<<blocks(n_batch, 256), 256, 0, st>>>(batch_nodes, n_batch, slot, epoch, n_nodes);
  if (n_nbr) k_mark_uncached<<<blocks(n_nbr, 256), 256, 0, st>>>(nbr, n_nbr, slot, epoch, cache_flag, n_nodes, bitmap);
  {
    const int64_t n_words = ceil_div64(n_nodes, 32);
    int n_wg = (int)ceil_div64(n_words, 1024 * SCAN_PER);
    n_wg = n_wg > SCAN_WGS ? SCAN_WGS : n_wg;
    const int64_t per_wg = ceil_div64(ceil_div64(n_words, n_wg), SCAN_PER) * SCAN_PER;
    k_scan_count<<<(unsigned)n_wg, 1024, 0, st>>>(bitmap, n_words, per_wg);
    k_scan_write<<<(unsigned)n_wg, 1024, 0, st>>>(bitmap, n_words, per_wg, n_wg, out, cap, count);
  }
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

// ---- the same set without the ordered scan (the native Path C step) ----------------------------------------------
// What the step needs of find_uncached_out_of_batch_neighbors is the SET -- every uncached node is evaluated on its own
// and pushed under its own id, so the order in which they are appended to the batch changes no result -- minus its
// smallest element (the `[1:]` quirk).  The first setter of a node's bit appends the id (one atomicAdd per wave), and a
// one-workgroup pass finds the minimum, moves the last entry into its place and clears the bits the call set: three
// launches and O(candidates) work instead of a two-pass scan of the whole bitmap (37 of Path C's 143 us per batch at
// BASELINE config 3's shape, 1.14 M nodes).
namespace {
__global__ void k_stamp_ids_zero(const int64_t* __restrict__ ids, int64_t n, int64_t* __restrict__ slot, int64_t epoch,
                                 int64_t n_nodes, int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { count[0] = 0; count[1] = 0; }
  if (i >= n) return;
  const int64_t v = ids[i];
  if (v >= 0 && v < n_nodes) slot[v] = epoch;
}
__global__ void k_mark_append(const int32_t* __restrict__ nbr, int64_t n, const int64_t* __restrict__ slot, int64_t epoch,
                              const uint8_t* __restrict__ cache_flag, int64_t n_nodes, uint32_t* __restrict__ bitmap,
                              int64_t* __restrict__ out, int64_t cap, int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool take = false;
  int64_t v = -1;
  if (i < n) {
    v = nbr[i];
    if (v == 0) {
      // the padding id: every short neighbourhood names it, hundreds of times per batch -- as many atomics on one bitmap
      // word serialise at L2 (7 of this kernel's 15 us).  It is the set's smallest element whenever it is in the set, i.e.
      // the one `[1:]` drops: it only has to be SEEN (count[1]), never listed.
      if (slot[0] != epoch && !cache_flag[0]) count[1] = 1;
    } else if (v > 0 && v < n_nodes && slot[v] != epoch && !cache_flag[v]) {
      // test before set: hubs repeat -- many atomics on one word serialise at L2
      const uint32_t bit = 1u << (v & 31);
      if (!(__hip_atomic_load(&bitmap[v >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit))
        take = !(atomicOr(&bitmap[v >> 5], bit) & bit);           // the first setter owns the id
    }
  }
  const unsigned long long m = __ballot(take);
  if (!m) return;
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1, 64);
  if (take) {
    const int64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (pos < cap) out[pos] = v;
  }
}
// one workgroup: drop the smallest id of out[0 .. n) (n = *count, at most cap entries stored), clear the bitmap words of
// every listed id; more than cap ids (the caller's overflow): the whole bitmap is cleared instead.  nodes != null (the
// native Path C step): the same workgroup also lays the evaluated set behind the Q roots -- entries Q .. Q+cap-1 = the
// listed ids, then padding: id -1 for the lists that write state (every id-indexed kernel skips it), id 0 (the padding
// node) where a row is gathered; all at time *tmax (embedding_module.py:256-259) -- and raises bit 1 of *err on overflow.
__global__ __launch_bounds__(1024) void k_uncached_finish(int64_t* __restrict__ out, int64_t cap, int32_t* __restrict__ count,
                                                          uint32_t* __restrict__ bitmap, int64_t n_words, int64_t Q,
                                                          const double* __restrict__ tmax, int64_t* __restrict__ nodes,
                                                          int32_t* __restrict__ nodes32, double* __restrict__ qts,
                                                          int32_t* __restrict__ err) {
  __shared__ long long wave_min[16];
  __shared__ long long the_min;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = *count, stored = n < cap ? n : cap;
  const bool zero_seen = count[1] != 0;          // the padding id is in the set: it is the element `[1:]` drops, and it is not listed
  long long mn = INT64_MAX;
  for (int64_t i = tid; i < stored; i += 1024) {
    const long long v = out[i];
    mn = v < mn ? v : mn;
    bitmap[v >> 5] = 0;                      // (every bit of that word was set by this call)
  }
  if (n > cap)
    for (int64_t w = tid; w < n_words; w += 1024) bitmap[w] = 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const long long x = __shfl_xor(mn, o, 64); mn = x < mn ? x : mn; }
  if (lane == 0) wave_min[wave] = mn;
  __syncthreads();
  if (tid == 0) {
    long long m2 = wave_min[0];
    for (int w = 1; w < 16; ++w) m2 = wave_min[w] < m2 ? wave_min[w] : m2;
    the_min = m2;
  }
  __syncthreads();
  // (zero_seen: the count is already that of the set without its first element.  With more ids than cap the true minimum
  // may be among the ones that were not stored: the caller's overflow flag already says the step is invalid; the stored
  // list still loses one entry so that the count stays n - 1)
  const bool drop = n > 0 && !zero_seen;
  const long long last = drop ? out[stored - 1] : 0;
  const int64_t n_out = drop ? (n > cap ? n : n - 1) : n;      // (n > cap: stays above cap, which is what callers test)
  const double t = nodes ? *tmax : 0.0;
  __syncthreads();
  for (int64_t i = tid; i < (nodes ? cap : stored); i += 1024) {
    long long v = -1;
    if (i < stored) {
      v = out[i];
      if (drop && v == the_min) out[i] = v = last;             // ids are distinct: exactly one hit
    }
    if (nodes) {
      const bool ok = i < n_out;
      nodes[Q + i] = ok ? v : -1;
      nodes32[Q + i] = ok ? (int32_t)v : 0;
      qts[Q + i] = t;
    }
  }
  if (tid == 0) {
    *count = (int32_t)n_out;
    if (nodes && err && n_out > cap) atomicOr(err, 2);         // more uncached neighbours than the launch bound
  }
}
}  // namespace

// out[0 .. *count) = the ids of orca_find_uncached in NO PARTICULAR ORDER (same set, same count, same bitmap contract)
int orca_find_uncached_unordered(const int32_t* nbr, int64_t n_nbr, const int64_t* batch_nodes, int64_t n_batch,
                                 const uint8_t* cache_flag, int64_t* slot, int64_t epoch, int64_t n_nodes, uint32_t* bitmap,
                                 int64_t* out, int64_t cap, int32_t* count, void* stream) {
  ORCA_REQUIRE(n_nbr >= 0 && n_batch >= 0 && n_nodes > 0 && cap >= 1 && cache_flag && slot && bitmap && count && out,
               "orca_find_uncached_unordered: bad arguments");
  ORCA_REQUIRE((n_nbr == 0 || nbr) && (n_batch == 0 || batch_nodes), "orca_find_uncached_unordered: null id array");
  hipStream_t st = (hipStream_t)stream;
  k_stamp_ids_zero<<<blocks(n_batch > 0 ? n_batch : 1, 256), 256, 0, st>>>(batch_nodes, n_batch, slot, epoch, n_nodes, count);
  if (n_nbr) k_mark_append<<<blocks(n_nbr, 256), 256, 0, st>>>(nbr, n_nbr, slot, epoch, cache_flag, n_nodes, bitmap, out, cap, count);
  k_uncached_finish<<<1, 1024, 0, st>>>(out, cap, count, bitmap, ceil_div64(n_nodes, 32), 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

// The two halves of the call above for the native Path C step, whose first kernel has already stamped the batch ids into
// `slot` (epoch) and zeroed count[0..1]: the marking pass, and the one-workgroup finish that also lays out the evaluated set.
int orca_uncached_mark(const int32_t* nbr, int64_t n_nbr, const uint8_t* cache_flag, const int64_t* slot, int64_t epoch,
                       int64_t n_nodes, uint32_t* bitmap, int64_t* out, int64_t cap, int32_t* count, void* stream) {
  if (n_nbr == 0) return ORCA_OK;
  k_mark_append<<<blocks(n_nbr, 256), 256, 0, (hipStream_t)stream>>>(nbr, n_nbr, slot, epoch, cache_flag, n_nodes, bitmap, out, cap, count);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}
int orca_uncached_finish_nodes(int64_t* out, int64_t cap, int32_t* count, uint32_t* bitmap, int64_t n_nodes, int64_t Q,
                               const double* tmax, int64_t* nodes, int32_t* nodes32, double* qts, int32_t* err, void* stream) {
  k_uncached_finish<<<1, 1024, 0, (hipStream_t)stream>>>(out, cap, count, bitmap, ceil_div64(n_nodes, 32), Q, tmax, nodes, nodes32, qts, err);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_time_encode(const float* t, int64_t n, const float* w, const float* b, float* out, void* stream) {
  ORCA_REQUIRE(n >= 0 && (n == 0 || (t && w && b && out)), "orca_time_encode: null pointer");
  if (n == 0) return ORCA_OK;
  k_time_encode<<<blocks(n * D, 256), 256, 0, (hipStream_t)stream>>>(t, n, w, b, out);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_fill_i32(int32_t* p, int64_t n, int32_t v, void* stream) {
  ORCA_REQUIRE(n >= 0 && (n == 0 || p), "orca_fill_i32: null pointer");
  if (n == 0) return ORCA_OK;
  k_fill_i32<<<blocks(n, 256), 256, 0, (hipStream_t)stream>>>(p, n, v);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_fill_u8(uint8_t* p, int64_t n, uint8_t v, void* stream) {
  ORCA_REQUIRE(n >= 0 && (n == 0 || p), "orca_fill_u8: null pointer");
  if (n == 0) return ORCA_OK;
  k_fill_u8<<<blocks(n, 256), 256, 0, (hipStream_t)stream>>>(p, n, v);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_worklist_build(const int32_t* ids32, int64_t n32, const int64_t* ids64, int64_t n64,
                                 const uint8_t* flags, int64_t* slot, int64_t epoch, int64_t n_nodes,
                                 int32_t* worklist, int32_t* count, void* stream) {
  ORCA_REQUIRE(n32 >= 0 && n64 >= 0 && slot && worklist && count && n_nodes > 0, "orca_worklist_build: bad arguments");
  ORCA_REQUIRE((n32 == 0 || ids32) && (n64 == 0 || ids64), "orca_worklist_build: null id array");
  if (n32 + n64 == 0) {
    ORCA_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream));
    return ORCA_OK;
  }
  k_worklist_claim<<<blocks(n32 + n64, 256), 256, 0, (hipStream_t)stream>>>(ids32, n32, ids64, n64, flags, slot, epoch,
                                                                            n_nodes, count);
  k_worklist_append<<<blocks(n32 + n64, 1024), 1024, 0, (hipStream_t)stream>>>(ids32, n32, ids64, n64, flags, slot, epoch,
                                                                               n_nodes, worklist, count);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

int orca_worklist_append(const int32_t* ids32, int64_t n32, const int64_t* ids64, int64_t n64, const uint8_t* flags,
                         const int64_t* slot, int64_t epoch, int64_t n_nodes, int32_t* worklist, int32_t* count,
                         void* stream) {
  if (n32 + n64 == 0) return ORCA_OK;
  k_worklist_append<<<blocks(n32 + n64, 1024), 1024, 0, (hipStream_t)stream>>>(ids32, n32, ids64, n64, flags, slot, epoch,
                                                                               n_nodes, worklist, count);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_set_flags(uint8_t* flags, const int64_t* ids64, const int32_t* ids32, int64_t n,
                            const int32_t* n_dev, uint8_t v, int64_t n_nodes, void* stream) {
  ORCA_REQUIRE(flags && n >= 0 && n_nodes >= 0 && (n == 0 || ids64 || ids32), "orca_set_flags: bad arguments");
  if (n == 0) return ORCA_OK;
  k_set_flags<<<blocks(n, 256), 256, 0, (hipStream_t)stream>>>(flags, ids64, ids32, n, n_dev, v, n_nodes);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_segment_last(const int64_t* ids, int64_t M, int64_t* slot, int64_t epoch, int64_t n_nodes,
                               uint8_t* winner, void* stream) {
  ORCA_REQUIRE(M >= 0 && slot && n_nodes > 0 && (M == 0 || (ids && winner)), "orca_segment_last: bad arguments");
  if (M == 0) return ORCA_OK;
  k_elect<<<blocks(M, 256), 256, 0, (hipStream_t)stream>>>(ids, M, slot, epoch, n_nodes);
  k_winner<<<blocks(M, 256), 256, 0, (hipStream_t)stream>>>(ids, M, slot, epoch, n_nodes, winner);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

// ---- the entry point that allocates nothing (include/orca_hip.h: orca_segment_last_static) ---------------------
// Its election table is this file-scope device array instead of a caller-owned `slot`: every GPU the library runs
// on holds its own 8 MiB copy at its own address until the process ends.  In host code the name `scratch` is only a
// stand-in for the symbol: the address kernels get is the one hipGetSymbolAddress returned for the current device.
// The table is all zero between calls: a call elects with atomicMax(scratch[id], 1 + position), reads the winners and
// then zeroes exactly the entries it touched.  Nothing about a call lives on the host (no epoch counter), so a call
// recorded into a HIP graph replays correctly any number of times, whatever other calls ran in between.
static __device__ __align__(256) int64_t scratch[ORCA_STATIC_IDS_MAX];
namespace {
int64_t* static_slot_ptr[64];      // device address of `scratch` per GPU (written once per device)
__global__ void k_clear_slots(const int64_t* __restrict__ ids, int64_t M, int64_t* slot, int64_t n_nodes) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int64_t id = ids[p];
  if (id >= 0 && id < n_nodes) slot[id] = 0;
}
}  // namespace

ORCA_API int orca_segment_last_static(const int64_t* ids, int64_t M, int64_t n_nodes, uint8_t* winner, void* stream) {
  ORCA_REQUIRE(M >= 0 && n_nodes > 0 && (M == 0 || (ids && winner)), "orca_segment_last_static: bad arguments");
  ORCA_REQUIRE(n_nodes <= ORCA_STATIC_IDS_MAX && M <= ORCA_STATIC_IDS_MAX,
               "orca_segment_last_static: n_nodes=%lld / M=%lld exceed the static table of %lld ids "
               "(use orca_segment_last with a caller-owned slot table)", (long long)n_nodes, (long long)M,
               (long long)ORCA_STATIC_IDS_MAX);
  int dev = 0;
  ORCA_HIP_TRY(hipGetDevice(&dev));
  ORCA_REQUIRE(dev >= 0 && dev < 64, "orca_segment_last_static: device %d out of range", dev);
  int64_t* table = __atomic_load_n(&static_slot_ptr[dev], __ATOMIC_ACQUIRE);
  if (!table) {     // once per device, with that device current (two host threads may both ask: they get the same address)
    ORCA_HIP_TRY(hipGetSymbolAddress((void**)&table, HIP_SYMBOL(scratch)));
    ORCA_REQUIRE(table != nullptr, "orca_segment_last_static: no device address for the static table");
    __atomic_store_n(&static_slot_ptr[dev], table, __ATOMIC_RELEASE);
  }
  if (M == 0) return ORCA_OK;
  hipStream_t s = (hipStream_t)stream;
  k_elect<<<blocks(M, 256), 256, 0, s>>>(ids, M, table, 1, n_nodes);
  k_winner<<<blocks(M, 256), 256, 0, s>>>(ids, M, table, 1, n_nodes, winner);
  k_clear_slots<<<blocks(M, 256), 256, 0, s>>>(ids, M, table, n_nodes);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_segment_mean(const int64_t* ids, int64_t M, int64_t Dm, const float* msgs, const float* ts,
                               int64_t* slot, int64_t epoch, int64_t n_nodes, uint8_t* winner, float* out_msgs,
                               void* stream) {
  (void)ts;   // the timestamp of a segment is ts[p] at its winner position p: nothing to compute
  ORCA_REQUIRE(M >= 0 && Dm > 0 && (M == 0 || (msgs && out_msgs)), "orca_segment_mean: bad arguments");
  int rc = orca_segment_last(ids, M, slot, epoch, n_nodes, winner, stream);
  if (rc != ORCA_OK || M == 0) return rc;
  k_segment_mean<<<(unsigned)M, 128, 0, (hipStream_t)stream>>>(ids, M, Dm, msgs, winner, out_msgs);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_message_store(const int64_t* src, const int64_t* dst, const int64_t* eidx, const double* t,
                                int64_t bs, const float* memory, const float* last_update, const float* edge_feats,
                                int d_e, const float* time_w, const float* time_b, int64_t* slot, int64_t epoch,
                                int64_t n_nodes, float* messages, float* msg_ts, uint8_t* pending, void* stream) {
  ORCA_REQUIRE(bs >= 0 && d_e >= 1 && n_nodes > 0, "orca_message_store: bad arguments");
  if (bs == 0) return ORCA_OK;
  ORCA_REQUIRE(src && dst && eidx && t && memory && last_update && edge_feats && time_w && time_b && slot &&
               messages && msg_ts && pending, "orca_message_store: null pointer");
  k_elect2<<<blocks(2 * bs, 256), 256, 0, (hipStream_t)stream>>>(src, dst, bs, slot, epoch, n_nodes);
  k_message_store<<<blocks(2 * bs, 4), 256, 0, (hipStream_t)stream>>>(src, dst, eidx, t, bs, memory, last_update,
                                                                      edge_feats, d_e, time_w, time_b, slot, epoch,
                                                                      nullptr, n_nodes, messages, msg_ts, pending);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

int orca_message_store_winners(const int64_t* src, const int64_t* dst, const int64_t* eidx, const double* t, int64_t bs,
                               const float* memory, const float* last_update, const float* edge_feats, int d_e,
                               const float* time_w, const float* time_b, const uint8_t* winner, int64_t n_nodes,
                               float* messages, float* msg_ts, uint8_t* pending, void* stream) {
  if (bs == 0) return ORCA_OK;
  k_message_store<<<blocks(2 * bs, 4), 256, 0, (hipStream_t)stream>>>(src, dst, eidx, t, bs, memory, last_update,
                                                                      edge_feats, d_e, time_w, time_b, nullptr, 0,
                                                                      winner, n_nodes, messages, msg_ts, pending);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_persist_overlay(const int64_t* ids, int64_t n, const uint8_t* pending, const float* overlay,
                                  float* memory, float* last_update, const float* msg_ts, int64_t n_nodes,
                                  void* stream) {
  ORCA_REQUIRE(n >= 0 && n_nodes > 0, "orca_persist_overlay: bad sizes");
  if (n == 0) return ORCA_OK;
  ORCA_REQUIRE(ids && pending && overlay && memory && last_update && msg_ts, "orca_persist_overlay: null pointer");
  k_persist_overlay<<<blocks(n, 4), 256, 0, (hipStream_t)stream>>>(ids, n, pending, overlay, memory, last_update, msg_ts,
                                                                  n_nodes, false);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

int orca_persist_rows(const int64_t* ids, int64_t n, const uint8_t* pending, const float* rows, float* memory,
                      float* last_update, const float* msg_ts, int64_t n_nodes, void* stream) {
  if (n == 0) return ORCA_OK;
  k_persist_overlay<<<blocks(n, 4), 256, 0, (hipStream_t)stream>>>(ids, n, pending, rows, memory, last_update, msg_ts,
                                                                  n_nodes, true);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_history_push(float* emb, const float* x, const int64_t* ids, int64_t n, int64_t* slot,
                               int64_t epoch, int64_t n_nodes, void* stream) {
  ORCA_REQUIRE(n >= 0 && n_nodes > 0 && (n == 0 || (emb && x && ids && slot)), "orca_history_push: bad arguments");
  if (n == 0) return ORCA_OK;
  k_elect<<<blocks(n, 256), 256, 0, (hipStream_t)stream>>>(ids, n, slot, epoch, n_nodes);
  k_push_rows<<<blocks(n, 4), 256, 0, (hipStream_t)stream>>>(emb, x, ids, n, slot, epoch, nullptr, n_nodes);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

int orca_history_push_winners(float* emb, const float* x, const int64_t* ids, int64_t n, const uint8_t* winner,
                              int64_t n_nodes, void* stream) {
  if (n == 0) return ORCA_OK;
  k_push_rows<<<blocks(n, 4), 256, 0, (hipStream_t)stream>>>(emb, x, ids, n, nullptr, 0, winner, n_nodes);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_gather_rows(const float* table, int64_t n_rows, int64_t width, const int64_t* idx64, const int32_t* idx32,
                              int64_t n, float* out, void* stream) {
  ORCA_REQUIRE(n >= 0 && width > 0 && n_rows >= 0 && (n == 0 || (table && out && (idx64 || idx32))), "orca_gather_rows: bad arguments");
  if (n == 0) return ORCA_OK;
  k_gather_rows<<<blocks(n, 4), 256, 0, (hipStream_t)stream>>>(table, n_rows, width, idx64, idx32, n, out);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_index_positions(const int32_t* list, int64_t n, const int32_t* n_dev, int32_t* pos, int64_t n_nodes,
                                  void* stream) {
  ORCA_REQUIRE(n >= 0 && n_nodes >= 0 && (n == 0 || (list && pos)), "orca_index_positions: bad arguments");
  if (n == 0) return ORCA_OK;
  k_index_positions<<<blocks(n, 256), 256, 0, (hipStream_t)stream>>>(list, n, n_dev, pos, n_nodes);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_scatter_add_flagged(const float* rows, const int64_t* idx64, const int32_t* idx32, int64_t n,
                                      const uint8_t* flags, const int32_t* pos, int64_t n_nodes, float* dst,
                                      void* stream) {
  ORCA_REQUIRE(n >= 0 && (n == 0 || (rows && (idx64 || idx32) && flags && pos && dst)),
               "orca_scatter_add_flagged: bad arguments");
  if (n == 0) return ORCA_OK;
  // (41 k positions: 25.0 us in chunks of 128, 22.6 in chunks of 256 -- half as many adds per hub row --, 30.1 in chunks of 512)
  if (n <= 16384)
    k_scatter_add_flagged<64><<<blocks(n, 64), 1024, 0, (hipStream_t)stream>>>(rows, idx64, idx32, n, flags, pos, n_nodes, dst);
  else
    k_scatter_add_flagged<256><<<blocks(n, 256), 1024, 0, (hipStream_t)stream>>>(rows, idx64, idx32, n, flags, pos, n_nodes, dst);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_scatter_add_flagged_pair(const float* rows_a, const int64_t* idx64_a, const int32_t* idx32_a, int64_t n_a,
                                           const float* rows_b, const int64_t* idx64_b, const int32_t* idx32_b, int64_t n_b,
                                           const uint8_t* flags, const int32_t* pos, int64_t n_nodes, float* dst,
                                           void* stream) {
  ORCA_REQUIRE(n_a >= 0 && n_b >= 0 && (n_a == 0 || (rows_a && (idx64_a || idx32_a))) && (n_b == 0 || (rows_b && (idx64_b || idx32_b))) &&
               (n_a + n_b == 0 || (flags && pos && dst)), "orca_scatter_add_flagged_pair: bad arguments");
  if (n_a == 0) return orca_scatter_add_flagged(rows_b, idx64_b, idx32_b, n_b, flags, pos, n_nodes, dst, stream);
  if (n_b == 0) return orca_scatter_add_flagged(rows_a, idx64_a, idx32_a, n_a, flags, pos, n_nodes, dst, stream);
  const int64_t n = n_a + n_b;
  // one launch: the short list's workgroups ride beside the long one's
#define ORCA_SCATTER_PAIR(CH)                                                                                              \
  k_scatter_add_flagged<CH><<<blocks(n_a, CH) + blocks(n_b, CH), 1024, 0, (hipStream_t)stream>>>(                        \
      rows_a, idx64_a, idx32_a, n_a, flags, pos, n_nodes, dst, (int)blocks(n_a, CH), rows_b, idx64_b, idx32_b, n_b)
  if (n <= 16384) ORCA_SCATTER_PAIR(64);
  else ORCA_SCATTER_PAIR(256);
#undef ORCA_SCATTER_PAIR
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_scatter_rows(float* table, int64_t n_rows, int64_t width, const int64_t* ids, int64_t n, const float* x,
                               void* stream) {
  ORCA_REQUIRE(n >= 0 && width > 0 && n_rows >= 0 && (n == 0 || (table && ids && x)), "orca_scatter_rows: bad arguments");
  if (n == 0) return ORCA_OK;
  k_scatter_rows<<<blocks(n, 4), 256, 0, (hipStream_t)stream>>>(table, n_rows, width, ids, n, x);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}
