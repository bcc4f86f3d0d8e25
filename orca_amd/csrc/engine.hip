// Native driver of one TGN batch step in the embedding-reuse (Orca-P) formulation: the launch sequence of
// model/tgn_model.py:95-147 + modules/embedding_module.py:154-182 issued from C++ on one stream, so a
// step costs three library calls instead of ~25 Python-dispatched launches (a step is ~100 microseconds of GPU
// time: host dispatch, not the kernels, was the limit).  All tables and work buffers are caller-owned
// (orca_engine_desc); the handle remembers pointers, the scratch-epoch counter and the eager-overlay state.
//
// The step is cut where ranks of a multi-GPU run have to exchange rows:
//   orca_engine_begin  : GRU of the ids the previous step stored (eager overlay; or of the ids this batch reads),
//                        build [src;dst;neg] queries + sample + elect, attention layer 1     -> emb[0][lo:hi)
//   orca_engine_layer  : history push of ALL Q rows of the previous layer (global order), attention layer l
//   orca_engine_finish : persist memory update + raw-message build/store (train or eval order)
// On one rank in train order the row writes of layer / finish ride inside the attention kernels (orca_tgat_side):
// 4 launches per step.  DESIGN.md section 3 has the table.
#include <stdlib.h>
#include <new>
#include <vector>
#include "common.h"
#include "sampler.h"
#include "gru_split.h"

struct orca_engine {
  orca_engine_desc d;
  int64_t epoch;       // next unused value of the `slot` scratch table
  int64_t bs, Q;       // current batch
  int train;
  bool overlay_covers_batch;   // train order and the overlay holds every pending positive of the batch (the whole
                               // batch on this rank, or an overlay-complete state)
  bool sliced;                 // this rank embeds a proper slice of the batch (several ranks)
  bool winners_valid;          // begin() elected the last occurrences of the push / message-store id lists
  bool persist_done;           // the memory persist of the batch was fused into attention layer 1
  int pushed_layer;            // output of this layer is already in its history table (fused push), 0 = none
  bool sharded;                // owner-computes step (several ranks, train order): see orca_engine_begin
  int64_t per;                 // owner-computes step: interactions per rank of the current batch = ceil(bs / world)
  // owner-computes step: every owner's count of the batch begun last, as partial counts per workgroup of k_owner_list in
  // host memory; the event behind that kernel; the row count they gave (orca_engine_xm_rows)
  uint32_t* counts_host = nullptr;   // [counts_blocks][2][ORCA_MAX_WORLD], pinned, written by k_owner_list
  int64_t counts_cap = 0;            // workgroups it is sized for
  int64_t counts_blocks = 0;         // workgroups of the last launch
  hipEvent_t counted = nullptr;
  bool count_pending = false;
  int64_t xm_rows = 0;
  int64_t push_rows = 0;             // rows per rank of the step's history-push exchanges (orca_engine_push_rows)
  int64_t epoch_push, epoch_msg;   // owner-computes step: the election values begin() claimed on slot / slot_m
  int fuse_gru;                // 1: the eager GRU rides in the first kernel (env ORCA_FUSE_GRU=0 keeps it a launch of its own)
  int gru_subs;                // 0: 4-wave GRU groups per block of that kernel chosen per launch (gru_subs_for); env ORCA_GRU_SUBS=1|2|4 pins it
  int sampling;                // ORCA_SAMPLE_*
  uint64_t seed;               // uniform sampling: seed of the next step
  bool store_done;             // the raw-message store of the batch was fused into the last attention layer
  const int64_t* eidx;         // batch edge ids handed to begin() (may be null)
  // Eager overlay (see orca_engine_begin).  overlay_complete: every pending id outside the outstanding list has
  // overlay[id] == GRU(messages[id], memory[id]).  have_list: eager_ids[cur ^ 1 half] holds the distinct ids whose
  // messages the last train-ordered step stored and whose overlay rows are not computed yet.
  bool overlay_complete, have_list, list_written;
  int cur;                     // half of eager_ids / eager_count this step's election writes
  int64_t list_bound;          // entries the outstanding list can hold (2 * bs of the step that wrote it)
  int profile;         // > 0: every profile-th step has HIP events attached to its sampler / GRU / attention launches
  int64_t step;        // steps begun since profiling was switched on
  bool timed_step;     // this step is one of those
  // Training capture (orca_engine_train_capture): the attention calls of begin() / layer() keep what the backward reads
  // (saved[l]) and apply dropout; the row writes of the batch (memory persist, raw-message store) are left to finish(),
  // which the caller issues after the backward has read the tables.
  bool capture;
  float* cap_saved[ORCA_ENGINE_MAX_LAYERS];
  float cap_p;
  uint64_t cap_seed[ORCA_ENGINE_MAX_LAYERS];
  float* cap_gates;
  orca_slot_cache sc;          // budget-bounded history of the native Path C step (orca_engine_set_slot_cache), rows == NULL: off
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[ORCA_PROF_CLASSES];
};

thread_local orca_launch_events orca_armed_events = {nullptr, nullptr};

// the launch shape (ORCA_TGAT_WAVES of orca_tgat_inputs.q_folded) of an attention launch of n queries (include/orca_hip.h, orca_engine_desc.small_batch);
// OrcaEngine._wpq is the same rule for the Python-dispatched launches
static inline int waves_per_query(const orca_engine_desc& d, int64_t n) {
  return (d.small_batch && d.world <= 1 && n <= 1024) ? 4 : 0;
}

namespace {

// Last-occurrence election of an id list in LDS: a hash table keyed by id holds the highest position seen (LDS
// atomics: hub ids of a power-law batch repeat hundreds of times, and the same election with global atomicMax
// serialises on them at L2).  winner[p] = 1 iff p is the last position of its id.  A list longer than one table
// holds is split over `parts` workgroups by id (part g owns the ids with id % parts == g): every workgroup scans
// the whole list and elects its own ids.
constexpr int ELECT_THREADS = 1024;
constexpr int ELECT_SLOTS = 8192;          // 2 x 32 KiB of LDS
constexpr int64_t ELECT_MAX = 6144;        // ids per table (load factor <= 0.75)
constexpr int ELECT_PARTS_MAX = 16;        // lists up to 49,152 ids (at <= 3072 ids per table)

// part of an id: a 16-bit hash scaled to [0, parts) -- every part scans the whole list, so this runs once per id and
// part and must not cost an integer division
__device__ __forceinline__ int elect_part(int64_t id, int parts) { return (int)(((((uint32_t)id * 0x85EBCA6Bu) >> 16) * (uint32_t)parts) >> 16); }

// An id that finds its table full (only possible when far more distinct ids hash to one part than the sizing
// assumes) is elected by scanning the rest of the list for a later occurrence: slow, but a full table stays
// full, so every occurrence of that id takes the same route and the result stays exact.
constexpr int ELECT_UNROLL = 6;            // ids a thread has in flight (one pass covers a full table's worth)

template <typename GetId>
__device__ __forceinline__ void elect_in_lds(int32_t* keys, int32_t* vals, int64_t n, int64_t n_nodes, GetId get,
                                             uint8_t* __restrict__ winner, int part, int parts,
                                             int32_t* __restrict__ list, int32_t* __restrict__ list_count) {
  constexpr int64_t CHUNK = (int64_t)ELECT_UNROLL * ELECT_THREADS;
  // the first chunk's ids are requested before the table is cleared: their round trip hides behind the clear
  int64_t id[ELECT_UNROLL];
  auto fetch = [&](int64_t p0) {
#pragma unroll
    for (int u = 0; u < ELECT_UNROLL; ++u) {
      const int64_t p = p0 + (int64_t)u * ELECT_THREADS + threadIdx.x;
      id[u] = get(p < n ? p : n - 1);                  // clamped, not skipped: the loads stay independent
      if (p >= n) id[u] = -1;
    }
  };
  fetch(0);
  for (int i = threadIdx.x; i < ELECT_SLOTS; i += ELECT_THREADS) { keys[i] = -1; vals[i] = -1; }
  __syncthreads();
  for (int64_t p0 = 0; p0 < n; p0 += CHUNK) {
    if (p0) fetch(p0);
#pragma unroll
    for (int u = 0; u < ELECT_UNROLL; ++u) {
      const int64_t p = p0 + (int64_t)u * ELECT_THREADS + threadIdx.x, v = id[u];
      if (v < 0 || v >= n_nodes || elect_part(v, parts) != part) continue;
      uint32_t h = ((uint32_t)v * 2654435761u) >> 19;
      int probes = 0;
      for (; probes < ELECT_SLOTS; ++probes) {
        const int32_t old = atomicCAS(&keys[h], -1, (int32_t)v);
        if (old == -1 || old == (int32_t)v) break;
        h = (h + 1) & (ELECT_SLOTS - 1);
      }
      if (probes < ELECT_SLOTS) atomicMax(&vals[h], (int32_t)p);
    }
  }
  __syncthreads();
  for (int64_t p0 = 0; p0 < n; p0 += CHUNK) {
    if (n > CHUNK) fetch(p0);                          // a single chunk still holds its ids
    unsigned long long wmask[ELECT_UNROLL];            // per slot u: the lanes of this wave that hold a winner
    int total = 0;
#pragma unroll
    for (int u = 0; u < ELECT_UNROLL; ++u) {
      const int64_t p = p0 + (int64_t)u * ELECT_THREADS + threadIdx.x, v = id[u];
      const bool valid = v >= 0 && v < n_nodes;
      const bool mine = p < n && (valid ? elect_part(v, parts) == part : part == 0);   // invalid ids: cleared by part 0
      uint8_t w = 0;
      if (mine && valid) {
        uint32_t h = ((uint32_t)v * 2654435761u) >> 19;
        int probes = 0;
        while (probes < ELECT_SLOTS && keys[h] != (int32_t)v) { h = (h + 1) & (ELECT_SLOTS - 1); ++probes; }
        if (probes < ELECT_SLOTS) {
          w = vals[h] == (int32_t)p;
        } else {
          w = 1;
          for (int64_t p2 = p + 1; p2 < n; ++p2) if (get(p2) == v) { w = 0; break; }
        }
      }
      if (mine) winner[p] = w;
      wmask[u] = __ballot(mine && w != 0);
      total += __popcll(wmask[u]);
    }
    if (list && total) {   // the distinct ids of the list, in no particular order: ONE atomicAdd per wavefront and chunk
      const int ln = threadIdx.x & 63;                 // (a returning atomic is a full L2 round trip: not one per slot)
      int base = 0;
      if (ln == 0) base = atomicAdd(list_count, total);
      base = __shfl(base, 0, 64);
#pragma unroll
      for (int u = 0; u < ELECT_UNROLL; ++u) {
        if ((wmask[u] >> ln) & 1ull) list[base + __popcll(wmask[u] & ((1ull << ln) - 1ull))] = (int32_t)id[u];
        base += __popcll(wmask[u]);
      }
    }
  }
}

// Segmented election for long lists (the global lists of a multi-rank step): this workgroup takes the ELECT_SEG
// positions from p0 on, finds the last occurrence of each id among them in an LDS table (never more than ELECT_SEG
// distinct ids in ELECT_SLOTS slots: no overflow), and publishes those with atomicMax(slot[id], epoch + p).  After the
// kernel p is the last occurrence of its id in the whole list iff slot[id] == epoch + p.  An id has at most one
// contender per segment, so hub ids do not serialise at L2 the way a per-position atomicMax does.  own_world > 1:
// only the ids with id % own_world == own_rank take part.
constexpr int ELECT_SEG = ELECT_THREADS;         // one id per thread: many short workgroups beside the sampler's waves
constexpr int ELECT_SEG_SLOTS = 2048;            // table of a segment (load factor <= 0.5)
template <typename GetId>
__device__ __forceinline__ void elect_segment(int32_t* keys, int32_t* vals, int64_t p0, int64_t n, int64_t n_nodes,
                                              GetId get, int64_t* __restrict__ slot, int64_t epoch, int own_world,
                                              int own_rank) {
  const int64_t p = p0 + threadIdx.x;
  int64_t v = get(p < n ? p : n - 1);
  if (p >= n || v < 0 || v >= n_nodes || (own_world > 1 && (int)((uint32_t)v % (uint32_t)own_world) != own_rank)) v = -1;
  for (int i = threadIdx.x; i < ELECT_SEG_SLOTS; i += ELECT_THREADS) { keys[i] = -1; vals[i] = -1; }
  __syncthreads();
  uint32_t h = ((uint32_t)v * 2654435761u) >> 21;
  if (v >= 0) {
    for (;;) {
      const int32_t old = atomicCAS(&keys[h], -1, (int32_t)v);
      if (old == -1 || old == (int32_t)v) break;
      h = (h + 1) & (ELECT_SEG_SLOTS - 1);
    }
    atomicMax(&vals[h], (int)threadIdx.x);
  }
  __syncthreads();
  if (v >= 0 && vals[h] == (int)threadIdx.x)
    atomicMax(reinterpret_cast<unsigned long long*>(&slot[v]), (unsigned long long)(epoch + p));
}

// First kernel of a step.  Query block b < nbs (block index ne + b), wave w: root query i = lo + 16 b + w of [src;dst;neg] x [t;t;t]
// (tgn_model.py:99-102) is written to nodes / qts / nodes32 and, when i lies in this rank's slice [lo, hi), its K
// neighbours are sampled into row i - lo of the sampler outputs (embedding_module.py:157), with the claim pass of
// the overlay work-list in train order.
// blocks 0 .. ne_m-1        (elect only): winners of the raw-message store over [src;dst]  (tgn_model.py:174-176)
// blocks ne_m .. ne-1       (elect only): winners of the history push over all 3*bs roots (embedding_module.py:103-113, H7)
// blocks ne .. ne+nb-1: the root queries
// The elections are either whole-list LDS tables that write a winner byte per position (elect_in_lds: one rank), or
// -- seg != 0, the owner-computes step of several ranks, whose lists are `world` times longer -- one workgroup per
// ELECT_SEG positions that publishes its segment's last occurrences with atomicMax on a table in HBM
// (elect_segment); the message list then only among the ids this rank owns.
struct begin_elect {
  int ne_m, ne;                 // message-list election blocks, all election blocks
  int seg;                      // 0: elect_in_lds (winner bytes); 1: elect_segment (slot tables)
  uint8_t *win_push, *win_msg;
  int32_t *msg_list, *msg_count;
  int64_t *slot_push, *slot_msg;
  int64_t epoch_push, epoch_msg;
  int own_world, own_rank;
};
struct begin_sample {
  int mode;                     // ORCA_SAMPLE_*
  uint64_t seed;
  orca_fences fences;           // the CSR's fence index (sampler.h)
};
// The eager GRU as a role of the first kernel (blocks 0 .. n_blocks-1, ahead of every other role: theirs is the
// longest chain of the launch): GRU(message, memory) -> overlay for the ids the previous step stored.  It shares no
// data with the other roles (they touch the batch, the CSR and the other half of the list buffers), and both are
// latency-bound launches that leave most of the chip idle: side by side they cost the longer of the two.
constexpr int GRU_SUBS_MAX = ELECT_THREADS / (GRU_KW * 64);   // 4-wave GRU groups a block can hold
// Groups per block (SUBS) is a launch-time choice.  Every wave of a CU shares its four matrix pipes, so four groups in one
// block take four times as long over their chains as one group alone.  A launch whose blocks all fit on the chip at once
// (the reference's batch size: 175 groups + 40 other blocks on 256 CUs) gives every group a block of its own -- twelve of
// its sixteen waves return at once: 17.9 -> 9.6 us for the kernel, 70 -> 61 us per step at BASELINE config 2's shape.  The
// bench's batches (1,197 groups for the bound of 2,730 rows, 392 of them with rows, + 252 other blocks) take two to a
// block: 16.1 -> 13.9 us (one: 17.0 -- the sampler's blocks queue behind the GRU's; four: 16.1).
__host__ inline int gru_subs_for(int groups, int other_blocks) {
  return groups + other_blocks <= 256 ? 1 : groups / 2 + other_blocks <= 1024 ? 2 : GRU_SUBS_MAX;
}
struct begin_gru {
  int n_blocks, n_wg;
  const float* blob;
  const int32_t* ids;
  int64_t n_max;
  const int32_t* n_dev;
  const float *x, *h;
  float* out;
};
// GRU_MSG: the message width of the GRU role (472 = the 172-d datasets, 301 = the zero-feature ones; 0 = no GRU role:
// it raises the register count of the whole kernel, and the plain variant keeps its occupancy)
template <int GRU_MSG, int SUBS>
__global__ __launch_bounds__(ELECT_THREADS) void k_begin_queries(
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const int64_t* __restrict__ neg,
    const double* __restrict__ ts, int64_t bs, int64_t* __restrict__ nodes, double* __restrict__ qts,
    int32_t* __restrict__ nodes32, begin_gru gru, begin_elect el, int nbs, int64_t n_nodes,
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ c_nbr, const int32_t* __restrict__ c_eid,
    const double* __restrict__ c_ts, int64_t n_rows, int64_t lo, int64_t hi, int by_e, int K, begin_sample smp,
    int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid, float* __restrict__ o_et, float* __restrict__ o_dt,
    orca_claim claim) {
  __shared__ __align__(16) int32_t tab[2 * ELECT_SLOTS];
  int32_t *keys = tab, *vals = tab + ELECT_SLOTS;
  auto root = [&](int64_t i) {   // one load through a selected pointer: no branch around it
    const int64_t* a = i < bs ? src : (i < 2 * bs ? dst - bs : neg - 2 * bs);
    return a[i];
  };
  constexpr bool WITH_GRU = GRU_MSG != 0;
  if (WITH_GRU && (int)blockIdx.x < gru.n_blocks) {
    static_assert(sizeof(tab) >= sizeof(f32x4) * GRU_SUBS_MAX * (GRU_KW - 1) * 3 * 64, "GRU partial sums must fit the election tables");
    if ((int)threadIdx.x >= SUBS * GRU_KW * 64) return;       // (waves that leave do not count at the group's barriers)
    gru_split_workgroup<WITH_GRU ? GRU_MSG : 3 * ORCA_D + 172, 2, 1, SUBS>(gru.blob, gru.ids, gru.n_max, gru.n_dev, gru.x, gru.h, gru.out, nullptr,
                                                          nullptr, (int)blockIdx.x * SUBS, gru.n_wg,
                                                          reinterpret_cast<f32x4 (*)[1][3][64]>(tab), nullptr);
    return;
  }
  // the election workgroups (the longest dependent chains after the GRU's) take the next block indices so that
  // they are dispatched early
  const int blk = (int)blockIdx.x - gru.n_blocks;
  if (blk < el.ne) {
    const int b = blk;
    if (el.seg) {
      if (b < el.ne_m) elect_segment(keys, vals, (int64_t)b * ELECT_SEG, 2 * bs, n_nodes, root, el.slot_msg, el.epoch_msg, el.own_world, el.own_rank);
      else elect_segment(keys, vals, (int64_t)(b - el.ne_m) * ELECT_SEG, 3 * bs, n_nodes, root, el.slot_push, el.epoch_push, 1, 0);
    } else {
      if (b < el.ne_m) elect_in_lds(keys, vals, 2 * bs, n_nodes, root, el.win_msg, b, el.ne_m, el.msg_list, el.msg_count);
      else elect_in_lds(keys, vals, 3 * bs, n_nodes, root, el.win_push, b - el.ne_m, el.ne - el.ne_m, nullptr, nullptr);
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t bq = (int64_t)(blk - el.ne);
  // this rank's roots: the contiguous slice [lo, hi) of the 3*bs roots, or (by_e, the owner-computes step) the src,
  // dst and neg roots of the INTERACTIONS [lo, hi) -- local query ql = k * (hi - lo) + (e - lo), root i = k * bs + e
  if (bq >= nbs) {   // query rows of the roots outside this rank's slice: one per thread (a wave each would be 8x the
    const int64_t i = (bq - nbs) * ELECT_THREADS + threadIdx.x;          // workgroups at 8 ranks, for three stores)
    if (i >= 3 * bs) return;
    const int64_t e = i < bs ? i : (i < 2 * bs ? i - bs : i - 2 * bs);
    if (by_e ? (e >= lo && e < hi) : (i >= lo && i < hi)) return;
    nodes[i] = root(i);
    if (!by_e) nodes32[i] = (int32_t)root(i);
    qts[i] = ts[e];
    return;
  }
  const int64_t ql = bq * (ELECT_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t nq = by_e ? 3 * (hi - lo) : hi - lo;
  if (ql >= nq) return;
  int64_t i, e;
  if (by_e) {
    const int64_t k = ql / (hi - lo);
    e = lo + ql - k * (hi - lo);
    i = k * bs + e;
  } else {
    i = lo + ql;
    e = i < bs ? i : (i < 2 * bs ? i - bs : i - 2 * bs);
  }
  const int64_t v = root(i);                                            // tgn_model.py:99
  const double t = ts[e];                                               // :102
  if (lane == 0) {
    nodes[i] = v;
    nodes32[by_e ? ql : i] = (int32_t)v;      // by_e: in local query order (the source ids of attention layer 1)
    qts[i] = t;
  }
  if (smp.mode == ORCA_SAMPLE_UNIFORM)
    sample_uniform_wave(indptr, c_nbr, c_eid, c_ts, smp.fences, n_rows, v, t, ql, nq, K, smp.seed, i, o_nbr, o_eid, o_et, o_dt, nullptr, claim, lane);
  else
    sample_recent_wave(indptr, c_nbr, c_eid, c_ts, smp.fences, n_rows, v, t, ql, nq, K, o_nbr, o_eid, o_et, o_dt, nullptr, claim, lane);
}

// ---- owner-computes step (several ranks): kernels over the exchange buffers ---------------------------------
// An xm chunk is a header of XM_HEADER 32-bit words (word 0 = count) and then one record per row: [id, 3 unused | the
// ORCA_D floats of the row].  Records, not an id array followed by a row array: where record r lies does not depend on
// how many rows the chunk is sized for, so the owner packs its rows before anyone knows how many rows this step's
// all-gather will carry (the caller sizes it from the owners' counts, include/orca_hip.h).
constexpr int XM_HEADER = 64;
constexpr int XM_REC = ORCA_D + 4;
__host__ __device__ inline int64_t xm_chunk_words(int64_t rows) { return XM_HEADER + rows * XM_REC; }

// History push across ranks (embedding_module.py:103-113, H7 last occurrence wins).  A rank sends only the rows that WIN
// the push election -- the last occurrence of their id among all the batch's roots -- as (id, row) records like the xm
// chunks: a power-law batch repeats its hubs, so about half of a step's 3*bs rows are winners, and this exchange is the one
// a step cannot hide (attention layer l + 1 pulls what it delivers).
// k_pack_push: this rank's rows of a layer's output (rank-local order: row k*n_r + j is root k*bs + lo + j) -> xl_own.
// 16 lanes per row; one atomicAdd per workgroup on the chunk's count (zeroed by k_owner_list / k_xpush_records before).
__global__ __launch_bounds__(1024) void k_pack_push(const int64_t* __restrict__ nodes, int64_t bs, int64_t lo, int64_t n_r,
                                                    const float* __restrict__ rows, const int64_t* __restrict__ slot,
                                                    int64_t epoch, int64_t n_nodes, int32_t* __restrict__ chunk, int64_t cap) {
  __shared__ int wave_n[16];
  __shared__ int block_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & 15;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  int64_t id = -1;
  bool win = false;
  if (i < 3 * n_r) {
    const int64_t k = i / n_r, p = k * bs + lo + (i - k * n_r);
    id = nodes[p];
    win = id >= 0 && id < n_nodes && slot[id] == epoch + p;
  }
  const unsigned long long m = __ballot(win && sub == 0);      // one bit per winning row of this wave
  if (lane == 0) wave_n[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {      // one atomicAdd per workgroup (64 rows) on the chunk's count
    int tot = 0;
    for (int w = 0; w < 16; ++w) { const int c = wave_n[w]; wave_n[w] = tot; tot += c; }
    block_base = tot ? atomicAdd(chunk, tot) : 0;
  }
  __syncthreads();
  if (!win) return;
  const int r = block_base + wave_n[wave] + __popcll(m & ((1ull << (lane & ~15)) - 1ull));
  if (r >= cap) return;      // (cannot happen with a count that started at zero: at most 3*n_r winners)
  int32_t* rec = chunk + XM_HEADER + (int64_t)r * XM_REC;
  if (sub == 0) rec[0] = (int32_t)id;
  const f32x4* src = reinterpret_cast<const f32x4*>(rows + i * ORCA_D);
  f32x4* dst = reinterpret_cast<f32x4*>(rec + 4);
  dst[sub] = src[sub];
  if (sub + 16 < ORCA_D / 4) dst[sub + 16] = src[sub + 16];
}
// k_xpush_records: after the all-gather of chunks sized for `rows` records, hist[id] = row for every record of every
// rank (distinct ids: each has one winner); also clears the count of this rank's chunk for the next layer's pack.
__global__ __launch_bounds__(256) void k_xpush_records(const int32_t* __restrict__ xl, int world, int64_t rows, int64_t n_nodes,
                                                       float* __restrict__ hist, int32_t* __restrict__ own_count) {
  const int sub = threadIdx.x & 15;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (w == 0 && sub == 0 && own_count) *own_count = 0;
  const int64_t c = w / rows, r = w - c * rows;
  if (c >= world) return;
  const int32_t* chunk = xl + c * xm_chunk_words(rows);
  if (r >= chunk[0]) return;
  const int32_t* rec = chunk + XM_HEADER + r * XM_REC;
  const int64_t id = rec[0];
  if (id < 0 || id >= n_nodes) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(rec + 4);
  f32x4* dst = reinterpret_cast<f32x4*>(hist + id * ORCA_D);
  dst[sub] = src[sub];
  if (sub + 16 < ORCA_D / 4) dst[sub + 16] = src[sub + 16];
}

// The last occurrences of the ids this rank owns in [src;dst] (begin()'s election on slot_m) -> the lists the
// owner's kernels walk: positions (own_pos) and ids (own_ids), *count of them; one atomicAdd per workgroup.
// counts (with slot_push): what the caller sizes the step's two exchanges with, the same numbers on every rank without a
// collective to agree on them.  (a) For EVERY owner o, a bound on the distinct positives o holds in this batch.  The
// message election is run per owner (a rank elects only its own ids), but the election of the history push runs over all
// the batch's roots on every rank: position p of [src;dst] is counted if it is the last occurrence of its id among ALL
// roots, or if that last occurrence is a negative (then every positive occurrence of the id is counted: an over-count,
// rare -- negatives are uniform draws -- and on the safe side).  (b) For every rank r, the roots of its interactions that
// win the push election: the records it will send of each layer's output.  counts is HOST memory the device can write
// ([gridDim.x][2][ORCA_MAX_WORLD]): each workgroup stores its own partial counts there and the host adds them up -- no
// device-side total, no copy engine pass between this kernel and the attention launch behind it.  The grid covers the
// 3*bs roots (n = 2*bs of them positives); thread 0 also clears the count of this rank's push chunk.
__global__ __launch_bounds__(1024) void k_owner_list(const int64_t* __restrict__ nodes, int64_t n,
                                                     const int64_t* __restrict__ slot, int64_t epoch, int64_t n_nodes,
                                                     int own_world, int own_rank, int32_t* __restrict__ own_pos,
                                                     int32_t* __restrict__ own_ids, int32_t* __restrict__ count,
                                                     const int64_t* __restrict__ slot_push, int64_t epoch_push,
                                                     int64_t per, uint32_t* __restrict__ counts,
                                                     int32_t* __restrict__ push_count) {
  __shared__ int wave_n[16];
  __shared__ int block_base;
  __shared__ unsigned owner_n[2 * ORCA_MAX_WORLD];
  if (threadIdx.x < 2 * ORCA_MAX_WORLD) owner_n[threadIdx.x] = 0u;
  if (push_count && blockIdx.x == 0 && threadIdx.x == 0) *push_count = 0;
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t id = -1;
  int owner = -1, sender = -1;        // of a counted position: the owner of its id / the rank that embeds the root
  bool take = false;
  if (p < n + n / 2) {
    id = nodes[p];
    const bool ok = id >= 0 && id < n_nodes;
    // (both election words requested together, from a clamped id: behind the range test and the `counts` test they were
    // two more dependent round trips)
    const int64_t idc = ok ? id : 0;
    const int64_t s_m = slot[idc], s_p = (counts ? slot_push : slot)[idc];
    if (ok) {
      const int o = (int)((uint32_t)id % (uint32_t)own_world);
      take = p < n && o == own_rank && s_m == epoch + p;
      if (counts) {
        const int64_t w = s_p - epoch_push;                 // last occurrence of the id among [src;dst;neg]
        if (p < n && (w == p || w >= n)) owner = o;
        if (w == p) sender = (int)((p % (n / 2)) / per);
      }
    }
  }
  const unsigned long long m = __ballot(take);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (counts)
    for (int o = 0; o < own_world; ++o) {       // one LDS add per wave, owner and sender
      const int c = __popcll(__ballot(owner == o)), c2 = __popcll(__ballot(sender == o));
      if (lane == 0 && c) atomicAdd(&owner_n[o], (unsigned)c);
      if (lane == 0 && c2) atomicAdd(&owner_n[ORCA_MAX_WORLD + o], (unsigned)c2);
    }
  if (lane == 0) wave_n[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) { int c = wave_n[w]; wave_n[w] = tot; tot += c; }
    block_base = tot ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  if (counts && (int)threadIdx.x < 2 * ORCA_MAX_WORLD) counts[(int64_t)blockIdx.x * 2 * ORCA_MAX_WORLD + threadIdx.x] = owner_n[threadIdx.x];
  if (take) {
    const int k = block_base + wave_n[wave] + __popcll(m & ((1ull << lane) - 1ull));
    own_pos[k] = (int32_t)p;
    own_ids[k] = (int32_t)id;
  }
}

// Owner's share of tgn_model.py:127-141 for the winner positions p of [src;dst] whose id this rank owns: the memory
// persist (memory_updater.py:25-39: the overlay row of a pending id is GRU(message, memory) already) and then the
// raw message built from the persisted rows (tgn_model.py:172-191, memory.py:25-28).  The rows of id and partner
// are read as `pending ? overlay : memory`: for a pending positive that IS its persisted row, whoever owns it, and
// this kernel writes neither overlay nor pending (the flags of the stored ids are raised by k_apply_rows on every
// rank, after the exchange), so no wave can see another wave's half-made update.
__global__ __launch_bounds__(256) void k_owner_persist_store(
    const int64_t* __restrict__ nodes, const double* __restrict__ qts, const int64_t* __restrict__ eidx, int64_t bs,
    const int32_t* __restrict__ own_pos, const int32_t* __restrict__ own_count, int64_t n_nodes,
    const uint8_t* __restrict__ pending,
    const float* __restrict__ overlay, float* __restrict__ memory, float* __restrict__ last_update,
    const float* __restrict__ edge_feats, int d_e, const float* __restrict__ tw, const float* __restrict__ tb,
    float* __restrict__ messages, float* __restrict__ msg_ts) {
  const int lane = threadIdx.x & 63, c1 = lane + 64, cb = c1 < ORCA_D ? c1 : 0;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= *own_count) return;
  // A wave per row, and what the row costs is its dependent round trips: the loads go out level by level, both candidates
  // of every `pending ? a : b` requested and the choice made on the values (a select on the ADDRESS is one more level, a
  // load behind a branch is waited for where it stands) -- position -> ids, edge, time -> flags, both time stamps, both
  // candidate rows of id and partner, the edge row -> the stores.  (Written as the reference reads it was seven levels.)
  const int64_t p = own_pos[r];
  const int64_t e = p < bs ? p : p - bs;
  // level 2
  const int64_t id = nodes[p], partner = nodes[p < bs ? p + bs : p - bs];
  const float tf = (float)qts[p];                                    // tgn_model.py:181
  const int64_t ei = eidx[e];
  const bool pok = partner >= 0 && partner < n_nodes;
  const int64_t pc = pok ? partner : 0;
  // level 3
  const uint8_t f_id = pending[id], f_pa = pending[pc];
  const float t_msg = msg_ts[id], t_lu = last_update[id];
  const float so_a = overlay[id * ORCA_D + lane], so_b = overlay[id * ORCA_D + cb];
  const float sm_a = memory[id * ORCA_D + lane], sm_b = memory[id * ORCA_D + cb];
  const float do_a = overlay[pc * ORCA_D + lane], do_b = overlay[pc * ORCA_D + cb];
  const float dm_a = memory[pc * ORCA_D + lane], dm_b = memory[pc * ORCA_D + cb];
  const float* ef = edge_feats + ei * d_e;
  constexpr int ME = (ORCA_MAX_DE + 63) / 64;
  float ev[ME];
#pragma unroll
  for (int i = 0; i < ME; ++i) { const int c = lane + 64 * i; ev[i] = ef[c < d_e ? c : 0]; }
  const float w_a = tw[lane], w_b = tw[cb], b_a = tb[lane], b_b = tb[cb];
  const bool pend = f_id != 0, ppend = pok && f_pa != 0;
  const float lu = pend ? t_msg : t_lu;                              // memory_updater.py:37
  const float delta = tf - lu;                                       // tgn_model.py:187
  const int msg_dim = 3 * ORCA_D + d_e;
  float* row = messages + id * msg_dim;
  const float a0 = pend ? so_a : sm_a, a1 = pend ? so_b : sm_b;
  row[lane] = a0;
  row[ORCA_D + lane] = ppend ? do_a : dm_a;
  row[2 * ORCA_D + d_e + lane] = orca_cosf(orca_time_arg(delta, w_a, b_a));       // tgn_model.py:188
  if (pend) memory[id * ORCA_D + lane] = a0;                         // memory_updater.py:38
  if (c1 < ORCA_D) {
    row[c1] = a1;
    row[ORCA_D + c1] = ppend ? do_b : dm_b;
    row[2 * ORCA_D + d_e + c1] = orca_cosf(orca_time_arg(delta, w_b, b_b));
    if (pend) memory[id * ORCA_D + c1] = a1;
  }
#pragma unroll
  for (int i = 0; i < ME; ++i) { const int c = lane + 64 * i; if (c < d_e) row[2 * ORCA_D + c] = ev[i]; }
  if (lane == 0) {
    if (pend) last_update[id] = lu;
    msg_ts[id] = tf;
  }
}

// this rank's chunk (xm_own) <- the overlay rows of the ids it owns among the batch's positives
__global__ __launch_bounds__(256) void k_pack_rows(const int32_t* __restrict__ ids, const int32_t* __restrict__ count,
                                                   int64_t cap, const float* __restrict__ overlay,
                                                   int32_t* __restrict__ chunk, int32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n = *count;
  if (r == 0 && lane == 0) {
    chunk[0] = (int32_t)(n < cap ? n : cap);
    if (n > cap) atomicOr(err, 1);
  }
  if (r >= n || r >= cap) return;
  const int32_t id = ids[r];
  int32_t* rec = chunk + XM_HEADER + r * XM_REC;
  if (lane == 0) rec[0] = id;
  float* dst = reinterpret_cast<float*>(rec + 4);
  const float* srcrow = overlay + (int64_t)id * ORCA_D;
  for (int c = lane; c < ORCA_D; c += 64) dst[c] = srcrow[c];
}

// after the all-gather of chunks sized for `rows` records: overlay[id] = row, pending[id] = 1 for the ids of every chunk
// (the own chunk: flags only)
__global__ __launch_bounds__(256) void k_apply_rows(const int32_t* __restrict__ xm, int world, int rank, int64_t rows,
                                                    int64_t n_nodes, float* __restrict__ overlay,
                                                    uint8_t* __restrict__ pending, int32_t* __restrict__ own_count) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w == 0 && lane == 0) *own_count = 0;       // the list this step's election filled is spent
  const int64_t c = w / rows, r = w - c * rows;
  if (c >= world) return;
  const int32_t* chunk = xm + c * xm_chunk_words(rows);
  if (r >= chunk[0]) return;
  const int32_t* rec = chunk + XM_HEADER + r * XM_REC;
  const int64_t id = rec[0];
  if (id < 0 || id >= n_nodes) return;
  if (c != rank) {
    const float* row = reinterpret_cast<const float*>(rec + 4);
    for (int c2 = lane; c2 < ORCA_D; c2 += 64) overlay[id * ORCA_D + c2] = row[c2];
  }
  if (lane == 0) pending[id] = 1;
}

// RAII bracket: while profiling is on, arms an event pair that the next timed launch (sampler / GRU / attention)
// attaches to its own dispatch -- the pair then holds that kernel's begin and end timestamps
struct Timed {
  orca_engine* e; int cls; hipEvent_t a, b; bool on;
  Timed(orca_engine* e_, int cls_, hipStream_t) : e(e_), cls(cls_), a(nullptr), b(nullptr), on(e_->timed_step) {
    if (on && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) orca_armed_events = orca_launch_events{a, b};
    else on = false;
  }
  ~Timed() {
    if (!on) return;
    const bool consumed = orca_armed_events.start != a;
    orca_armed_events = orca_launch_events{nullptr, nullptr};
    if (consumed) { e->ev[cls].emplace_back(a, b); return; }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
};

int64_t take_epochs(orca_engine* e, int64_t n) {
  int64_t s = e->epoch;
  e->epoch += n + 1;
  return s;
}

}  // namespace

ORCA_API int orca_engine_create(const orca_engine_desc* desc, orca_engine_t** out) {
  ORCA_REQUIRE(desc && out, "orca_engine_create: null pointer");
  const orca_engine_desc& d = *desc;
  ORCA_REQUIRE(d.n_nodes > 0 && d.n_layers >= 1 && d.n_layers <= ORCA_ENGINE_MAX_LAYERS && d.K >= 1 && d.K <= ORCA_MAX_K &&
               d.d_e >= 1 && d.d_e <= ORCA_MAX_DE && d.q_max > 0, "orca_engine_create: bad sizes");
  ORCA_REQUIRE(d.edge_feats && d.time_w && d.time_b && d.gru_blob && d.memory && d.last_update && d.messages &&
               d.msg_ts && d.pending && d.overlay && d.slot && d.worklist && d.count && d.nodes && d.qts && d.nodes32 &&
               d.nbr && d.eid && d.et && d.dt && d.emb[0] && d.emb[1], "orca_engine_create: null table or work buffer");
  for (int l = 0; l < d.n_layers; ++l) ORCA_REQUIRE(d.att_blob[l], "orca_engine_create: missing attention blob %d", l);
  // (hist[l] may be NULL for a caller that keeps layer l's history in a slot cache: the steps that need the table say so)
  ORCA_REQUIRE(d.world <= 1 || (d.rank >= 0 && d.rank < d.world && d.q_max % d.world == 0),
               "orca_engine_create: bad rank %d of %d (q_max must be a multiple of world)", d.rank, d.world);
  orca_engine* e = new (std::nothrow) orca_engine();
  if (!e) { orca_set_error("orca_engine_create: out of host memory"); return ORCA_ENOMEM; }
  e->d = d;
  e->epoch = 1;
  e->bs = e->Q = 0;
  e->train = 1;
  e->overlay_covers_batch = false;
  e->winners_valid = false;
  e->persist_done = false;
  e->pushed_layer = 0;
  e->store_done = false;
  e->sharded = false;
  e->per = 0;
  e->epoch_push = e->epoch_msg = 0;
  e->sampling = ORCA_SAMPLE_RECENT;
  e->seed = 0;
  {
    const char* v = getenv("ORCA_FUSE_GRU");
    e->fuse_gru = !(v && v[0] == '0');
    const char* g = getenv("ORCA_GRU_SUBS");      // A/B switch of the profiles
    e->gru_subs = (g && (g[0] == '1' || g[0] == '2' || g[0] == '4') && !g[1]) ? g[0] - '0' : 0;
  }
  e->sc = orca_slot_cache{};
  e->eidx = nullptr;
  e->overlay_complete = false;
  e->have_list = false;
  e->list_written = false;
  e->cur = 0;
  e->list_bound = 0;
  e->profile = 0;
  e->step = 0;
  e->timed_step = false;
  *out = e;
  return ORCA_OK;
}

ORCA_API int orca_engine_profile(orca_engine_t* e, int enable) {
  ORCA_REQUIRE(e, "orca_engine_profile: null handle");
  e->profile = enable > 0 ? enable : 0;
  e->step = 0;
  e->timed_step = false;
  return ORCA_OK;
}

ORCA_API int orca_engine_profile_read(orca_engine_t* e, int cls, int64_t* launches, double* mean_ms) {
  ORCA_REQUIRE(e && cls >= 0 && cls < ORCA_PROF_CLASSES && launches && mean_ms, "orca_engine_profile_read: bad arguments");
  double total = 0.0;
  int64_t n = 0;
  for (auto& p : e->ev[cls]) {
    float ms = 0.f;
    if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
      total += ms;
      ++n;
    }
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  e->ev[cls].clear();
  *launches = n;
  *mean_ms = n ? total / (double)n : 0.0;
  return ORCA_OK;
}

ORCA_API int orca_engine_destroy(orca_engine_t* e) {
  if (e) {
    int64_t n; double ms;
    for (int c = 0; c < ORCA_PROF_CLASSES; ++c) (void)orca_engine_profile_read(e, c, &n, &ms);
    if (e->counted) { (void)hipEventSynchronize(e->counted); (void)hipEventDestroy(e->counted); }
    if (e->counts_host) (void)hipHostFree(e->counts_host);
  }
  delete e;
  return ORCA_OK;
}

ORCA_API int orca_engine_set_csr(orca_engine_t* e, const orca_csr_t* csr) {
  ORCA_REQUIRE(e, "orca_engine_set_csr: null handle");
  e->d.csr = csr;
  return ORCA_OK;
}

ORCA_API int64_t orca_engine_take_epochs(orca_engine_t* e, int64_t n) {
  if (!e || n < 0) return -1;
  return take_epochs(e, n);
}

ORCA_API int64_t orca_engine_xm_chunk_words(int64_t cap) { return cap > 0 ? xm_chunk_words(cap) : 0; }

ORCA_API int orca_engine_set_sampling(orca_engine_t* e, int mode, uint64_t seed) {
  ORCA_REQUIRE(e && (mode == ORCA_SAMPLE_RECENT || mode == ORCA_SAMPLE_UNIFORM), "orca_engine_set_sampling: bad arguments");
  e->sampling = mode;
  e->seed = seed;
  return ORCA_OK;
}

namespace {
// the save pointers and dropout fields of attention layer index l (0-based) of a captured training step
void capture_side(const orca_engine_t* e, int l, int64_t n, orca_tgat_inputs& in, orca_tgat_side& side) {
  if (e->cap_saved[l]) {
    const orca_saved_layout SL = orca_saved(e->d.d_e, n, e->d.K);
    float* sv = e->cap_saved[l];
    side.save_qt = sv + SL.qt; side.save_a = sv + SL.a; side.save_xbar = sv + SL.xbar;
    side.save_wsum = sv + SL.wsum; side.save_invalid = sv + SL.invalid;
  }
  if (e->cap_p > 0.f) {
    in.dropout_p = e->cap_p;
    in.dropout_seed = e->cap_seed[l];
    in.dropout_q0 = 0;
  }
}
}  // namespace

ORCA_API int orca_engine_train_capture(orca_engine_t* e, int on, float* const* saved, float dropout_p,
                                       const uint64_t* dropout_seeds, float* gru_gates) {
  ORCA_REQUIRE(e, "orca_engine_train_capture: null handle");
  e->capture = on != 0;
  e->cap_p = 0.f;
  e->cap_gates = nullptr;
  for (int l = 0; l < ORCA_ENGINE_MAX_LAYERS; ++l) { e->cap_saved[l] = nullptr; e->cap_seed[l] = 0; }
  if (!on) return ORCA_OK;
  ORCA_REQUIRE(e->d.world <= 1, "orca_engine_train_capture: one rank only (several ranks back-propagate slices: the "
               "dispatched path)");
  ORCA_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || dropout_seeds),
               "orca_engine_train_capture: dropout_p=%g needs seeds and must lie in [0,1)", (double)dropout_p);
  ORCA_REQUIRE(!gru_gates || e->d.d_e == 172, "orca_engine_train_capture: gru_gates needs d_e = 172");
  e->cap_p = dropout_p;
  e->cap_gates = gru_gates;
  for (int l = 0; l < e->d.n_layers; ++l) {
    if (saved) e->cap_saved[l] = saved[l];
    if (dropout_seeds) e->cap_seed[l] = dropout_seeds[l];
  }
  return ORCA_OK;
}

ORCA_API int orca_engine_begin(orca_engine_t* e, const int64_t* src, const int64_t* dst, const int64_t* neg,
                               const int64_t* eidx, const double* ts, int64_t bs, int64_t lo, int64_t hi, int train,
                               void* stream_) {
  ORCA_REQUIRE(e && e->d.csr, "orca_engine_begin: no neighbour finder set");
  const orca_engine_desc& d = e->d;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t Q = 3 * bs;
  const int world = d.world > 1 ? d.world : 1;
  const bool sharded = world > 1 && train;
  ORCA_REQUIRE(bs >= 0 && Q <= d.q_max && 0 <= lo && lo <= hi && hi <= (sharded ? bs : Q), "orca_engine_begin: batch of "
               "%lld interactions (slice %lld..%lld) does not fit q_max=%lld", (long long)bs, (long long)lo, (long long)hi,
               (long long)d.q_max);
  e->bs = bs; e->Q = Q; e->train = train;
  e->timed_step = e->profile > 0 && (e->step++ % e->profile) == 0;
  e->sliced = sharded || !(lo == 0 && hi == Q);
  // Several ranks, train order: the owner-computes step.  Every rank embeds its slice; the row writes of the batch
  // (memory persist, raw-message store, GRU of the stored ids) are done by the owner of each id, and the rows other
  // ranks read travel through the emb / xm buffers (include/orca_hip.h).
  e->sharded = sharded;
  e->per = ceil_div64(bs, world);
  if (e->sharded) {
    ORCA_REQUIRE(d.xm && d.xm_own && d.xm_cap > 0 && d.own_ids && d.own_pos && d.own_count && d.err && d.slot_m,
                 "orca_engine_begin: a train-ordered step of %d ranks needs the exchange buffers", world);
    ORCA_REQUIRE(world <= ORCA_MAX_WORLD, "orca_engine_begin: %d ranks (at most %d)", world, ORCA_MAX_WORLD);
    ORCA_REQUIRE(d.n_layers < 2 || (d.xl && d.xl_own), "orca_engine_begin: a train-ordered step of %d ranks with %d layers needs "
                 "the history-push exchange buffers (xl, xl_own)", world, d.n_layers);
    ORCA_REQUIRE(e->overlay_complete, "orca_engine_begin: the owner-computes step needs an overlay-complete state "
                 "(rebuild the overlay rows of the pending ids first)");
    const int64_t my_lo = (int64_t)d.rank * e->per < bs ? (int64_t)d.rank * e->per : bs;
    const int64_t my_hi = my_lo + e->per < bs ? my_lo + e->per : bs;
    ORCA_REQUIRE(lo == my_lo && hi == my_hi, "orca_engine_begin: interactions %lld..%lld are not rank %d's (%lld..%lld)",
                 (long long)lo, (long long)hi, d.rank, (long long)my_lo, (long long)my_hi);
    ORCA_REQUIRE(3 * e->per * world <= d.q_max, "orca_engine_begin: %d x %lld rows per rank do not fit q_max=%lld", world,
                 (long long)(3 * e->per), (long long)d.q_max);
  } else {
    ORCA_REQUIRE(!(train && e->sliced), "orca_engine_begin: a train-ordered slice needs world > 1 in the descriptor");
  }
  // the ids of a list are spread over `parts` tables of at most ELECT_MAX / 2 ids each: the workgroups run side by
  // side, and a table filled to a third keeps the probe chains -- LDS atomic round trips, paid by the whole wave --
  // short (a hub-free batch fills a table with distinct ids; 8 x 4095 roots: 34 -> 30.6 us against 6144 per table)
  auto parts_for = [](int64_t n) { return n <= ELECT_MAX / 3 ? 1 : (int)ceil_div64(n, ELECT_MAX / 2); };   // <= 3072 ids per table
  const int parts = parts_for(Q);
  e->winners_valid = !e->sharded && d.winners && parts <= ELECT_PARTS_MAX;
  e->persist_done = false;
  e->pushed_layer = 0;
  e->store_done = false;
  e->eidx = eidx;
  if (bs == 0) return ORCA_OK;
  ORCA_REQUIRE(src && dst && neg && ts, "orca_engine_begin: null batch array");
  const int64_t n = e->sharded ? 3 * (hi - lo) : hi - lo;     // root queries of this rank
  const int K = d.K;
  int rc;
  // ---- eager overlay ----------------------------------------------------------------------------------------
  // get_updated_memory (memory_updater.py:56-78) is GRU(message, memory) of the pending ids a batch reads.  Its
  // value for an id is fixed from the moment the message is stored until the id is a positive again (nothing
  // else writes its message or its memory), so it can be computed ONCE, right after the store: the election of
  // the raw-message list also emits the list of distinct ids, and the next begin() runs the GRU over exactly that
  // list (the owner-computes step: finish() does, for the ids the rank owns).  While `overlay_complete` holds no
  // batch needs a work-list of the ids it reads (no claim pass, no append kernel, no GRU over a 45,045-candidate
  // bound).  Steps that leave pending ids without an overlay row (callers that store messages outside the driver)
  // clear the flag and the driver falls back to the per-batch work-list until the state is declared clean again.
  e->cur ^= 1;
  const bool lists = !e->sharded && d.eager_ids && d.eager_count && e->winners_valid;
  int32_t* my_list = lists ? d.eager_ids + (int64_t)e->cur * d.q_max : nullptr;
  int32_t* my_count = lists ? d.eager_count + e->cur : nullptr;
  e->list_written = lists;
  begin_gru gru{};
  if (e->have_list) {
    e->have_list = false;
    if (e->overlay_complete) {
      const int32_t* ids = d.eager_ids + (int64_t)(e->cur ^ 1) * d.q_max;
      const int64_t tiles = ceil_div64(e->list_bound, 16);
      if ((d.d_e == 172 || d.d_e == 1) && e->fuse_gru && !e->timed_step && tiles <= GRU_SPLIT_TILES) {
        // a role of the first kernel (begin_gru): every tile has its GRU_NWG groups, so a group loops at most once
        gru.n_wg = (int)tiles * GRU_NWG;      // (n_blocks: at the launch, when the other roles' blocks are known)
        gru.blob = d.gru_blob; gru.ids = ids; gru.n_max = e->list_bound; gru.n_dev = d.eager_count + (e->cur ^ 1);
        gru.x = d.messages; gru.h = d.memory; gru.out = d.overlay;
      } else {
        Timed t(e, ORCA_PROF_GRU, stream);
        rc = orca_gru_update(d.gru_blob, 3 * ORCA_D + d.d_e, ids, e->list_bound, d.eager_count + (e->cur ^ 1), d.messages,
                             d.memory, d.overlay, nullptr, nullptr, stream_);
        if (rc != ORCA_OK) return rc;
      }
    }
  }
  const bool full = !e->sharded && lo == 0 && hi == Q;
  const bool eager = train && e->overlay_complete;
  const bool lazy = train && !eager;
  e->overlay_covers_batch = train && (full || eager);
  const int64_t cand = n * (K + 1);
  const int64_t claim_epoch = (lazy && n > 0) ? take_epochs(e, cand) : 0;
  {
    int64_t n_rows = 0;
    const int64_t* indptr; const int32_t *c_nbr, *c_eid; const double* c_ts;
    if ((rc = orca_csr_arrays(d.csr, &n_rows, nullptr, &indptr, &c_nbr, &c_eid, &c_ts)) != ORCA_OK) return rc;
    const orca_claim claim = (lazy && n > 0) ? orca_claim{d.pending, d.slot, claim_epoch, d.n_nodes, d.count} : orca_claim{};
    const int nbs = (int)ceil_div64(n, ELECT_THREADS / 64);                   // a wave per root of the slice
    const int nb = nbs + (n < Q ? (int)ceil_div64(Q, ELECT_THREADS) : 0);     // a thread per root outside it
    begin_elect el{};
    if (e->sharded) {   // the global lists, a workgroup per ELECT_SEG positions, published on the two slot tables
      el.seg = 1;
      el.ne_m = (int)ceil_div64(2 * bs, ELECT_SEG);
      el.ne = el.ne_m + (d.n_layers >= 2 ? (int)ceil_div64(Q, ELECT_SEG) : 0);
      el.slot_msg = d.slot_m; el.slot_push = d.slot;
      el.epoch_msg = e->epoch_msg = take_epochs(e, 2 * bs);
      el.epoch_push = e->epoch_push = take_epochs(e, Q);
      el.own_world = world; el.own_rank = d.rank;
    } else if (e->winners_valid) {
      el.ne_m = parts; el.ne = 2 * parts;
      el.win_msg = d.winners + d.q_max;
      el.win_push = d.winners;
      el.msg_list = my_list; el.msg_count = my_count;
    }
    begin_sample smp{e->sampling, e->seed, {}};
    if ((rc = orca_csr_fences(d.csr, &smp.fences)) != ORCA_OK) return rc;
    Timed t(e, ORCA_PROF_SAMPLE, stream);
    const int subs = gru.n_wg ? (e->gru_subs ? e->gru_subs : gru_subs_for(gru.n_wg, nb + el.ne)) : 1;
    gru.n_blocks = gru.n_wg ? (int)ceil_div64(gru.n_wg, subs) : 0;
#define BEGIN_LAUNCH(MSG, SUBS) \
      ORCA_LAUNCH_TIMED((k_begin_queries<MSG, SUBS>), (unsigned)(gru.n_blocks + nb + el.ne), ELECT_THREADS, 0, stream, src, dst, neg, ts, \
                        bs, d.nodes, d.qts, d.nodes32, gru, el, nbs, d.n_nodes, \
                        indptr, c_nbr, c_eid, c_ts, n_rows, lo, hi, e->sharded ? 1 : 0, K, smp, d.nbr, d.eid, d.et, d.dt, claim)
    if (!gru.n_wg) BEGIN_LAUNCH(0, 1);
    else if (d.d_e == 1) { if (subs == 1) BEGIN_LAUNCH(3 * ORCA_D + 1, 1); else if (subs == 2) BEGIN_LAUNCH(3 * ORCA_D + 1, 2); else BEGIN_LAUNCH(3 * ORCA_D + 1, 4); }
    else { if (subs == 1) BEGIN_LAUNCH(3 * ORCA_D + 172, 1); else if (subs == 2) BEGIN_LAUNCH(3 * ORCA_D + 172, 2); else BEGIN_LAUNCH(3 * ORCA_D + 172, 4); }
#undef BEGIN_LAUNCH
    ORCA_LAUNCH_CHECK();
  }
  if (e->sharded) {   // the owner's list, and every owner's count for the caller to size the xm exchange with: here rather
    // than in finish(), so that the counts reach the host while attention layer 1 runs
    const bool count = d.n_layers >= 2;      // (the counts lean on the election of the history push)
    const int64_t blocks = ceil_div64(count ? 3 * bs : 2 * bs, 1024);
    if (count && blocks > e->counts_cap) {
      if (e->counted) ORCA_HIP_TRY(hipEventSynchronize(e->counted));
      if (e->counts_host) ORCA_HIP_TRY(hipHostFree(e->counts_host));
      e->counts_host = nullptr;
      e->counts_cap = 0;
      ORCA_HIP_TRY(hipHostMalloc((void**)&e->counts_host, (size_t)blocks * 2 * ORCA_MAX_WORLD * sizeof(uint32_t), hipHostMallocDefault));
      e->counts_cap = blocks;
      if (!e->counted) ORCA_HIP_TRY(hipEventCreateWithFlags(&e->counted, hipEventDisableTiming));
    }
    uint32_t* counts_dev = nullptr;
    if (count) ORCA_HIP_TRY(hipHostGetDevicePointer((void**)&counts_dev, e->counts_host, 0));
    k_owner_list<<<(unsigned)blocks, 1024, 0, stream>>>(d.nodes, 2 * bs, d.slot_m, e->epoch_msg, d.n_nodes, world, d.rank, d.own_pos,
                                                        d.own_ids, d.own_count, d.slot, e->epoch_push, e->per, counts_dev,
                                                        count ? d.xl_own : nullptr);
    ORCA_LAUNCH_CHECK();
    e->count_pending = false;
    if (count) {
      ORCA_HIP_TRY(hipEventRecord(e->counted, stream));
      e->counts_blocks = blocks;
      e->count_pending = true;
    }
    e->xm_rows = d.xm_cap;
    e->push_rows = 3 * e->per;
  }
  // the other half of the list buffers (consumed above) is the next step's: its counter restarts at zero
  int32_t* next_count = (!e->sharded && d.eager_ids && d.eager_count) ? d.eager_count + (e->cur ^ 1) : nullptr;
  if (n == 0) {
    if (next_count) ORCA_HIP_TRY(hipMemsetAsync(next_count, 0, sizeof(int32_t), stream));
    return ORCA_OK;
  }
  orca_tgat_inputs in{};
  in.src_table = d.memory; in.src_idx = e->sharded ? d.nodes32 : d.nodes32 + lo;
  in.nbr_table = d.memory; in.nbr_idx = d.nbr; in.nbr_ids = d.nbr;
  in.edge_table = d.edge_feats; in.edge_idx = d.eid;
  in.dt = d.dt; in.time_w = d.time_w; in.time_b = d.time_b; in.q_folded = d.att_q_folded;
  if (lazy) {    // :159-161 get_updated_memory(unique(neighbors ∪ sources)) -> overlay rows
    rc = orca_worklist_append(d.nbr, n * K, d.nodes + lo, n, d.pending, d.slot, claim_epoch, d.n_nodes, d.worklist,
                              d.count, stream_);
    if (rc != ORCA_OK) return rc;
    {
      Timed t(e, ORCA_PROF_GRU, stream);
      if (e->capture && e->cap_gates)
        rc = orca_gru_update_saving(d.gru_blob, 3 * ORCA_D + d.d_e, d.worklist, cand < d.n_nodes ? cand : d.n_nodes, d.count,
                                    d.messages, d.memory, d.overlay, e->cap_gates, stream_);
      else
        rc = orca_gru_update(d.gru_blob, 3 * ORCA_D + d.d_e, d.worklist, cand < d.n_nodes ? cand : d.n_nodes, d.count,
                             d.messages, d.memory, d.overlay, nullptr, nullptr, stream_);
    }
    if (rc != ORCA_OK) return rc;
  }
  if (train) {
    in.src_overlay = d.overlay; in.src_flags = d.pending;
    in.nbr_overlay = d.overlay; in.nbr_flags = d.pending;
  }
  // a rank that embeds the whole batch lets the attention kernel do the row writes that would otherwise be
  // kernels of their own: the memory persist of the positives (train order) and the history push of its output
  orca_tgat_side side{};
  side.zero_word = next_count;
  if (e->overlay_covers_batch && full) {
    side.persist_memory = d.memory; side.persist_last_update = d.last_update; side.persist_msg_ts = d.msg_ts;
    side.persist_n = 2 * bs;
    e->persist_done = true;
  }
  ORCA_REQUIRE(d.n_layers < 2 || d.hist[0], "orca_engine_begin: no history table for layer 1");
  if (full && e->winners_valid && d.n_layers >= 2) {
    side.push_table = d.hist[0]; side.push_ids = d.nodes; side.push_winner = d.winners;
    e->pushed_layer = 1;
  }
  if (e->capture) {
    ORCA_REQUIRE(full && lazy, "orca_engine_begin: training capture needs the whole batch on this rank, train order and the "
                 "per-batch work-list (clear overlay_complete first)");
    side.persist_memory = nullptr; side.persist_last_update = nullptr; side.persist_msg_ts = nullptr; side.persist_n = 0;
    e->persist_done = false;
    capture_side(e, 0, n, in, side);
  }
  // owner-computes step: rank-major rows (3 * per rows per rank) instead of the batch's root order
  float* out = d.emb[0] + (e->sharded ? 3 * e->per * d.rank : lo) * ORCA_D;
  in.q_folded |= waves_per_query(d, n) << 8;
  {
    Timed t(e, ORCA_PROF_TGAT, stream);
    rc = orca_tgat_forward_side(d.att_blob[0], d.d_e, &in, n, K, out, nullptr, &side, stream_);   // :177 / :181
  }
  if (rc == ORCA_OK && e->sharded && d.n_layers >= 2) {      // the rows of this output that win the history push -> xl_own
    k_pack_push<<<(unsigned)ceil_div64(n, 64), 1024, 0, stream>>>(d.nodes, bs, lo, hi - lo, out, d.slot, e->epoch_push, d.n_nodes,
                                                                 d.xl_own, 3 * e->per);
    ORCA_LAUNCH_CHECK();
  }
  return rc;
}

ORCA_API int orca_engine_layer(orca_engine_t* e, int layer, int64_t lo, int64_t hi, void* stream_) {
  ORCA_REQUIRE(e, "orca_engine_layer: null handle");
  const orca_engine_desc& d = e->d;
  ORCA_REQUIRE(layer >= 2 && layer <= d.n_layers, "orca_engine_layer: layer %d outside [2,%d]", layer, d.n_layers);
  ORCA_REQUIRE(0 <= lo && lo <= hi && hi <= (e->sharded ? e->bs : e->Q), "orca_engine_layer: bad slice");
  if (e->Q == 0) return ORCA_OK;
  float* h = d.hist[layer - 2];
  ORCA_REQUIRE(h && (layer == d.n_layers || d.hist[layer - 1]), "orca_engine_layer: no history table for layer %d", layer - 1);
  orca_tgat_inputs in{};
  in.nbr_table = h; in.nbr_idx = d.nbr; in.nbr_ids = d.nbr;   // pull (history.py:32-36) fused into the gather
  in.edge_table = d.edge_feats; in.edge_idx = d.eid;
  in.dt = d.dt; in.time_w = d.time_w; in.time_b = d.time_b; in.q_folded = d.att_q_folded;
  if (e->sharded) {
    // the all-gathered records of the previous layer (every rank's push winners): into the history table, then attend
    // this rank's rows
    const float* prev = d.emb[layer & 1];
    const int64_t pr = e->push_rows;
    ORCA_REQUIRE(pr > 0 && pr <= 3 * e->per, "orca_engine_layer: %lld rows per rank in the history-push exchange",
                 (long long)pr);
    // (xl_own inside xl: a pretended rank, whose chunk is not copied -- its count is then left to the next step's begin())
    k_xpush_records<<<(unsigned)ceil_div64((int64_t)d.world * pr, 16), 256, 0, (hipStream_t)stream_>>>(
        d.xl, d.world, pr, d.n_nodes, h, d.xl_own == d.xl ? nullptr : d.xl_own);
    ORCA_LAUNCH_CHECK();
    if (hi == lo) return ORCA_OK;
    const int64_t row0 = 3 * e->per * d.rank * ORCA_D;      // this rank's rows of the rank-major buffers
    in.src_table = prev + row0;
    float* out = d.emb[(layer + 1) & 1] + row0;
    int rc;
    {
      Timed t(e, ORCA_PROF_TGAT, (hipStream_t)stream_);
      rc = orca_tgat_forward_side(d.att_blob[layer - 1], d.d_e, &in, 3 * (hi - lo), d.K, out, nullptr, nullptr, stream_);
    }
    if (rc == ORCA_OK && layer < d.n_layers) {      // this output feeds the next layer's history
      k_pack_push<<<(unsigned)ceil_div64(3 * (hi - lo), 64), 1024, 0, (hipStream_t)stream_>>>(d.nodes, e->bs, lo, hi - lo, out, d.slot,
                                                                                             e->epoch_push, d.n_nodes, d.xl_own, 3 * e->per);
      ORCA_LAUNCH_CHECK();
    }
    return rc;
  }
  const float* prev = d.emb[layer & 1];            // layer 2 reads emb[0], layer 3 reads emb[1], ...
  float* next = d.emb[(layer + 1) & 1];
  // embedding_module.py:103-113 push of every root's layer-(l-1) embedding, last occurrence wins (H7)
  int rc = ORCA_OK;
  if (e->pushed_layer != layer - 1)
    rc = e->winners_valid ? orca_history_push_winners(h, prev, d.nodes, e->Q, d.winners, d.n_nodes, stream_)
                          : orca_history_push(h, prev, d.nodes, e->Q, d.slot, take_epochs(e, e->Q), d.n_nodes, stream_);
  if (rc != ORCA_OK || hi == lo) return rc;
  in.src_table = prev + lo * ORCA_D;               // dense rows of this rank's slice
  orca_tgat_side side{};
  if (lo == 0 && hi == e->Q && e->winners_valid && layer < d.n_layers) {   // this layer's output feeds the next history
    side.push_table = d.hist[layer - 1]; side.push_ids = d.nodes; side.push_winner = d.winners;
    e->pushed_layer = layer;
  }
  // train order, whole batch on this rank, memory already persisted by layer 1: the last layer reads none of the
  // memory / message tables, so its waves also build and store the raw messages (tgn_model.py:131-141)
  if (e->capture) capture_side(e, layer - 1, hi - lo, in, side);
  if (layer == d.n_layers && lo == 0 && hi == e->Q && e->train && e->persist_done && e->winners_valid && e->eidx) {
    side.store_bs = e->bs; side.store_nodes = d.nodes; side.store_ts = d.qts; side.store_eidx = e->eidx;
    side.store_winner = d.winners + d.q_max;
    side.store_memory = d.memory; side.store_last_update = d.last_update;
    side.store_messages = d.messages; side.store_msg_ts = d.msg_ts; side.store_pending = d.pending;
    e->store_done = true;
  }
  in.q_folded |= waves_per_query(d, hi - lo) << 8;
  Timed t(e, ORCA_PROF_TGAT, (hipStream_t)stream_);
  return orca_tgat_forward_side(d.att_blob[layer - 1], d.d_e, &in, hi - lo, d.K, next + lo * ORCA_D, nullptr, &side, stream_);
}

// ---- Path C (modules/embedding_module.py:215-285, --reuse --budget B) issued from C++ without a host round trip ------
namespace {
// nodes = [src;dst;neg], qts = [t;t;t], nodes32 = the same ids; *tmax = max t (the time at which the uncached
// neighbours are evaluated, embedding_module.py:256-259).  One workgroup.
__global__ __launch_bounds__(1024) void k_budget_roots(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                       const int64_t* __restrict__ neg, const double* __restrict__ ts,
                                                       int64_t bs, int64_t* __restrict__ nodes, double* __restrict__ qts,
                                                       int32_t* __restrict__ nodes32, double* __restrict__ tmax) {
  __shared__ double red[16];
  double m = -1.0e300;
  for (int64_t i = threadIdx.x; i < 3 * bs; i += blockDim.x) {
    const int64_t b = i % bs;
    const int64_t id = i < bs ? src[b] : i < 2 * bs ? dst[b] : neg[b];
    const double t = ts[b];
    nodes[i] = id;
    nodes32[i] = (int32_t)id;
    qts[i] = t;
    m = t > m ? t : m;
  }
  for (int o = 32; o > 0; o >>= 1) { const double v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = red[w] > m ? red[w] : m;
    if (tmax) *tmax = m;
  }
}

// The CSR as the kernels below take it
struct budget_csr {
  const int64_t* indptr;
  const int32_t *nbr, *eid;
  const double* ts;
  int64_t n_rows;
  orca_fences fences;
};

// First kernel of the native Path C step, three block roles:
//  (GRU role)         : the eager overlay of the ids the previous step stored, as in k_begin_queries (blocks ahead of all others)
//  ne_m blocks        : last-occurrence election of the raw-message list [src;dst] in LDS (tgn_model.py:174-176) -> a winner
//                       byte per position, and this step's list of distinct stored ids; the first of them also reduces
//                       *tmax = max t and zeroes the two words of the uncached count
//  the others         : 16 waves = 16 root queries: root i of [src;dst;neg] x [t;t;t] (tgn_model.py:99-102) is written to
//                       nodes / qts / nodes32, stamped into `slot` (the batch-membership test of find_uncached,
//                       embedding_module.py:200-212) and its K neighbours are sampled -- into row i of the step's sampler
//                       buffers (the first 3*bs rows of the layer-1 launch) and of the caller's top-level copies (the keys
//                       of layer 2)
template <int GRU_MSG, int SUBS>
__global__ __launch_bounds__(ELECT_THREADS) void k_budget_begin(
    const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const int64_t* __restrict__ neg,
    const double* __restrict__ ts, int64_t bs, int64_t* __restrict__ nodes, double* __restrict__ qts,
    int32_t* __restrict__ nodes32, double* __restrict__ tmax, int32_t* __restrict__ unc_count, uint8_t* __restrict__ win_msg,
    int32_t* __restrict__ msg_list, int32_t* __restrict__ msg_count, begin_gru gru,
    int ne_m, int64_t n_nodes, budget_csr csr, int K, int64_t* __restrict__ stamp, int64_t stamp_epoch,
    int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid, float* __restrict__ o_et, float* __restrict__ o_dt,
    int32_t* __restrict__ t_nbr, int32_t* __restrict__ t_eid, float* __restrict__ t_et, float* __restrict__ t_dt) {
  __shared__ __align__(16) int32_t tab[2 * ELECT_SLOTS];
  auto root = [&](int64_t i) {
    const int64_t* a = i < bs ? src : (i < 2 * bs ? dst - bs : neg - 2 * bs);
    return a[i];
  };
  constexpr bool WITH_GRU = GRU_MSG != 0;
  if (WITH_GRU && (int)blockIdx.x < gru.n_blocks) {     // the eager overlay, as in k_begin_queries
    if ((int)threadIdx.x >= SUBS * GRU_KW * 64) return;
    gru_split_workgroup<WITH_GRU ? GRU_MSG : 3 * ORCA_D + 172, 2, 1, SUBS>(gru.blob, gru.ids, gru.n_max, gru.n_dev, gru.x, gru.h, gru.out, nullptr,
                                                          nullptr, (int)blockIdx.x * SUBS, gru.n_wg,
                                                          reinterpret_cast<f32x4 (*)[1][3][64]>(tab), nullptr);
    return;
  }
  const int blk0 = (int)blockIdx.x - gru.n_blocks;
  if (blk0 < ne_m) {
    if (blk0 == 0) {
      double* red = reinterpret_cast<double*>(tab);
      double m = -1.0e300;
      for (int64_t i = threadIdx.x; i < bs; i += ELECT_THREADS) { const double t = ts[i]; m = t > m ? t : m; }
      for (int o = 32; o > 0; o >>= 1) { const double v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int w = 1; w < ELECT_THREADS / 64; ++w) m = red[w] > m ? red[w] : m;
        *tmax = m;
        unc_count[0] = 0; unc_count[1] = 0;
      }
      __syncthreads();
    }
    elect_in_lds(tab, tab + ELECT_SLOTS, 2 * bs, n_nodes, root, win_msg, blk0, ne_m, msg_list, msg_count);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)(blk0 - ne_m) * (ELECT_THREADS / 64) + (threadIdx.x >> 6);
  if (i >= 3 * bs) return;
  const int64_t e = i < bs ? i : (i < 2 * bs ? i - bs : i - 2 * bs);
  const int64_t v = root(i);
  const double t = ts[e];
  if (lane == 0) {
    nodes[i] = v;
    nodes32[i] = (int32_t)v;
    qts[i] = t;
    if (v >= 0 && v < n_nodes) stamp[v] = stamp_epoch;
  }
  sample_recent_wave(csr.indptr, csr.nbr, csr.eid, csr.ts, csr.fences, csr.n_rows, v, t, i, 3 * bs, K, o_nbr, o_eid, o_et, o_dt,
                     nullptr, orca_claim{}, lane);
  if (lane < K) {    // (a lane reads back what it wrote itself)
    const int64_t o = i * K + lane;
    t_nbr[o] = o_nbr[o]; t_eid[o] = o_eid[o]; t_et[o] = o_et[o]; t_dt[o] = o_dt[o];
  }
}

// ---- budget-bounded history (slot cache) of the native Path C step ---------------------------------------------------
// The reference keeps layer-1 embeddings in a dense [n_nodes, 100] table (modules/history.py:11) of which, under
// --budget B, only the B rows of the current cache plan plus the rows pushed in the running batch are ever pulled
// (embedding_module.py:186-212).  Here they live in `rows`:
//   row 0                         zeros: what the dense table holds for an id that was never pushed, and for the padding id 0
//   rows 1 .. 2*plan_cap          two plan regions; region `cur` holds row i = the embedding of prev_plan[i], the other one is
//                                 written by this step's plan (so the copy never overwrites a row it still has to read)
//   rows 1 + 2*plan_cap ..        the batch region: the layer-1 attention launch writes its output rows straight into it --
//                                 the history push of the batch (history.py:38-40) moves no row at all
// and are found through stamp[id] = generation << 21 | kind << 20 | index (int64 per node, zero-initialised):
//   kind 0, generation g   : id was evaluated in step g at position `index` of the evaluated set; the LAST position wins
//                             because the election is an atomicMax on this word
//   kind 1, generation g   : step g's plan keeps id, as row `index` of the plan region that step wrote
// A stamp is valid for step g's pulls iff its generation is g (kind 0) or g - 1 (kind 1): nothing is ever freed, swept or
// reset, and what the plan does not keep simply goes stale.  BASELINE config 3 (1.14 M nodes, budget 1000, 600 + cap
// evaluated nodes): 3.4 MB of rows + 9.1 MB of stamps instead of 456 MB, and the pulls of layer 2 stay in L2.
constexpr int STAMP_SHIFT = 21;
constexpr int64_t STAMP_KIND = 1ll << 20, STAMP_INDEX = STAMP_KIND - 1;
struct slot_view {
  int64_t* stamp;
  int64_t gen;
  int64_t cur_base, batch_base;     // first row of the plan region that holds the cached rows / of the batch region
};
__device__ __forceinline__ int64_t slot_row(int64_t val, const slot_view& sv) {
  const int64_t g = val >> STAMP_SHIFT, idx = val & STAMP_INDEX;
  if (val & STAMP_KIND) return g == sv.gen - 1 ? sv.cur_base + idx : 0;
  return g == sv.gen ? sv.batch_base + idx : 0;
}

// Second kernel group of the step, after the evaluated set [roots; uncached; padding] is final.  Block roles, in grid order:
//  A (slot cache)    the previous plan's cache flags are cleared (history.py:25-30; nobody reads them again in this step)
//  B                 last-occurrence election of the push list (the evaluated set): slot cache -- atomicMax on the stamps;
//                    dense table -- elect_in_lds -> a winner byte per position for the attention launch's fused push
//  C (train order)   work-list claims of the roots and their neighbours (sampled by the first kernel): slot[id] = epoch + p
//  D                 16 waves = 16 uncached nodes: sample their K neighbours at *tmax into rows 3*bs.. of the sampler
//                    buffers, with their own work-list claims; waves past the device-side count return
struct expand_args {
  int nA, nB, nC;
  const int64_t* prev_plan; int64_t prev_n; uint8_t* cache_flag;
  int64_t* stamp; int64_t gen;                 // slot cache (else null)
  uint8_t* win_push; int parts;                // dense table (else null)
  orca_claim claim;                            // slot == null: no claims (evaluation order)
  const int32_t* unc_count; int64_t cap;
};
__global__ __launch_bounds__(ELECT_THREADS) void k_budget_expand(
    expand_args a, const int64_t* __restrict__ nodes, const double* __restrict__ qts, int64_t Q, int64_t n_nodes, budget_csr csr,
    int K, int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid, float* __restrict__ o_et, float* __restrict__ o_dt) {
  __shared__ __align__(16) int32_t tab[2 * ELECT_SLOTS];
  const int64_t Qc = Q + a.cap;
  int blk = (int)blockIdx.x;
  if (blk < a.nA) {
    const int64_t j = (int64_t)blk * ELECT_THREADS + threadIdx.x;
    if (j < a.prev_n) {
      const int64_t id = a.prev_plan[j];
      if (id >= 0 && id < n_nodes) a.cache_flag[id] = 0;
    }
    return;
  }
  blk -= a.nA;
  if (blk < a.nB) {
    if (a.stamp) {
      const int64_t p = (int64_t)blk * ELECT_THREADS + threadIdx.x;
      if (p < Qc) {
        const int64_t id = nodes[p];
        if (id >= 0 && id < n_nodes)
          atomicMax(reinterpret_cast<unsigned long long*>(&a.stamp[id]), (unsigned long long)((a.gen << STAMP_SHIFT) | p));
      }
    } else {
      elect_in_lds(tab, tab + ELECT_SLOTS, Qc, n_nodes, [&](int64_t p) { return nodes[p]; }, a.win_push, blk, a.parts, nullptr,
                   nullptr);
    }
    return;
  }
  blk -= a.nB;
  if (blk < a.nC) {    // positions as the append pass numbers them: [neighbours (Qc*K) ; evaluated nodes (Qc)]
    const int64_t i = (int64_t)blk * ELECT_THREADS + threadIdx.x;
    if (i == 0) *a.claim.count = 0;
    int64_t id = -1, pos = 0;
    if (i < Q * K) { id = o_nbr[i]; pos = i; }
    else if (i < Q * K + Q) { id = nodes[i - Q * K]; pos = Qc * K + (i - Q * K); }
    if (id >= 0 && id < n_nodes && a.claim.flags[id]) a.claim.slot[id] = a.claim.epoch + pos;
    return;
  }
  blk -= a.nC;
  const int lane = threadIdx.x & 63;
  const int64_t u = (int64_t)blk * (ELECT_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t n_unc = *a.unc_count < a.cap ? *a.unc_count : a.cap;
  if (u >= n_unc) return;
  const int64_t q = Q + u;
  sample_recent_wave(csr.indptr, csr.nbr, csr.eid, csr.ts, csr.fences, csr.n_rows, nodes[q], qts[q], q, Qc, K, o_nbr, o_eid, o_et,
                     o_dt, nullptr, a.claim, lane);
}

// update_flag (history.py:25-30: the flags become exactly the plan) and what it implies for the rows: entry i of the plan
// takes row i of the plan region this step writes -- copied from wherever the id's row is now (the batch region if it was
// evaluated in this step, the other plan region if the previous plan kept it, the zero row otherwise), 16 lanes a row --
// and the id is stamped (kind 1, this generation), flagged and remembered in prev_plan.
// The blocks behind the n_keep plan blocks translate the history pull of layer 2 (history.py:32-36) into row numbers -- the
// attention launch gathers rows[pull_idx[i]] exactly as it gathered hist[nbr[i]].  The two roles race on the stamps of the
// ids the plan keeps, harmlessly: such a neighbour resolves to its old row or to its copy in the region being written,
// and by the time layer 2 runs both hold the same values.
__global__ __launch_bounds__(256) void k_slot_keep(int n_keep, const int64_t* __restrict__ plan, int64_t plan_n, int64_t n_nodes,
                                                   slot_view sv, int64_t next_base, float* __restrict__ rows,
                                                   int64_t* __restrict__ prev_plan, uint8_t* __restrict__ cache_flag,
                                                   const int32_t* __restrict__ nbr_top, int64_t n_top,
                                                   int32_t* __restrict__ pull_idx) {
  if ((int)blockIdx.x >= n_keep) {
    const int64_t i = (int64_t)((int)blockIdx.x - n_keep) * 256 + threadIdx.x;
    if (i >= n_top) return;
    const int64_t v = nbr_top[i];
    int64_t r = 0;
    if (v >= 0 && v < n_nodes) {
      const int64_t val = __hip_atomic_load(&sv.stamp[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r = ((val & STAMP_KIND) && (val >> STAMP_SHIFT) == sv.gen) ? next_base + (val & STAMP_INDEX) : slot_row(val, sv);
    }
    pull_idx[i] = (int32_t)r;
    return;
  }
  const int sub = threadIdx.x & 15;
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= plan_n) return;
  const int64_t id = plan[i];
  const bool valid = id >= 0 && id < n_nodes;
  const int64_t from = valid ? slot_row(sv.stamp[id], sv) : 0;
  const f32x4* s = reinterpret_cast<const f32x4*>(rows + from * ORCA_D);
  f32x4* d = reinterpret_cast<f32x4*>(rows + (next_base + i) * ORCA_D);
  const f32x4 x0 = s[sub], x1 = s[sub + 16 < ORCA_D / 4 ? sub + 16 : sub];
  d[sub] = x0;
  if (sub + 16 < ORCA_D / 4) d[sub + 16] = x1;
  if (sub == 0) {
    prev_plan[i] = id;
    if (valid) {
      sv.stamp[id] = (sv.gen << STAMP_SHIFT) | STAMP_KIND | i;
      cache_flag[id] = 1;
    }
  }
}
// dense[prev_plan[i]] = row i of the current plan region: the dense table of the reference (History.emb) restricted to
// the ids the plan keeps, for whoever wants to look at it
__global__ __launch_bounds__(256) void k_slot_materialize(const float* __restrict__ rows, const int64_t* __restrict__ prev_plan,
                                                          int64_t prev_n, int64_t cur_base, int64_t n_nodes,
                                                          float* __restrict__ dense) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= prev_n) return;
  const int64_t id = prev_plan[i];
  if (id < 0 || id >= n_nodes) return;
  for (int c = lane; c < ORCA_D; c += 64) dense[id * ORCA_D + c] = rows[(cur_base + i) * ORCA_D + c];
}
}  // namespace

ORCA_API int orca_engine_set_slot_cache(orca_engine_t* e, const orca_slot_cache* c) {
  ORCA_REQUIRE(e, "orca_engine_set_slot_cache: null handle");
  if (!c || !c->rows) { e->sc = orca_slot_cache{}; return ORCA_OK; }
  ORCA_REQUIRE(c->stamp && c->pull_idx && c->prev_plan && c->plan_cap >= 1 && c->q_bound >= 1 && c->q_bound <= STAMP_INDEX,
               "orca_engine_set_slot_cache: incomplete cache");
  ORCA_REQUIRE(c->gen >= 2 && (c->cur == 0 || c->cur == 1) && c->prev_n >= 0 && c->prev_n <= c->plan_cap,
               "orca_engine_set_slot_cache: generation %lld / region %d / %lld cached ids out of range", (long long)c->gen, c->cur,
               (long long)c->prev_n);
  e->sc = *c;
  return ORCA_OK;
}

ORCA_API int orca_engine_get_slot_cache(orca_engine_t* e, orca_slot_cache* c) {
  ORCA_REQUIRE(e && c, "orca_engine_get_slot_cache: null pointer");
  *c = e->sc;
  return ORCA_OK;
}

ORCA_API int orca_slot_cache_materialize(const orca_slot_cache* c, int64_t n_nodes, float* dense, void* stream) {
  ORCA_REQUIRE(c && c->rows && c->prev_plan && dense && n_nodes > 0, "orca_slot_cache_materialize: null pointer");
  if (c->prev_n == 0) return ORCA_OK;
  k_slot_materialize<<<(unsigned)ceil_div64(c->prev_n, 4), 256, 0, (hipStream_t)stream>>>(
      c->rows, c->prev_plan, c->prev_n, 1 + (int64_t)c->cur * c->plan_cap, n_nodes, dense);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_build_roots(const int64_t* src, const int64_t* dst, const int64_t* neg, const double* ts, int64_t bs,
                              int64_t* nodes, double* qts, int32_t* nodes32, double* tmax, void* stream) {
  ORCA_REQUIRE(bs >= 0 && (bs == 0 || (src && dst && neg && ts && nodes && qts && nodes32)), "orca_build_roots: bad arguments");
  if (bs == 0) return ORCA_OK;
  k_budget_roots<<<1, 1024, 0, (hipStream_t)stream>>>(src, dst, neg, ts, bs, nodes, qts, nodes32, tmax);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_engine_step_budget(orca_engine_t* e, const int64_t* src, const int64_t* dst, const int64_t* neg,
                                     const double* ts, const int64_t* eidx, int64_t bs, int train,
                                     const int64_t* plan, int64_t plan_n, uint8_t* cache_flag, uint32_t* bitmap,
                                     int64_t* unc, int64_t cap, int32_t* unc_count, double* tmax,
                                     int32_t* nbr_top, int32_t* eid_top, float* et_top, float* dt_top, float* out,
                                     int32_t* err, void* stream_) {
  ORCA_REQUIRE(e, "orca_engine_step_budget: null handle");
  const orca_engine_desc& d = e->d;
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(d.n_layers == 2 && d.world <= 1, "orca_engine_step_budget: two layers on one rank (others take the dispatched path)");
  ORCA_REQUIRE(bs >= 0 && cap >= 1 && 3 * bs + cap <= d.q_max, "orca_engine_step_budget: 3*bs + cap = %lld exceeds q_max = %lld",
               (long long)(3 * bs + cap), (long long)d.q_max);
  if (bs == 0) return ORCA_OK;
  ORCA_REQUIRE(src && dst && neg && ts && eidx && cache_flag && bitmap && unc && unc_count && tmax && nbr_top && eid_top &&
               et_top && dt_top && out && d.csr && d.winners, "orca_engine_step_budget: null pointer");
  orca_slot_cache& c = e->sc;
  const bool slots = c.rows != nullptr;
  ORCA_REQUIRE(slots || d.hist[0], "orca_engine_step_budget: neither a history table nor a slot cache");
  ORCA_REQUIRE(!slots || plan_n >= 0, "orca_engine_step_budget: the slot cache follows a cache plan; a step without one (flags "
               "accumulate, history.py:25-30) needs the dense table -- materialise the slots and switch the cache off first");
  ORCA_REQUIRE(!slots || (plan_n <= c.plan_cap && 3 * bs + cap <= c.q_bound),
               "orca_engine_step_budget: slot cache for plans of %lld ids / %lld evaluated nodes is too small for %lld / %lld",
               (long long)c.plan_cap, (long long)c.q_bound, (long long)plan_n, (long long)(3 * bs + cap));
  const int64_t Q = 3 * bs, Qc = Q + cap;
  const int K = d.K;
  auto parts_for = [](int64_t n) { return n <= ELECT_MAX / 3 ? 1 : (int)ceil_div64(n, ELECT_MAX / 2); };
  const int parts_m = parts_for(2 * bs), parts_p = parts_for(Qc);
  ORCA_REQUIRE(parts_m <= ELECT_PARTS_MAX && (slots || parts_p <= ELECT_PARTS_MAX),
               "orca_engine_step_budget: a batch of %lld interactions / %lld evaluated nodes exceeds the in-kernel elections "
               "(the dispatched path takes it)", (long long)bs, (long long)Qc);
  uint8_t *win_push = d.winners, *win_msg = d.winners + d.q_max;
#define STEP(expr) do { const int rc_ = (expr); if (rc_ != ORCA_OK) return rc_; } while (0)
  (void)hipGetLastError();      // (a status left behind by an earlier runtime call is not this step's)
  budget_csr csr{};
  STEP(orca_csr_arrays(d.csr, &csr.n_rows, nullptr, &csr.indptr, &csr.nbr, &csr.eid, &csr.ts));
  STEP(orca_csr_fences(d.csr, &csr.fences));
  slot_view sv{};
  int64_t next_base = 0;
  float* comb = d.emb[0];
  if (slots) {
    sv.stamp = c.stamp; sv.gen = c.gen;
    sv.cur_base = 1 + (int64_t)c.cur * c.plan_cap;
    sv.batch_base = 1 + 2 * c.plan_cap;
    next_base = 1 + (int64_t)(c.cur ^ 1) * c.plan_cap;
    comb = c.rows + sv.batch_base * ORCA_D;        // layer 1 writes the batch region: the push moves nothing
  }
  // Eager overlay, as in orca_engine_begin: while `overlay_complete` holds, GRU(message, memory) of a pending id is
  // computed once, right after its message is stored -- by the next step's first kernel, over the list this step's
  // message election emits -- and no batch needs a work-list of the ids it reads.
  e->cur ^= 1;
  const bool lists = d.eager_ids && d.eager_count;
  int32_t* my_list = lists ? d.eager_ids + (int64_t)e->cur * d.q_max : nullptr;
  int32_t* my_count = lists ? d.eager_count + e->cur : nullptr;
  int32_t* next_count = lists ? d.eager_count + (e->cur ^ 1) : nullptr;     // consumed below: restarts at zero for the next step
  begin_gru gru{};
  if (e->have_list) {
    e->have_list = false;
    if (e->overlay_complete) {
      const int32_t* ids = d.eager_ids + (int64_t)(e->cur ^ 1) * d.q_max;
      const int64_t tiles = ceil_div64(e->list_bound, 16);
      if ((d.d_e == 172 || d.d_e == 1) && e->fuse_gru && tiles <= GRU_SPLIT_TILES) {
        gru.n_wg = (int)tiles * GRU_NWG;
        gru.blob = d.gru_blob; gru.ids = ids; gru.n_max = e->list_bound; gru.n_dev = d.eager_count + (e->cur ^ 1);
        gru.x = d.messages; gru.h = d.memory; gru.out = d.overlay;
      } else {
        STEP(orca_gru_update(d.gru_blob, 3 * ORCA_D + d.d_e, ids, e->list_bound, d.eager_count + (e->cur ^ 1), d.messages,
                             d.memory, d.overlay, nullptr, nullptr, stream_));
      }
    }
  }
  const bool lazy = train && !e->overlay_complete;      // the per-batch work-list of the ids layer 1 reads
  // 1. roots, their neighbours (the layer-2 keys), batch stamps; the message election rides along (tgn_model.py:99-102,174-176)
  const int64_t stamp_epoch = take_epochs(e, 1);
  {
    const int other = parts_m + (int)ceil_div64(Q, ELECT_THREADS / 64);
    const int subs = gru.n_wg ? (e->gru_subs ? e->gru_subs : gru_subs_for(gru.n_wg, other)) : 1;
    gru.n_blocks = gru.n_wg ? (int)ceil_div64(gru.n_wg, subs) : 0;
    const unsigned grid = (unsigned)(gru.n_blocks + other);
#define BEGIN_LAUNCH(MSG, SUBS) k_budget_begin<MSG, SUBS><<<grid, ELECT_THREADS, 0, s>>>( \
        src, dst, neg, ts, bs, d.nodes, d.qts, d.nodes32, tmax, unc_count, win_msg, my_list, my_count, gru, parts_m, d.n_nodes, csr, K, \
        d.slot, stamp_epoch, d.nbr, d.eid, d.et, d.dt, nbr_top, eid_top, et_top, dt_top)
    if (!gru.n_wg) BEGIN_LAUNCH(0, 1);
    else if (d.d_e == 1) { if (subs == 1) BEGIN_LAUNCH(3 * ORCA_D + 1, 1); else if (subs == 2) BEGIN_LAUNCH(3 * ORCA_D + 1, 2); else BEGIN_LAUNCH(3 * ORCA_D + 1, 4); }
    else { if (subs == 1) BEGIN_LAUNCH(3 * ORCA_D + 172, 1); else if (subs == 2) BEGIN_LAUNCH(3 * ORCA_D + 172, 2); else BEGIN_LAUNCH(3 * ORCA_D + 172, 4); }
#undef BEGIN_LAUNCH
    ORCA_LAUNCH_CHECK();
  }
  // 2. which of those neighbours have no cached layer-1 embedding (:200-212, :251), and the evaluated set behind the roots
  STEP(orca_uncached_mark(nbr_top, Q * K, cache_flag, d.slot, stamp_epoch, d.n_nodes, bitmap, unc, cap, unc_count, stream_));
  STEP(orca_uncached_finish_nodes(unc, cap, unc_count, bitmap, d.n_nodes, Q, tmax, d.nodes, d.nodes32, d.qts, err, stream_));
  // 3. their neighbours; the push election; the work-list claims of everything layer 1 reads (:224-246)
  expand_args xa{};
  if (slots) {
    xa.nA = (int)ceil_div64(c.prev_n, ELECT_THREADS);
    xa.prev_plan = c.prev_plan; xa.prev_n = c.prev_n; xa.cache_flag = cache_flag;
    xa.nB = (int)ceil_div64(Qc, ELECT_THREADS);
    xa.stamp = c.stamp; xa.gen = c.gen;
  } else {
    xa.nB = parts_p; xa.parts = parts_p; xa.win_push = win_push;
  }
  const int64_t cand = Qc * (K + 1);
  int64_t claim_epoch = 0;
  if (lazy) {
    claim_epoch = take_epochs(e, cand);
    xa.nC = (int)ceil_div64(Q * K + Q, ELECT_THREADS);
    xa.claim = orca_claim{d.pending, d.slot, claim_epoch, d.n_nodes, d.count};
  }
  xa.unc_count = unc_count; xa.cap = cap;
  k_budget_expand<<<(unsigned)(xa.nA + xa.nB + xa.nC + ceil_div64(cap, ELECT_THREADS / 64)), ELECT_THREADS, 0, s>>>(
      xa, d.nodes, d.qts, Q, d.n_nodes, csr, K, d.nbr, d.eid, d.et, d.dt);
  ORCA_LAUNCH_CHECK();
  orca_tgat_inputs in{};
  in.src_table = d.memory; in.src_idx = d.nodes32;
  in.nbr_table = d.memory; in.nbr_idx = d.nbr; in.nbr_ids = d.nbr;
  in.edge_table = d.edge_feats; in.edge_idx = d.eid;
  in.dt = d.dt; in.time_w = d.time_w; in.time_b = d.time_b; in.q_folded = d.att_q_folded;
  // 4. (no eager overlay) the work-list itself and GRU(message, memory) of what layer 1 reads
  if (lazy) {
    STEP(orca_worklist_append(d.nbr, Qc * K, d.nodes, Qc, d.pending, d.slot, claim_epoch, d.n_nodes, d.worklist, d.count, stream_));
    STEP(orca_gru_update(d.gru_blob, 300 + d.d_e, d.worklist, cand < d.n_nodes ? cand : d.n_nodes, d.count, d.messages, d.memory,
                         d.overlay, nullptr, nullptr, stream_));
  }
  if (train) { in.src_overlay = d.overlay; in.src_flags = d.pending; in.nbr_overlay = d.overlay; in.nbr_flags = d.pending; }
  // 5. layer 1 over [roots; uncached].  Train order: the overlay holds GRU(message, memory) of every pending positive (the
  // roots are part of the work-list), so the launch also commits those rows (memory_updater.py:25-39; the flags need no
  // clearing: the message store raises them for exactly these ids).  Dense table: it also pushes the winners' rows.
  {
    orca_tgat_side side{};
    side.q_extra = unc_count;
    side.q_base = Q;
    side.zero_word = next_count;
    if (train) {
      side.persist_memory = d.memory; side.persist_last_update = d.last_update; side.persist_msg_ts = d.msg_ts;
      side.persist_n = 2 * bs;
    }
    if (!slots) { side.push_table = d.hist[0]; side.push_ids = d.nodes; side.push_winner = win_push; }
    // (the launch is sized for the bound Qc; what it evaluates is the roots and the uncached few: the shape follows the
    // roots plus a nominal 256 -- OrcaEngine._embed_budget applies the same rule to the dispatched launch)
    in.q_folded |= waves_per_query(d, Q + 256) << 8;
    STEP(orca_tgat_forward_side(d.att_blob[0], d.d_e, &in, Qc, K, comb, nullptr, &side, stream_));
  }
  // 6. the cache flags move on (history.py:25-30)
  if (slots) {
    const int n_keep = (int)ceil_div64(plan_n, 16);
    k_slot_keep<<<(unsigned)(n_keep + ceil_div64(Q * K, 256)), 256, 0, s>>>(n_keep, plan, plan_n, d.n_nodes, sv, next_base, c.rows, c.prev_plan,
                                                                            cache_flag, nbr_top, Q * K, c.pull_idx);
    ORCA_LAUNCH_CHECK();
  } else if (plan_n >= 0) {
    STEP(orca_fill_u8(cache_flag, d.n_nodes, 0, stream_));
    if (plan_n > 0) STEP(orca_set_flags(cache_flag, plan, nullptr, plan_n, nullptr, 1, d.n_nodes, stream_));
  } else {
    STEP(orca_set_flags(cache_flag, d.nodes, nullptr, Qc, nullptr, 1, d.n_nodes, stream_));
  }
  // 7. layer 2 over the roots: their own layer-1 rows, neighbours pulled from the history (:273-285).  Train order: it reads
  // none of the memory / message / pending tables, so its waves also build and store the raw messages (tgn_model.py:131-141)
  orca_tgat_inputs in2{};
  in2.src_table = comb;
  in2.nbr_table = slots ? c.rows : d.hist[0];
  in2.nbr_idx = slots ? c.pull_idx : nbr_top;
  in2.nbr_ids = nbr_top;
  in2.edge_table = d.edge_feats; in2.edge_idx = eid_top;
  in2.dt = dt_top; in2.time_w = d.time_w; in2.time_b = d.time_b; in2.q_folded = d.att_q_folded;
  {
    orca_tgat_side side{};
    if (train) {
      side.store_bs = bs; side.store_nodes = d.nodes; side.store_ts = d.qts; side.store_eidx = eidx;
      side.store_winner = win_msg;
      side.store_memory = d.memory; side.store_last_update = d.last_update;
      side.store_messages = d.messages; side.store_msg_ts = d.msg_ts; side.store_pending = d.pending;
    }
    in2.q_folded |= waves_per_query(d, Q) << 8;
    STEP(orca_tgat_forward_side(d.att_blob[1], d.d_e, &in2, Q, K, out, nullptr, &side, stream_));
  }
  if (!train) {
    // Evaluation order (tgn_model.py:131-145): store, then GRU of the messages just stored, out of place into the overlay
    // (the split kernels; in place a row would have to be read and written by one workgroup), then the commit.  The
    // overlay rows of ids that stop being pending are dead.
    STEP(orca_message_store_winners(src, dst, eidx, ts, bs, d.memory, d.last_update, d.edge_feats, d.d_e, d.time_w, d.time_b,
                                    win_msg, d.n_nodes, d.messages, d.msg_ts, d.pending, stream_));
    STEP(orca_worklist_build(nullptr, 0, d.nodes, 2 * bs, d.pending, d.slot, take_epochs(e, 2 * bs), d.n_nodes, d.worklist,
                             d.count, stream_));
    const int64_t bound = 2 * bs < d.n_nodes ? 2 * bs : d.n_nodes;
    STEP(orca_gru_update(d.gru_blob, 300 + d.d_e, d.worklist, bound, d.count, d.messages, d.memory, d.overlay, nullptr,
                         nullptr, stream_));
    STEP(orca_commit_rows(d.worklist, bound, d.count, d.overlay, d.memory, d.last_update, d.msg_ts, d.n_nodes, stream_));
    STEP(orca_set_flags(d.pending, d.nodes, nullptr, 2 * bs, nullptr, 0, d.n_nodes, stream_));
  }
#undef STEP
  if (slots) {      // the plan region just written holds the cached rows of the next step
    c.gen += 1;
    c.cur ^= 1;
    c.prev_n = plan_n;
  }
  if (train) {      // the ids that now hold a fresh message: the election listed them for the next step's first kernel
    if (lists) {
      e->have_list = true;
      e->list_bound = 2 * bs < d.q_max ? 2 * bs : d.q_max;
    } else {
      e->overlay_complete = false;
    }
  }
  return ORCA_OK;
}

ORCA_API int orca_engine_finish(orca_engine_t* e, const int64_t* src, const int64_t* dst, const int64_t* eidx,
                                const double* ts, void* stream_) {
  ORCA_REQUIRE(e, "orca_engine_finish: null handle");
  const orca_engine_desc& d = e->d;
  const int64_t bs = e->bs;
  if (bs == 0) return ORCA_OK;
  ORCA_REQUIRE(src && dst && eidx && ts, "orca_engine_finish: null batch array");
  const int msg_dim = 3 * ORCA_D + d.d_e;
  if (e->sharded) {   // the owner's share of tgn_model.py:127-141, then GRU(message, memory) of the ids just stored
    hipStream_t st = (hipStream_t)stream_;
    int64_t bound = 2 * bs < d.q_max ? 2 * bs : d.q_max;     // rows the owner's kernels are launched for: more owned
    if (bound > d.xm_cap) bound = d.xm_cap;                   // ids than xm_cap is the overflow k_pack_rows reports
    k_owner_persist_store<<<(unsigned)ceil_div64(bound, 4), 256, 0, st>>>(
        d.nodes, d.qts, eidx, bs, d.own_pos, d.own_count, d.n_nodes, d.pending, d.overlay, d.memory, d.last_update, d.edge_feats,
        d.d_e, d.time_w, d.time_b, d.messages, d.msg_ts);
    ORCA_LAUNCH_CHECK();
    int rc;
    {
      Timed t(e, ORCA_PROF_GRU, st);
      rc = orca_gru_update(d.gru_blob, msg_dim, d.own_ids, bound, d.own_count, d.messages, d.memory, d.overlay, nullptr,
                           nullptr, stream_);
    }
    if (rc != ORCA_OK) return rc;
    k_pack_rows<<<(unsigned)ceil_div64(bound > 0 ? bound : 1, 4), 256, 0, st>>>(d.own_ids, d.own_count, d.xm_cap, d.overlay,
                                                                              d.xm_own, d.err);
    ORCA_LAUNCH_CHECK();
    return ORCA_OK;
  }
  auto persist = [&]() -> int {   // memory_updater.update_memory (:25-39) + clear_messages (memory.py:49-50)
    // When this rank's begin() covered the whole batch in train order, the overlay already holds
    // GRU(message, memory) of every pending positive (same messages, same memory: positives are roots of the
    // batch): the rows are committed by one kernel, and the flags need no clearing because the message store
    // that follows sets them for exactly these ids.
    if (e->persist_done) return ORCA_OK;   // attention layer 1 committed the rows (orca_tgat_side)
    if (e->overlay_covers_batch)
      return orca_persist_overlay(d.nodes, 2 * bs, d.pending, d.overlay, d.memory, d.last_update, d.msg_ts, d.n_nodes,
                                  stream_);
    const int64_t bound = 2 * bs < d.n_nodes ? 2 * bs : d.n_nodes;
    int rc = orca_worklist_build(nullptr, 0, d.nodes, 2 * bs, d.pending, d.slot, take_epochs(e, 2 * bs), d.n_nodes,
                                 d.worklist, d.count, stream_);
    if (rc != ORCA_OK) return rc;
    {   // GRU(message, memory) of the pending positives goes to the overlay table, then the rows are committed
      Timed t(e, ORCA_PROF_GRU, (hipStream_t)stream_);
      rc = orca_gru_update(d.gru_blob, msg_dim, d.worklist, bound, d.count, d.messages, d.memory, d.overlay, nullptr,
                           nullptr, stream_);
      if (rc != ORCA_OK) return rc;
    }
    rc = orca_commit_rows(d.worklist, bound, d.count, d.overlay, d.memory, d.last_update, d.msg_ts, d.n_nodes, stream_);
    if (rc != ORCA_OK) return rc;
    return orca_set_flags(d.pending, d.nodes, nullptr, 2 * bs, nullptr, 0, d.n_nodes, stream_);
  };
  auto store = [&]() -> int {     // tgn_model.py:131-141,172-191 + memory.py:25-28
    if (e->store_done) return ORCA_OK;   // the last attention layer stored the messages (orca_tgat_side)
    if (e->winners_valid)
      return orca_message_store_winners(src, dst, eidx, ts, bs, d.memory, d.last_update, d.edge_feats, d.d_e, d.time_w,
                                        d.time_b, d.winners + d.q_max, d.n_nodes, d.messages, d.msg_ts, d.pending, stream_);
    return orca_message_store(src, dst, eidx, ts, bs, d.memory, d.last_update, d.edge_feats, d.d_e, d.time_w, d.time_b,
                              d.slot, take_epochs(e, 2 * bs), d.n_nodes, d.messages, d.msg_ts, d.pending, stream_);
  };
  int rc;
  if (e->train) {                 // tgn_model.py:127-141
    if ((rc = persist()) != ORCA_OK) return rc;
    if ((rc = store()) != ORCA_OK) return rc;
    // the ids that now hold a fresh message: begin() elected them into this step's list
    if (e->list_written && e->overlay_covers_batch) {
      e->have_list = true;
      e->list_bound = 2 * bs < d.q_max ? 2 * bs : d.q_max;
    } else {
      e->overlay_complete = false;
    }
    return ORCA_OK;
  }
  if ((rc = store()) != ORCA_OK) return rc;   // :131-145
  return persist();                           // the positives end without a pending message: nothing to list
}

namespace {
// the counts k_owner_list left in host memory -> e->xm_rows, e->push_rows (once per step: the first caller waits)
int fetch_exchange_rows(orca_engine* e) {
  const orca_engine_desc& d = e->d;
  static const bool off = getenv("ORCA_XM_SIZED") && atoi(getenv("ORCA_XM_SIZED")) == 0;     // (A/B: always the caps, no wait)
  if (!e->sharded || !e->count_pending || off) return ORCA_OK;
  ORCA_HIP_TRY(hipEventSynchronize(e->counted));     // (behind begin()'s first two kernels)
  e->count_pending = false;
  uint32_t most = 0, most_push = 0;
  for (int o = 0; o < d.world; ++o) {
    uint32_t c = 0, c2 = 0;
    for (int64_t b = 0; b < e->counts_blocks; ++b) {
      c += e->counts_host[(b * 2 + 0) * ORCA_MAX_WORLD + o];
      c2 += e->counts_host[(b * 2 + 1) * ORCA_MAX_WORLD + o];
    }
    if (c > most) most = c;
    if (c2 > most_push) most_push = c2;
  }
  auto in64s = [](int64_t n, int64_t cap) { int64_t r = (n + 63) / 64 * 64; r = r < 64 ? 64 : r; return r > cap ? cap : r; };
  e->xm_rows = in64s(most, d.xm_cap);
  e->push_rows = in64s(most_push, 3 * e->per);
  return ORCA_OK;
}
}  // namespace

ORCA_API int orca_engine_xm_rows(orca_engine_t* e, int64_t* rows) {
  ORCA_REQUIRE(e && rows, "orca_engine_xm_rows: null argument");
  const int rc = fetch_exchange_rows(e);
  *rows = e->sharded ? e->xm_rows : e->d.xm_cap;
  return rc;
}

ORCA_API int orca_engine_push_rows(orca_engine_t* e, int64_t* rows) {
  ORCA_REQUIRE(e && rows, "orca_engine_push_rows: null argument");
  const int rc = fetch_exchange_rows(e);
  *rows = e->sharded ? e->push_rows : 0;
  return rc;
}

ORCA_API int orca_engine_apply(orca_engine_t* e, int64_t rows, void* stream_) {
  ORCA_REQUIRE(e, "orca_engine_apply: null handle");
  const orca_engine_desc& d = e->d;
  if (!e->sharded || e->bs == 0) return ORCA_OK;
  ORCA_REQUIRE(rows > 0 && rows <= d.xm_cap, "orca_engine_apply: chunks of %lld rows (xm_cap=%lld)", (long long)rows,
               (long long)d.xm_cap);
  k_apply_rows<<<(unsigned)ceil_div64((int64_t)d.world * rows, 4), 256, 0, (hipStream_t)stream_>>>(
      d.xm, d.world, d.rank, rows, d.n_nodes, d.overlay, d.pending, d.own_count);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_engine_set_overlay_complete(orca_engine_t* e, int complete) {
  ORCA_REQUIRE(e, "orca_engine_set_overlay_complete: null handle");
  e->overlay_complete = complete != 0;
  e->have_list = false;
  return ORCA_OK;
}

ORCA_API int orca_engine_get_overlay_complete(orca_engine_t* e) { return (e && e->overlay_complete) ? 1 : 0; }

ORCA_API int orca_engine_drain(orca_engine_t* e, void* stream_) {
  ORCA_REQUIRE(e, "orca_engine_drain: null handle");
  const orca_engine_desc& d = e->d;
  if (!e->have_list) return ORCA_OK;
  e->have_list = false;
  if (!e->overlay_complete) return ORCA_OK;
  return orca_gru_update(d.gru_blob, 3 * ORCA_D + d.d_e, d.eager_ids + (int64_t)e->cur * d.q_max, e->list_bound,
                         d.eager_count + e->cur, d.messages, d.memory, d.overlay, nullptr, nullptr, stream_);
}
