/*
 * orca_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C CPU restatement of the integer / index half of Orca's temporal hot path.  It is the
 * checker for the HIP path (tests/, __graft_entry__.smoke()) and the sampler leg of bench.py's
 * cpu_baseline; nothing under orca_amd/ may import, link or call it.
 *
 * Every function cites the reference lines (relative to the LuckyLYM/Orca tree) it restates.
 * Pinned against tests/golden/sampler.npz, aggregators.npz and mrd.npz, which were produced by
 * running the reference itself (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -fPIC -shared -o _build/liborca_oracle.so orca_oracle.c   (see oracle/Makefile)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double t; int32_t nbr; int32_t eid; } entry_t;

/* stable merge sort of entry_t by .t (Python's sorted(key=ts) is stable; utils/util.py:101) */
static void merge_sort(entry_t* a, entry_t* tmp, int64_t n) {
  if (n < 2) return;
  if (n <= 16) {
    for (int64_t i = 1; i < n; ++i) {
      entry_t x = a[i];
      int64_t j = i;
      while (j > 0 && a[j - 1].t > x.t) { a[j] = a[j - 1]; --j; }
      a[j] = x;
    }
    return;
  }
  int64_t h = n / 2;
  merge_sort(a, tmp, h);
  merge_sort(a + h, tmp, n - h);
  int64_t i = 0, j = h, k = 0;
  while (i < h && j < n) tmp[k++] = (a[j].t < a[i].t) ? a[j++] : a[i++];
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof(entry_t));
}

/*
 * utils/util.py:91-105  get_neighbor_finder
 * Undirected adjacency: for every edge in input order append (dst,eidx,ts) to src's list and then
 * (src,eidx,ts) to dst's list (:94-96); each list is then stably sorted by ts (:101).  Lists are
 * stored as int32 / int32 / float64 (:102-104).  n_rows = max_node_idx + 1.
 * Output is the ragged lists flattened as a CSR: indptr[n_rows+1], nbr/eid/ts[2E].
 */
int orc_csr_build(int64_t E, const int64_t* src, const int64_t* dst, const int64_t* eidx, const double* ts,
                  int64_t n_rows, int64_t* indptr, int32_t* nbr, int32_t* eid, double* t) {
  for (int64_t i = 0; i <= n_rows; ++i) indptr[i] = 0;
  for (int64_t e = 0; e < E; ++e) {
    if (src[e] < 0 || src[e] >= n_rows || dst[e] < 0 || dst[e] >= n_rows) return -1; /* IndexError upstream */
    indptr[src[e] + 1]++;
    indptr[dst[e] + 1]++;
  }
  for (int64_t i = 0; i < n_rows; ++i) indptr[i + 1] += indptr[i];
  int64_t* cur = (int64_t*)malloc((size_t)n_rows * sizeof(int64_t));
  entry_t* ent = (entry_t*)malloc((size_t)(2 * E + 1) * sizeof(entry_t));
  entry_t* tmp = (entry_t*)malloc((size_t)(2 * E + 1) * sizeof(entry_t));
  if (!cur || !ent || !tmp) { free(cur); free(ent); free(tmp); return -2; }
  memcpy(cur, indptr, (size_t)n_rows * sizeof(int64_t));
  for (int64_t e = 0; e < E; ++e) {
    entry_t a = { ts[e], (int32_t)dst[e], (int32_t)eidx[e] };
    entry_t b = { ts[e], (int32_t)src[e], (int32_t)eidx[e] };
    ent[cur[src[e]]++] = a;
    ent[cur[dst[e]]++] = b;
  }
  for (int64_t v = 0; v < n_rows; ++v) merge_sort(ent + indptr[v], tmp, indptr[v + 1] - indptr[v]);
  for (int64_t i = 0; i < 2 * E; ++i) { nbr[i] = ent[i].nbr; eid[i] = ent[i].eid; t[i] = ent[i].t; }
  free(cur); free(ent); free(tmp);
  return 0;
}

/* utils/util.py:129-131  find_before: i = np.searchsorted(ts[node], cut_time)  (side='left':
 * number of entries with ts < cut_time, i.e. neighbours STRICTLY before t -- SURVEY H1). */
int64_t orc_find_before(const int64_t* indptr, const double* t, int64_t node, double cut) {
  int64_t lo = indptr[node], hi = indptr[node + 1];
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (t[mid] < cut) lo = mid + 1; else hi = mid;
  }
  return lo - indptr[node];
}

/*
 * utils/util.py:134-151  get_temporal_neighbor  (serial loop over queries, :141)
 * + modules/embedding_module.py:130-131  edge_deltas = fp32( fp64(t) - fp64(fp32(t_edge)) ).
 * Outputs are [Q, max(K,1)] row-major, zero-initialised, the last min(K,i) entries before t
 * written right-aligned (:143-150).  K == 0 gives width-1 all-zero rows (:136);
 * the Δt of a padded slot is fp32(t - 0).
 */
int orc_sample_recent(const int64_t* indptr, const int32_t* nbr, const int32_t* eid, const double* t, int64_t n_rows,
                      int64_t Q, const int64_t* nodes, const double* qts, int K,
                      int32_t* out_nbr, int32_t* out_eid, float* out_et, float* out_dt) {
  int W = K > 0 ? K : 1;
  for (int64_t q = 0; q < Q; ++q) {
    int64_t v = nodes[q];
    if (v < 0 || v >= n_rows) return -1;
    int32_t* on = out_nbr + q * W;
    int32_t* oe = out_eid + q * W;
    float* ot = out_et + q * W;
    for (int j = 0; j < W; ++j) { on[j] = 0; oe[j] = 0; ot[j] = 0.0f; }
    int64_t i = orc_find_before(indptr, t, v, qts[q]);
    if (i > 0 && K > 0) {
      int64_t n = i < K ? i : K;
      int64_t base = indptr[v] + i - n;
      for (int64_t j = 0; j < n; ++j) {
        on[K - n + j] = nbr[base + j];
        oe[K - n + j] = eid[base + j];
        ot[K - n + j] = (float)t[base + j];
      }
    }
    if (out_dt) for (int j = 0; j < W; ++j) out_dt[q * W + j] = (float)(qts[q] - (double)ot[j]);
  }
  return 0;
}

/*
 * model/tgn_model.py:174-176  last-occurrence dedupe of a node-id list:
 *   unique, rev_index = np.unique(np.flip(ids), return_index=True); index = len-rev_index-1
 * -> unique ids ascending and, for each, the position of its LAST occurrence in ids.
 * Also the "last" rule of modules/message_aggregator.py:38-54.  Returns U.
 */
typedef struct { int64_t id; int64_t pos; } idpos_t;
static int cmp_idpos(const void* a, const void* b) {
  const idpos_t* x = (const idpos_t*)a; const idpos_t* y = (const idpos_t*)b;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
int64_t orc_segment_last(int64_t M, const int64_t* ids, int64_t* out_unique, int64_t* out_index) {
  if (M == 0) return 0;
  idpos_t* a = (idpos_t*)malloc((size_t)M * sizeof(idpos_t));
  for (int64_t i = 0; i < M; ++i) { a[i].id = ids[i]; a[i].pos = i; }
  qsort(a, (size_t)M, sizeof(idpos_t), cmp_idpos);
  int64_t U = 0;
  for (int64_t i = 0; i < M; ++i) {
    if (i + 1 == M || a[i + 1].id != a[i].id) { out_unique[U] = a[i].id; out_index[U] = a[i].pos; ++U; }
  }
  free(a);
  return U;
}

/*
 * modules/message_aggregator.py:61-81  mean aggregation: per unique id (ascending) the mean over
 * its messages in arrival order (torch.mean over the stacked rows: fp32 sum * (1/m)) and the
 * timestamp of its last message.  msgs is [M, D] row-major.  Returns U.
 */
int64_t orc_segment_mean(int64_t M, int64_t D, const int64_t* ids, const float* msgs, const float* ts,
                         int64_t* out_unique, float* out_msgs, float* out_ts) {
  if (M == 0) return 0;
  idpos_t* a = (idpos_t*)malloc((size_t)M * sizeof(idpos_t));
  for (int64_t i = 0; i < M; ++i) { a[i].id = ids[i]; a[i].pos = i; }
  qsort(a, (size_t)M, sizeof(idpos_t), cmp_idpos);
  int64_t U = 0, start = 0;
  for (int64_t i = 0; i < M; ++i) {
    if (i + 1 == M || a[i + 1].id != a[i].id) {
      int64_t m = i - start + 1;
      float inv = 1.0f / (float)m;
      for (int64_t d = 0; d < D; ++d) {
        float s = 0.0f;
        for (int64_t k = start; k <= i; ++k) s += msgs[a[k].pos * D + d];
        out_msgs[U * D + d] = s * inv;
      }
      out_unique[U] = a[i].id;
      out_ts[U] = ts[a[i].pos];
      ++U;
      start = i + 1;
    }
  }
  free(a);
  return U;
}

/*
 * Uniform temporal sampling.  NOT IN THE REFERENCE (utils/util.py:121-151 has only the most-recent
 * mode) -- "parity unpinned": this restates the definition the product documents in
 * include/orca_hip.h (orca_sample_uniform) so that the HIP kernel has an independent CPU twin.
 * For a query q with i > 0 entries before t, slot j draws r_j = mix(seed ^ mix(q*64 + j)) mod i
 * (splitmix64 finaliser); the K draws are ordered by position (ties by slot) and written to slots
 * 0..K-1; i == 0 gives an all-zero row.  Δt as in orc_sample_recent.
 */
static uint64_t orc_mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
int orc_sample_uniform(const int64_t* indptr, const int32_t* nbr, const int32_t* eid, const double* t, int64_t n_rows,
                       int64_t Q, const int64_t* nodes, const double* qts, int K, uint64_t seed,
                       int32_t* out_nbr, int32_t* out_eid, float* out_et, float* out_dt) {
  if (K < 1 || K > 64) return -3;
  for (int64_t q = 0; q < Q; ++q) {
    int64_t v = nodes[q];
    if (v < 0 || v >= n_rows) return -1;
    int64_t cnt = orc_find_before(indptr, t, v, qts[q]);
    int64_t r[64];
    for (int j = 0; j < K; ++j) {
      out_nbr[q * K + j] = 0; out_eid[q * K + j] = 0; out_et[q * K + j] = 0.0f;
      r[j] = cnt > 0 ? (int64_t)(orc_mix64(seed ^ orc_mix64((uint64_t)q * 64u + (uint64_t)j)) % (uint64_t)cnt) : -1;
    }
    if (cnt > 0) {
      for (int j = 0; j < K; ++j) {
        int rank = 0;
        for (int k = 0; k < K; ++k) rank += (r[k] < r[j]) || (r[k] == r[j] && k < j);
        int64_t p = indptr[v] + r[j];
        out_nbr[q * K + rank] = nbr[p];
        out_eid[q * K + rank] = eid[p];
        out_et[q * K + rank] = (float)t[p];
      }
    }
    for (int j = 0; j < K; ++j) out_dt[q * K + j] = (float)(qts[q] - (double)out_et[q * K + j]);
  }
  return 0;
}
