/*
 * orca_hip.h -- C ABI of liborca_hip.so: the MI355X (gfx950) implementation of Orca's temporal
 * sample -> TGAT aggregate -> GRU memory / raw message -> embedding-reuse hot path.
 *
 * The reference (LuckyLYM/Orca) is pure Python + Numba and has no FFI of its own:
 * its boundary is a set of Python classes.  The Python package orca_amd/ mirrors those classes
 * one-for-one and binds these entry points with ctypes; INTEGRATION.md shows the binding a
 * reference maintainer would add.  Each entry point cites the reference lines it replaces
 * (paths relative to the LuckyLYM/Orca tree).
 *
 * Conventions
 *   - every function returns 0 on success, a negative ORCA_E* code otherwise, and never throws;
 *     orca_last_error() returns a thread-local message for the last failure;
 *   - all array arguments are DEVICE pointers unless the name ends in _host; buffers are
 *     caller-owned; nothing here allocates device memory except orca_csr_create_*;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are asynchronous;
 *   - node id 0 / edge id 0 are the padding entries of the reference's datasets
 *     (utils/preprocess_data.py:55-57,78-79); node/memory/time widths are fixed at ORCA_D = 100
 *     and the attention has 2 heads (train.py:24,56-58 defaults), the edge width d_e is a run-time
 *     argument (<= ORCA_MAX_DE);
 *   - handles are immutable after creation; mutable tables are single-writer per stream.
 */
#ifndef ORCA_HIP_H
#define ORCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORCA_D 100          /* node / memory / time feature width */
#define ORCA_HEADS 2
#define ORCA_MAX_DE 192     /* widest supported edge feature row */
#define ORCA_MAX_K 20       /* most neighbours per query in orca_tgat_forward */
#define ORCA_MAX_WORLD 64   /* most ranks of an owner-computes step (orca_engine_desc) */

#define ORCA_OK 0
#define ORCA_EINVAL -1      /* bad argument (ValueError / assert upstream) */
#define ORCA_EHIP -2        /* HIP runtime failure */
#define ORCA_ENOMEM -3
#define ORCA_ERANGE -4      /* node id out of range (IndexError upstream) */

typedef struct orca_csr orca_csr_t;   /* opaque time-sorted CSR living in HBM */

const char* orca_last_error(void);
int orca_version(void);

/* ---- sampler: utils/util.py:91-151 ------------------------------------------------------ */

/* get_neighbor_finder (utils/util.py:91-105): undirected per-node adjacency, each list stably
 * sorted by timestamp, stored as int32 nbr / int32 edge idx / float64 ts.  n_rows = max_node_idx+1.
 * _host: arrays in host memory, built on the host and uploaded to `device`.
 * _device: arrays already in HBM on the current device, built there (radix sorts). */
int orca_csr_create_host(int64_t n_edges, const int64_t* src_host, const int64_t* dst_host,
                         const int64_t* eidx_host, const double* ts_host, int64_t n_rows, int device,
                         orca_csr_t** out);
int orca_csr_create_device(int64_t n_edges, const int64_t* src, const int64_t* dst, const int64_t* eidx,
                           const double* ts, int64_t n_rows, void* stream, orca_csr_t** out);
/* NeighborFinder(node_to_neighbors, node_to_edge_idxs, node_to_edge_timestamps) (utils/util.py:123-126):
 * adopts ragged lists that are already flattened (host arrays, each list time-sorted). */
int orca_csr_from_arrays_host(int64_t n_rows, const int64_t* indptr_host, const int32_t* nbr_host,
                              const int32_t* eidx_host, const double* ts_host, int device, orca_csr_t** out);
int orca_csr_destroy(orca_csr_t* csr);
/* device pointers of the CSR arrays (for tests / export); any out pointer may be NULL */
int orca_csr_arrays(const orca_csr_t* csr, int64_t* n_rows, int64_t* nnz, const int64_t** indptr,
                    const int32_t** nbr, const int32_t** eidx, const double** ts);
/* copies the CSR arrays back into caller-owned HOST buffers ([n_rows+1], [nnz] x3; any may be NULL):
 * the node_to_neighbors / node_to_edge_idxs / node_to_edge_timestamps lists of utils/util.py:98-105 */
int orca_csr_copy_host(const orca_csr_t* csr, int64_t* indptr_host, int32_t* nbr_host, int32_t* eidx_host,
                       double* ts_host);

/* same for the entries [begin, end) only (the adjacency lists of a few nodes of a graph too large to copy whole) */
int orca_csr_copy_range_host(const orca_csr_t* csr, int64_t begin, int64_t end, int32_t* nbr_host, int32_t* eidx_host,
                             double* ts_host);

/* NeighborFinder.get_temporal_neighbor (utils/util.py:134-151) + the Δt of
 * modules/embedding_module.py:130-131.  Outputs are [Q, K] row-major, right-aligned, zero padded;
 * neighbours are STRICTLY before t (np.searchsorted side='left').  out_dt may be NULL.  K >= 1
 * (the K == 0 quirk of :136 is handled by the Python shim).  Node ids outside [0, n_rows) set
 * *err_flag (device int, may be NULL) and produce an all-padding row. */
int orca_sample_recent(const orca_csr_t* csr, const int64_t* nodes, const double* ts, int64_t Q, int K,
                       int32_t* out_nbr, int32_t* out_eidx, float* out_et, float* out_dt,
                       int* err_flag, void* stream);
/* NeighborFinder.find_before (utils/util.py:129-131): out_count[q] = #entries of nodes[q] before ts[q]. */
int orca_find_before(const orca_csr_t* csr, const int64_t* nodes, const double* ts, int64_t Q,
                     int64_t* out_count, void* stream);
/* Not in the reference ("parity unpinned"): K draws with replacement among the
 * entries before t from a counter RNG keyed (seed, q0 + query, slot), returned in time order.  q0 is the index of
 * nodes[0] in the caller's numbering: a call on the slice [q0, q0+Q) of a query list draws what a call on the
 * whole list draws for those queries. */
int orca_sample_uniform(const orca_csr_t* csr, const int64_t* nodes, const double* ts, int64_t Q, int K,
                        uint64_t seed, int64_t q0, int32_t* out_nbr, int32_t* out_eidx, float* out_et, float* out_dt,
                        int* err_flag, void* stream);

/* ---- time encoder: model/time_encoding.py:23-28 ------------------------------------------ */
/* out[i, :] = cos(t[i] * w + b), w,b [ORCA_D] */
int orca_time_encode(const float* t, int64_t n, const float* w, const float* b, float* out, void* stream);

/* ---- TGAT attention layer: model/temporal_attention.py:24-56 ----------------------------- */

/* Number of floats of the packed parameter blob for edge width d_e. */
int64_t orca_tgat_blob_floats(int d_e);
/* Re-lays one TemporalAttentionLayer's parameters (nn.MultiheadAttention with kdim != embed_dim,
 * MergeLayer utils/util.py:12-24, LayerNorm) into MFMA-fragment order.  Inputs are the raw
 * row-major state_dict tensors; must be re-run after the parameters change. */
int orca_tgat_pack(int d_e, const float* q_proj_w, const float* k_proj_w, const float* v_proj_w,
                   const float* in_proj_b, const float* out_proj_w, const float* out_proj_b,
                   const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                   const float* ln_w, const float* ln_b, float* blob, void* stream);
/* Optional second step after orca_tgat_pack for blobs used in gather mode: the source's own time encoding
 * there is the constant cos(time_b) (modules/embedding_module.py:126, a zero time delta), so the q projection
 * (model/temporal_attention.py:48, in_proj of nn.MultiheadAttention) multiplies into the key map and the
 * kernel skips one GEMM stage.  The blob is then tied to this time_b: gather-mode calls must pass the same
 * values in orca_tgat_inputs.time_b; dense-mode calls are unaffected.  Re-run after either changes. */
int orca_tgat_pack_gather(int d_e, const float* q_proj_w, const float* k_proj_w, const float* in_proj_b,
                          const float* time_b, float* blob, void* stream);

/* Where the rows of one attention call come from.  Two modes:
 *  gather: rows are fetched by index from feature tables (the fused path used by
 *          modules/embedding_module.py:116-285), time features are computed in-kernel;
 *  dense:  the six dense tensors of TemporalAttentionLayer.forward (the drop-in boundary). */
typedef struct {
  /* source-node rows [*,ORCA_D]: row = src_idx ? src_idx[q] : q.  If src_flags && src_flags[row]
   * the row is read from src_overlay instead (memory with pending messages applied). */
  const float* src_table;
  const int32_t* src_idx;
  const float* src_overlay;
  const uint8_t* src_flags;
  /* neighbour rows: row = nbr_idx[q*K+j] (gather) or q*K+j (nbr_idx == NULL, dense) */
  const float* nbr_table;
  const int32_t* nbr_idx;
  const float* nbr_overlay;
  const uint8_t* nbr_flags;
  /* padding mask: nbr_ids[q*K+j] == 0 (embedding_module.py:134) or, if nbr_ids == NULL, mask bytes
   * (torch.bool, true = padded).  In mask mode the all-padded fix-up of temporal_attention.py:46-47
   * is written back into `mask`, as the reference mutates its argument. */
  const int32_t* nbr_ids;
  uint8_t* mask;
  /* edge rows [*, d_e]: row = edge_idx[q*K+j], or q*K+j when edge_idx == NULL */
  const float* edge_table;
  const int32_t* edge_idx;
  /* time features: computed as cos(dt*w+b) when dt != NULL (src uses dt = 0), else read from
   * src_time [Q,ORCA_D] and nbr_time [Q*K,ORCA_D] */
  const float* dt;
  const float* time_w;
  const float* time_b;
  const float* src_time;
  const float* nbr_time;
  /* bit 0 set (gather mode only): the caller states that `blob` was completed by orca_tgat_pack_gather with these
   * time_b values; the kernel then takes q~ straight from the source rows.  0 = the general path.
   * Bits 8-15: the launch shape, ORCA_TGAT_WAVES(q_folded) = waves per query.  0 or 1: one wavefront per query, 16 queries
   * per workgroup -- the form for batches that fill the chip.  4 (q_folded = 1 | 4 << 8): four wavefronts share a query (each
   * takes every fourth neighbour slot through the per-query stages; the parts meet in LDS) and a workgroup holds 4 queries:
   * for launches of <= ~1000 queries (the reference's batch size of 200 interactions), where one workgroup's latency is what
   * the launch costs.  Honoured for K = 10, d_e = 172 or 1, a folded q and the plain forward (not orca_tgat_forward_saving);
   * ignored otherwise.  Same function; the sums over the slots are taken in another order, so results agree with the
   * one-wave form to rounding, not to the bit.  (Kept inside this word: as a field of its own it cost the bench's launches
   * 1 %, reproducibly and for a reason that was not isolated -- padding the kernel's argument list alone does not.) */
  int32_t q_folded;
#define ORCA_TGAT_WAVES(q_folded) (((q_folded) >> 8) & 0xff)
  /* Attention-weight dropout of nn.MultiheadAttention (model/temporal_attention.py:17-21; train.py:30 default 0.4):
   * weight (query q, head h, slot j) is kept with probability 1 - dropout_p and scaled by 1 / (1 - dropout_p).  The
   * keep decision is a counter RNG keyed (dropout_seed, q0 + q, h, j) -- orca_tgat_dropout_mask writes the same
   * multipliers out; the RNG STREAM is not torch's (parity unpinned), the masked arithmetic is pinned by feeding
   * that mask to the oracle.  0 = off (module.eval()).  dropout_q0: index of query 0 in the caller's numbering. */
  float dropout_p;
  uint64_t dropout_seed;
  int64_t dropout_q0;
} orca_tgat_inputs;
/* mult[q][h][j] (float [Q, ORCA_HEADS, K]) = 0 or 1 / (1 - p): the dropout multipliers orca_tgat_forward /
 * orca_tgat_backward apply for (seed, q0) */
int orca_tgat_dropout_mask(uint64_t seed, int64_t q0, int64_t Q, int K, float p, float* mult, void* stream);

/* out [Q, ORCA_D]; out_weights [Q, K] (head-averaged attention, may be NULL).  1 <= K <= ORCA_MAX_K. */
int orca_tgat_forward(const float* blob, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                      float* out, float* out_weights, void* stream);

/* ---- backward of the attention layer: torch autograd of model/temporal_attention.py:24-56 as the training
 * loop runs it (train.py:236-237 loss.backward()) -------------------------------------------------------- */
/* The layer's raw row-major state_dict tensors, or gradient buffers of the same shapes. */
typedef struct {
  float* q_w;    /* multi_head_target.q_proj_weight [200,200] */
  float* k_w;    /* multi_head_target.k_proj_weight [200,200+d_e] */
  float* v_w;    /* multi_head_target.v_proj_weight [200,200+d_e] */
  float* in_b;   /* multi_head_target.in_proj_bias [600] */
  float* o_w;    /* multi_head_target.out_proj.weight [200,200] */
  float* o_b;    /* multi_head_target.out_proj.bias [200] */
  float* fc1_w;  /* merger.fc1.weight [100,300] */
  float* fc1_b;  /* merger.fc1.bias [100] */
  float* fc2_w;  /* merger.fc2.weight [100,100] */
  float* fc2_b;  /* merger.fc2.bias [100] */
  float* ln_w;   /* norm.weight [100] */
  float* ln_b;   /* norm.bias [100] */
} orca_tgat_params;
/* Floats of scratch orca_tgat_backward needs for Q queries of K slots (-1: unsupported sizes). */
int64_t orca_tgat_backward_workspace_floats(int d_e, int64_t Q, int K);
/* Given grad_out = dL/d out [Q, ORCA_D] of orca_tgat_forward on the same `in` (dropout off), ADDS the parameter
 * gradients into `grads` (caller zeroes them, as optimizer.zero_grad does) and writes the input gradients:
 * d_src [Q,ORCA_D] (per query; may be NULL), d_nbr [Q*K,ORCA_D] (per slot, zero for padded slots; the caller
 * scatter-adds rows that came from a table), and for the time features either d_src_time [Q,ORCA_D] /
 * d_nbr_time [Q*K,ORCA_D] (dense mode, may be NULL) or, in gather mode, ADDS into d_time_w / d_time_b [ORCA_D]
 * (time_encoding.py:14; both or neither).  Edge features are data upstream and get no gradient.  `in->mask`
 * must be the mask as it was BEFORE the forward call (the forward un-masks slot 0 of all-padded queries in
 * place, :46-47).  `params` are read, not modified.  d_nbr may be NULL when nobody reads the neighbour rows' gradient (detached history
 * rows, modules/history.py:32-40).  workspace: orca_tgat_backward_workspace_floats floats. */
int orca_tgat_backward(const orca_tgat_params* params, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                       const float* grad_out, const orca_tgat_params* grads, float* d_src, float* d_nbr,
                       float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b, float* workspace,
                       void* stream);

/* A forward call that keeps what the backward would otherwise recompute (the attention core: q~, the softmax weights,
 * the weighted key sums; train.py:225-237 keeps the same tensors alive in the autograd graph): `saved` is
 * orca_tgat_saved_floats(d_e, Q, K) floats, written by orca_tgat_forward_saving and read by orca_tgat_backward_saved,
 * which is orca_tgat_backward without its recompute of those (same inputs, same results to rounding). */
int64_t orca_tgat_saved_floats(int d_e, int64_t Q, int K);
int orca_tgat_forward_saving(const float* blob, int d_e, const orca_tgat_inputs* in, int64_t Q, int K, float* out,
                             float* out_weights, float* saved, void* stream);
int orca_tgat_backward_saved(const orca_tgat_params* params, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                             const float* grad_out, const orca_tgat_params* grads, float* d_src, float* d_nbr,
                             float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b, float* workspace,
                             const float* saved, void* stream);
/* The same call in two parts (autograd's engine is free to order the parameter-gradient nodes of train.py:236
 * loss.backward() after the data-gradient chain; here the caller does): phases 1 = the data gradients (d_src, d_nbr,
 * the time gradients' per-query terms), 2 = the parameter gradients (grads, d_time_w/b), which read what phase 1 left in
 * `workspace` -- same arguments, workspace untouched in between, any stream ordered after phase 1 (a second stream
 * lets the reductions over the queries run beside the next layer's chain); 3 = both = orca_tgat_backward(_saved).
 * saved may be NULL (recompute, as orca_tgat_backward). */
int orca_tgat_backward_phase(const orca_tgat_params* params, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                             const float* grad_out, const orca_tgat_params* grads, float* d_src, float* d_nbr,
                             float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b, float* workspace,
                             const float* saved, int phases, void* stream);

/* orca_tgat_pack (+ orca_tgat_pack_gather when time_b != NULL) for n_layers layers at once: params[l] are layer l's raw
 * tensors, blobs[l] its blob.  The folds of all layers share launches (3 per two layers instead of 8 per layer): the
 * re-pack after every optimizer step (train.py:238) is otherwise two dozen latency-bound launches.
 * blobs_packed_before != 0: every blobs[l] already holds a packed layer of the same d_e (only the values change). */
int orca_tgat_pack_layers(int n_layers, int d_e, const orca_tgat_params* params, const float* time_b,
                          float* const* blobs, int blobs_packed_before, void* stream);


/* Profiling aid (no counterpart in the reference): when buf != NULL (device int64 [workgroups * 32]) later
 * orca_tgat_forward launches write, per workgroup, the 100 MHz real-time counter at each internal phase boundary
 * (words 0..15) and the shader clock counter beside it (words 16..31). */
int orca_debug_tgat_phase_clock(void* buf);
/* same for the streaming GRU kernel: device int64 [ceil(n/64)*7 * 4 waves * 4 stamps] */
int orca_debug_gru_phase_clock(void* buf);

/* ---- GRU memory updater: modules/memory_updater.py:25-86 --------------------------------- */

int64_t orca_gru_blob_floats(int msg_dim);
/* nn.GRUCell(msg_dim -> ORCA_D) parameters (weight_ih [300,msg_dim], weight_hh [300,100], biases) */
int orca_gru_pack(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                  const float* bias_hh, float* blob, void* stream);
/* the same for a blob that orca_gru_pack filled before with the same msg_dim (new values after optimizer.step(),
 * train.py:238): skips the zero fill of the padding */
int orca_gru_repack(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                    const float* bias_hh, float* blob, void* stream);
/* For r in [0, n): id = ids ? ids[r] : r;  out[id] = GRUCell(x[id], h[id]).  n is read from *n_dev
 * when n_dev != NULL (device-resident work-list length; rows >= *n_dev are skipped and `n` is
 * then the launch upper bound).  out may alias h (in-place persist, update_memory :25-39) or be the
 * overlay table (get_updated_memory :56-78 without the full-table clone).  If last_update != NULL,
 * last_update[id] = msg_ts[id] (:37). */
int orca_gru_update(const float* blob, int msg_dim, const int32_t* ids, int64_t n, const int32_t* n_dev,
                    const float* x, const float* h, float* out, float* last_update, const float* msg_ts,
                    void* stream);

/* orca_gru_update that also keeps, for row r (id = ids[r]), the gates r, z, n and the hidden part of n's pre-activation
 * in gates[r][4][ORCA_D] (what autograd keeps of nn.GRUCell for train.py:236): orca_gru_backward_saved then does not
 * recompute them.  Built for msg_dim = 3 * ORCA_D + 172 with an id list, out of place (EINVAL otherwise: call
 * orca_gru_update and orca_gru_backward). */
int orca_gru_update_saving(const float* blob, int msg_dim, const int32_t* ids, int64_t n, const int32_t* n_dev,
                           const float* x, const float* h, float* out, float* gates, void* stream);

/* Backward of GRUCell as get_updated_memory calls it (modules/memory_updater.py:56-78,86; train.py:236-237): for the
 * n rows id = ids ? ids[r] : r with inputs x[id] [msg_dim], h[id] [ORCA_D] and d_out [n, ORCA_D] = dL/d(new h) in
 * row order, ADDS the parameter gradients into g_* (shapes of the raw tensors) and writes d_x [n, msg_dim] and
 * d_h [n, ORCA_D] when they are not NULL (upstream both are detached: the stored messages and the memory table
 * carry no graph).  The parameters are the raw row-major tensors, not the packed blob. */
int64_t orca_gru_backward_workspace_floats(int msg_dim, int64_t n);
int orca_gru_backward(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                      const float* bias_hh, const int32_t* ids, int64_t n, const float* x, const float* h,
                      const float* d_out, float* g_weight_ih, float* g_weight_hh, float* g_bias_ih, float* g_bias_hh,
                      float* d_x, float* d_h, float* workspace, void* stream);
/* in two parts like orca_tgat_backward_phase: phases 1 = row gathers, gate derivatives, d_x / d_h; 2 = the parameter
 * gradients from what phase 1 left in `workspace`; 3 = both = orca_gru_backward */
int orca_gru_backward_phase(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                            const float* bias_hh, const int32_t* ids, int64_t n, const float* x, const float* h,
                            const float* d_out, float* g_weight_ih, float* g_weight_hh, float* g_bias_ih, float* g_bias_hh,
                            float* d_x, float* d_h, float* workspace, int phases, void* stream);

/* orca_gru_backward_phase reading the gates orca_gru_update_saving kept for the same id list (gates == NULL: as
 * orca_gru_backward_phase) */
int orca_gru_backward_saved(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                            const float* bias_hh, const int32_t* ids, int64_t n, const float* x, const float* h,
                            const float* d_out, float* g_weight_ih, float* g_weight_hh, float* g_bias_ih, float* g_bias_hh,
                            float* d_x, float* d_h, float* workspace, const float* gates, int phases, void* stream);

/* dst_table[id] = src_table[id] (rows of ORCA_D floats) and, if last_update != NULL, last_update[id] =
 * msg_ts[id], for the first min(n, *n_dev) ids of a work-list: commits GRU rows held in the overlay into
 * the persistent memory (modules/memory_updater.py:36-39).  Ids outside [0, n_nodes) are skipped. */
int orca_commit_rows(const int32_t* ids, int64_t n, const int32_t* n_dev, const float* src_table,
                     float* dst_table, float* last_update, const float* msg_ts, int64_t n_nodes, void* stream);

/* For every position p < n with pending[ids[p]]: memory[id] = overlay[id], last_update[id] = msg_ts[id]
 * (duplicates copy the same values).  modules/memory_updater.py:25-39 when the overlay already holds
 * GRU(message, memory) of those ids; pending flags are not cleared (see orca_engine_finish). */
int orca_persist_overlay(const int64_t* ids, int64_t n, const uint8_t* pending, const float* overlay,
                         float* memory, float* last_update, const float* msg_ts, int64_t n_nodes, void* stream);

/* ---- node-id bookkeeping that the reference does with NumPy on the host ------------------- */

/* Writes to worklist every distinct id of the concatenation of ids32 (n32 int32 values, e.g. the
 * sampled neighbours) and ids64 (n64 int64 values, e.g. the roots) whose flags[id] != 0 (flags may
 * be NULL = take all), and their number to *count (device int, zeroed here).  `slot` is the caller's
 * int64 [n_nodes] scratch table and `epoch` a value greater than any used before on it plus the
 * length of that use (as in orca_segment_last).  Replaces np.unique + memory.nodes[index] of
 * modules/memory_updater.py:60-65 and embedding_module.py:121,160.  Order of the list is unspecified. */
int orca_worklist_build(const int32_t* ids32, int64_t n32, const int64_t* ids64, int64_t n64,
                        const uint8_t* flags, int64_t* slot, int64_t epoch, int64_t n_nodes,
                        int32_t* worklist, int32_t* count, void* stream);
int orca_fill_i32(int32_t* p, int64_t n, int32_t v, void* stream);
int orca_fill_u8(uint8_t* p, int64_t n, uint8_t v, void* stream);
/* flags[ids[i]] = v for i < n (ids int64, or the int32 work-list when ids64 == NULL).  Every id-indexed entry point
 * of this header takes the size of the table it indexes and SKIPS ids outside it (a gather reads zeros): a bad id is
 * never an out-of-bounds access; callers that want the reference's IndexError check their ids on the host, as
 * orca_amd.api does for the NumPy ids the reference's interface passes. */
int orca_set_flags(uint8_t* flags, const int64_t* ids64, const int32_t* ids32, int64_t n,
                   const int32_t* n_dev, uint8_t v, int64_t n_nodes, void* stream);

/* ---- raw messages: model/tgn_model.py:131-141,172-191 + modules/memory.py:25-28 ----------- */

/* Last-occurrence segment reduce ("scatter-argmax keyed by node id"): for the id list
 * ids[0..M) marks winner[p] = 1 iff no later position holds the same id
 * (np.unique(np.flip(ids), return_index=True), tgn_model.py:174-176; also the 'last' rule of
 * modules/message_aggregator.py:38-54).  `slot` is a caller-owned int64 [n_nodes] scratch table,
 * `epoch` a value greater than any used before on it. */
int orca_segment_last(const int64_t* ids, int64_t M, int64_t* slot, int64_t epoch, int64_t n_nodes,
                      uint8_t* winner, void* stream);
/* orca_segment_last for callers that cannot allocate or own a scratch table (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the reference.  The election table is
 * a file-scope device array of ORCA_STATIC_IDS_MAX int64 (8 MiB per GPU, held until the process ends); the call
 * allocates nothing and rejects n_nodes or M beyond that maximum with ORCA_EINVAL.  The table is all zero between
 * calls (a call zeroes the entries it used) and no state of a call lives on the host, so a call recorded into a HIP
 * graph may be replayed any number of times, with other calls in between (take the first call on a device outside
 * the capture: it looks the table's address up).  Every call on a GPU shares that GPU's table: calls whose KERNELS
 * overlap on one GPU (two streams) race -- serialise them; host threads may call concurrently otherwise. */
#define ORCA_STATIC_IDS_MAX (1 << 20)
int orca_segment_last_static(const int64_t* ids, int64_t M, int64_t n_nodes, uint8_t* winner, void* stream);
/* modules/message_aggregator.py:61-81: per distinct id the mean of its rows (arrival order) and the
 * timestamp of its last row, written at the id's LAST position p (winner[p] = 1). */
int orca_segment_mean(const int64_t* ids, int64_t M, int64_t D, const float* msgs, const float* ts,
                      int64_t* slot, int64_t epoch, int64_t n_nodes, uint8_t* winner, float* out_msgs,
                      void* stream);
/* For the batch (src,dst,eidx,t)[bs]: ids = [src;dst], partner = [dst;src]; for every last
 * occurrence p:  messages[id] = [memory[id] | memory[partner] | edge_feats[eidx] |
 * cos((fp32(t) - last_update[id]) * w + b)], msg_ts[id] = fp32(t), pending[id] = 1. */
int orca_message_store(const int64_t* src, const int64_t* dst, const int64_t* eidx, const double* t,
                       int64_t bs, const float* memory, const float* last_update,
                       const float* edge_feats, int d_e, const float* time_w, const float* time_b,
                       int64_t* slot, int64_t epoch, int64_t n_nodes,
                       float* messages, float* msg_ts, uint8_t* pending, void* stream);

/* ---- embedding-reuse cache: modules/history.py:25-46, embedding_module.py:103-113,186-212 - */

/* History.push (history.py:38-40): emb[ids[p]] = x[p]; duplicate ids resolve to the LAST position
 * (SURVEY H7).  slot/epoch as in orca_segment_last. */
int orca_history_push(float* emb, const float* x, const int64_t* ids, int64_t n, int64_t* slot,
                      int64_t epoch, int64_t n_nodes, void* stream);
/* History.pull (history.py:32-36) / any row gather: out[p] = table[idx[p]], rows of `width` floats.
 * idx is int64, or int32 when idx64 == NULL; an index outside [0, n_rows) gives a row of zeros. */
int orca_gather_rows(const float* table, int64_t n_rows, int64_t width, const int64_t* idx64, const int32_t* idx32,
                     int64_t n, float* out, void* stream);
/* Gradient accumulation onto the rows that get_updated_memory produced (memory_updater.py:56-78): autograd's
 * index backward of `updated_memory[nodes]` (embedding_module.py:121,160), restricted to the flagged (pending) ids,
 * whose rows are GRU outputs -- the other rows of the table are detached upstream.
 * orca_index_positions: pos[list[i]] = i for i < min(n, *n_dev) (list = the distinct pending ids, n_dev may be NULL).
 * orca_scatter_add_flagged: dst[pos[id]] += rows[p] (ORCA_D floats) for every p < n whose id = idx[p] has flags[id];
 * idx is int64, or int32 when idx64 == NULL; dst is zeroed by the caller. */
int orca_index_positions(const int32_t* list, int64_t n, const int32_t* n_dev, int32_t* pos, int64_t n_nodes, void* stream);
int orca_scatter_add_flagged(const float* rows, const int64_t* idx64, const int32_t* idx32, int64_t n,
                             const uint8_t* flags, const int32_t* pos, int64_t n_nodes, float* dst, void* stream);
/* two lists into the same dst in one launch (the gradients of the source rows and of the neighbour rows of attention
 * layer 1 both flow into the updated-memory rows, embedding_module.py:159-175): same result as two calls up to the
 * order of the floating-point additions */
int orca_scatter_add_flagged_pair(const float* rows_a, const int64_t* idx64_a, const int32_t* idx32_a, int64_t n_a,
                                  const float* rows_b, const int64_t* idx64_b, const int32_t* idx32_b, int64_t n_b,
                                  const uint8_t* flags, const int32_t* pos, int64_t n_nodes, float* dst, void* stream);
/* find_uncached_out_of_batch_neighbors (modules/embedding_module.py:200-212): the distinct ids of nbr [n_nbr] (int32,
 * the sampled neighbours) that are not among batch_nodes [n_batch] and have cache_flag[id] == 0, ascending, WITHOUT
 * the first of them (the reference's `[1:]`, which assumes it is the padding id 0: SURVEY H8) -> out[0 .. *count),
 * at most cap entries written (*count is the full number).  slot / epoch as in orca_segment_last (one epoch);
 * bitmap: caller-owned uint32 [ceil(n_nodes / 32)], all zero on entry and left all zero. */
int orca_find_uncached(const int32_t* nbr, int64_t n_nbr, const int64_t* batch_nodes, int64_t n_batch,
                       const uint8_t* cache_flag, int64_t* slot, int64_t epoch, int64_t n_nodes, uint32_t* bitmap,
                       int64_t* out, int64_t cap, int32_t* count, void* stream);
/* out_table[ids[p]] = x[p] (ids unique; Memory.set_memory modules/memory.py:33-34); ids outside [0, n_rows) are skipped */
int orca_scatter_rows(float* table, int64_t n_rows, int64_t width, const int64_t* ids, int64_t n, const float* x,
                      void* stream);

/* The link decoder of a TRAINING step with its loss, forward and backward (model/tgn_model.py:150-157: affinity_score =
 * MergeLayer utils/util.py:12-24 on (src, dst) and (src, neg), sigmoid; train.py:231-233: BCELoss(pos, 1) +
 * BCELoss(neg, 0), mean reduction).  emb [3*bs, ORCA_D] = the embeddings of [src; dst; neg]; fc1_w [ORCA_D, 2*ORCA_D],
 * fc1_b [ORCA_D], fc2_w [1, ORCA_D], fc2_b [1] (affinity_score.fc{1,2}.*).  Out: *loss (device scalar), prob [2*bs] =
 * [pos; neg] probabilities (may be NULL), d_emb [3*bs, ORCA_D] = dL/d emb, and the parameter gradients ACCUMULATED into
 * g_* (zeroed by the caller).  workspace: orca_decoder_train_workspace_floats(bs) floats. */
int64_t orca_decoder_train_workspace_floats(int64_t bs);
int orca_decoder_train(const float* emb, int64_t bs, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                       const float* fc2_b, float* loss, float* prob, float* d_emb, float* g_fc1_w, float* g_fc1_b,
                       float* g_fc2_w, float* g_fc2_b, float* workspace, void* stream);

/* optimizer.step() of train.py:238 for torch.optim.Adam(tgn.parameters(), lr) (train.py:180; no weight decay, no
 * amsgrad) over n tensors in one launch per 24 tensors: params / grads / exp_avg / exp_avg_sq and sizes are HOST arrays
 * of n device pointers / element counts; step = 1 for the first call (the caller counts).  Arithmetic in the order of
 * torch's multi-tensor implementation. */
int orca_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* sizes, double lr, double beta1, double beta2, double eps,
                   int64_t step, void* stream);

/* ---- native step driver: model/tgn_model.py:95-147 with modules/embedding_module.py:154-182 ----- */

/* One TGN batch step in the embedding-reuse (--reuse, budget 0) formulation, issued from C++ on one
 * stream.  Every table and work buffer is caller-owned device memory, described once: */
#define ORCA_ENGINE_MAX_LAYERS 4
typedef struct orca_engine orca_engine_t;
typedef struct {
  int64_t n_nodes;                 /* rows of the state tables (args.n_nodes, train.py:129) */
  int32_t n_layers, K, d_e;
  int32_t small_batch;             /* nonzero: attention launches of <= 1024 queries take the four-waves-per-query form
                                      (ORCA_TGAT_WAVES of orca_tgat_inputs.q_folded = 4): faster at the reference's batch size, equal to
                                      the default form to rounding, not to the bit.  One rank only (an N-rank run is held
                                      bit-equal to one rank in the default form) */
  const orca_csr_t* csr;           /* may be NULL here and set later (orca_engine_set_csr) */
  const float* edge_feats;         /* [n_edges+1, d_e] */
  const float* time_w;             /* [ORCA_D] time_encoder.w.weight */
  const float* time_b;             /* [ORCA_D] time_encoder.w.bias */
  const float* att_blob[ORCA_ENGINE_MAX_LAYERS];   /* orca_tgat_pack output per layer */
  const float* gru_blob;           /* orca_gru_pack output */
  float* memory;                   /* [n_nodes, ORCA_D]   modules/memory.py:18 */
  float* last_update;              /* [n_nodes]           :19 */
  float* messages;                 /* [n_nodes, 300+d_e]  :22 */
  float* msg_ts;                   /* [n_nodes]           :23 */
  uint8_t* pending;                /* [n_nodes]           :21 (host bool array upstream) */
  float* overlay;                  /* [n_nodes, ORCA_D]   GRU(message, memory) rows of get_updated_memory */
  float* hist[ORCA_ENGINE_MAX_LAYERS - 1];         /* [n_nodes, ORCA_D] per inner layer, modules/history.py:11 */
  int64_t* slot;                   /* [n_nodes] election scratch (epochs handed out by the engine) */
  int32_t* worklist;               /* [n_nodes] */
  int32_t* count;                  /* [1] */
  int64_t q_max;                   /* capacity of the work buffers in root queries */
  int64_t* nodes;                  /* [q_max]  [src;dst;neg]        tgn_model.py:99 */
  double* qts;                     /* [q_max]  [t;t;t]              :102 */
  int32_t* nodes32;                /* [q_max] */
  int32_t* nbr;                    /* [q_max*K] sampler outputs of this rank's slice */
  int32_t* eid;                    /* [q_max*K] */
  float* et;                       /* [q_max*K] */
  float* dt;                       /* [q_max*K] */
  float* emb[2];                   /* [q_max, ORCA_D] each: layer outputs ping-pong (layer l -> emb[(l+1)&1]) */
  uint8_t* winners;                /* [2*q_max] last-occurrence flags of the push / message-store id lists; may be
                                      NULL (the elections then run on `slot` with global atomics) */
  int32_t* eager_ids;              /* [2, q_max] or NULL: the distinct ids whose raw message a train-ordered step
                                      stored (two halves, written and consumed by alternate steps) */
  int32_t* eager_count;            /* [2] their counts; zero-initialised by the caller */
  int32_t att_q_folded;            /* nonzero: every att_blob was completed by orca_tgat_pack_gather with time_b */
  /* ---- several ranks (not in the reference): owner-computes sharding of the state ----
   * owner(id) = id % world.  In a train-ordered step every rank embeds its slice of the batch's root queries and
   * runs the memory persist, the raw-message store and the GRU only for the ids it owns; the rows other ranks
   * read are exchanged through buffers the CALLER all-gathers (RCCL) between the calls:
   *   emb[b]: rank-major, 3*per rows per rank, per = ceil(bs / world): a rank embeds the src, dst and neg roots of
   *          ITS INTERACTIONS [lo, hi) = [rank*per, ...) (rows [src | dst | neg] of that range, the layout a
   *          data-parallel link decoder wants, and the same mix of hub and random roots on every rank); a rank
   *          writes and reads only its own rows of it (callers that want every rank's output gather it themselves);
   *   xl    : the history push of a layer's output on every rank's replica.  begin() (layer 1) and layer() (inner
   *          layers) pack the rows of this rank's output that win the push election -- the last occurrence of their
   *          id among the batch's roots -- into xl_own as a chunk like xm's: [count, 63 unused | records of (id, 3
   *          unused, row)]; the caller all-gathers the first orca_engine_xm_chunk_words(P) words of every rank's
   *          xl_own into xl, P = orca_engine_push_rows (the most winners any rank has, in 64s, at most 3*per; the same
   *          for every layer of the step), and the next layer() applies them.  Sized for q_max / world rows;
   *   xm    : GRU(message, memory) of the ids a rank owns among the batch's positives.  A chunk of R rows is
   *          orca_engine_xm_chunk_words(R) 32-bit words: [count, 63 unused | R records of (id, 3 unused, the ORCA_D
   *          floats of the row)].  finish() packs this rank's chunk into xm_own (sized for xm_cap rows; record r lies
   *          at the same place whatever R); the caller picks this step's R -- any R with max over the owners of
   *          their count <= R <= xm_cap, the same on every rank -- all-gathers the first orca_engine_xm_chunk_words(R)
   *          words of every rank's xm_own into xm (rank-major, chunk c at word c * orca_engine_xm_chunk_words(R)) and
   *          passes R to orca_engine_apply.  orca_engine_xm_rows gives that R: with n_layers >= 2 begin() counts, for
   *          every owner, a number that is at least the distinct positives of the batch it owns (and rarely more) --
   *          every rank from the same global id list, so no collective is needed to agree on R; with one layer
   *          R = xm_cap.  A rank whose count exceeds xm_cap sets bit 0 of *err and sends xm_cap rows.
   * world <= 1: all of these may be 0 / NULL. */
  int32_t world, rank;
  int32_t* xm;                     /* [world * orca_engine_xm_chunk_words(xm_cap)] the gathered chunks */
  int32_t* xm_own;                 /* [orca_engine_xm_chunk_words(xm_cap)] this rank's chunk */
  int32_t* xl;                     /* [world * orca_engine_xm_chunk_words(q_max / world)] gathered push records (n_layers >= 2) */
  int32_t* xl_own;                 /* [orca_engine_xm_chunk_words(q_max / world)] this rank's push records */
  int64_t xm_cap;                  /* rows a rank can send per step; more owned ids than this set bit 0 of *err */
  int32_t* own_ids;                /* [q_max] the distinct ids this rank owns among the batch's positives */
  int32_t* own_pos;                /* [q_max] the position of the last occurrence of each in [src;dst] */
  int32_t* own_count;              /* [1], zero-initialised by the caller */
  int64_t* slot_m;                 /* [n_nodes] second election table (message list; `slot` serves the history push) */
  int32_t* err;                    /* [1] device error word, zero-initialised by the caller (bit 0: xm overflow) */
} orca_engine_desc;
int64_t orca_engine_xm_chunk_words(int64_t rows);

/* Neighbour sampling of the step driver: ORCA_SAMPLE_RECENT (utils/util.py:134-151, the reference's only mode) or
 * ORCA_SAMPLE_UNIFORM (orca_sample_uniform: not in the reference, BASELINE config 5) with the seed of the NEXT
 * step; the RNG key of a root is its index in the batch's [src;dst;neg] list, so slices of several ranks draw what
 * one rank draws. */
#define ORCA_SAMPLE_RECENT 0
#define ORCA_SAMPLE_UNIFORM 1
int orca_engine_set_sampling(orca_engine_t* e, int mode, uint64_t seed);

int orca_engine_create(const orca_engine_desc* desc, orca_engine_t** out);
/* Measurement aid: enable = n > 0 attaches a HIP event pair to the sampler / GRU / attention dispatches of every
 * n-th step (the events take the kernel's own begin / end timestamps on the launch stream); 0 switches it off.
 * orca_engine_profile_read waits for the recorded pairs and returns their count and mean duration in ms. */
#define ORCA_PROF_SAMPLE 0
#define ORCA_PROF_GRU 1
#define ORCA_PROF_TGAT 2
#define ORCA_PROF_CLASSES 3
int orca_engine_profile(orca_engine_t* e, int enable);
int orca_engine_profile_read(orca_engine_t* e, int cls, int64_t* launches, double* mean_ms);
int orca_engine_destroy(orca_engine_t* e);
int orca_engine_set_csr(orca_engine_t* e, const orca_csr_t* csr);           /* tgn_model.py:193-195 */
/* hands out n consecutive epochs of the shared `slot` table (for callers that use it between steps) */
int64_t orca_engine_take_epochs(orca_engine_t* e, int64_t n);
/* With world > 1 a train-ordered step is the owner-computes step (it needs an overlay-complete state and the
 * exchange buffers; lo / hi of begin and layer are then bounds in INTERACTIONS, not roots); the caller's sequence is
 *   begin -> [all-gather xl_own -> xl] -> finish -> [all-gather xm_own -> xm] -> layer 2 .. L ([all-gather xl_own -> xl]
 *   between layers) -> apply.
 * finish may be enqueued while the first all-gather is in flight; apply must follow the xm all-gather and precede
 * the next begin.  Eval-ordered steps with world > 1 run on replicated tables: every rank redoes the row writes of
 * the whole batch (the caller makes the replicas coherent first: memory, last_update, messages, msg_ts of an id
 * are only maintained by its owner during owner-computes steps).
 * Builds the Q = 3*bs root queries [src;dst;neg] x [t;t;t]; for this rank's slice [lo,hi) of them: samples
 * the K most recent neighbours, (train) applies GRU(message, memory) for the rows it will read into the
 * overlay, runs attention layer 1 -> emb[0][lo:hi).  embedding_module.py:156-177
 * eidx (the batch's edge ids, tgn_model.py:95) may be NULL; when given, and the slice is the whole batch in
 * train order, the raw-message store of orca_engine_finish is done by the last attention kernel.  The batch
 * arrays must stay valid and unchanged until orca_engine_finish has been enqueued. */
int orca_engine_begin(orca_engine_t* e, const int64_t* src, const int64_t* dst, const int64_t* neg,
                      const int64_t* eidx, const double* ts, int64_t bs, int64_t lo, int64_t hi, int train,
                      void* stream);
/* layer in 2..n_layers: pushes ALL Q rows of the previous layer's output into the layer's history table
 * (callers with several ranks all-gather emb[layer&1] first), then attention over the pulled rows for the
 * slice -> emb[(layer+1)&1][lo:hi).  embedding_module.py:103-113,177-181 */
int orca_engine_layer(orca_engine_t* e, int layer, int64_t lo, int64_t hi, void* stream);
/* Training capture for the steps that follow (train.py:206-237: the forward of a training batch keeps its autograd
 * graph alive and touches no table before loss.backward() has run).  on != 0: the attention calls of
 * orca_engine_begin / orca_engine_layer write what orca_tgat_backward_saved reads into saved[l] (layer l + 1 of the
 * model; orca_tgat_saved_floats(d_e, 3 * bs, K) floats each; entries or the array may be NULL) and apply attention
 * dropout (dropout_p, dropout_seeds[l], query numbering from 0; train.py:30), and the batch's row writes -- memory
 * persist, raw-message store -- are not fused into those kernels: orca_engine_finish performs them, and the caller
 * issues it after the backward has read memory / messages / overlay.  One rank, whole batch, train order, with
 * overlay_complete cleared (the per-batch work-list `worklist` / `count` is what the GRU's backward walks).
 * gru_gates != NULL (d_e = 172): the work-list's GRU runs as orca_gru_update_saving into it
 * ([min(3 * bs * (K + 1), n_nodes)][4][ORCA_D] floats).  on == 0: back to the plain step. */
int orca_engine_train_capture(orca_engine_t* e, int on, float* const* saved, float dropout_p,
                              const uint64_t* dropout_seeds, float* gru_gates);

/* Memory persist for the batch's positives and raw-message build + store, in the train or eval order the
 * batch was begun with.  tgn_model.py:127-145 */
int orca_engine_finish(orca_engine_t* e, const int64_t* src, const int64_t* dst, const int64_t* eidx,
                       const double* ts, void* stream);
/* owner-computes step: the rows per chunk to all-gather in this step (orca_engine_desc: xm): the largest of the owners'
 * counts for the batch begun last, in 64s, at most xm_cap -- the same number on every rank.  begin()'s second kernel
 * leaves the counts in host memory of the engine's own; this call waits for that kernel (a host wait for work enqueued
 * before attention layer 1) and adds them up.  xm_cap when there are no counts (one layer; ORCA_XM_SIZED=0 in the
 * environment, for comparisons). */
int orca_engine_xm_rows(orca_engine_t* e, int64_t* rows);
/* the same for the history-push exchanges of the step (xl): the most push winners any rank has among its roots, in 64s, at
 * most 3*per.  Whichever of the two is called first in a step does the waiting. */
int orca_engine_push_rows(orca_engine_t* e, int64_t* rows);
/* owner-computes step: after the xm all-gather of chunks of `rows` rows (orca_engine_desc), overlay[id] = row and
 * pending[id] = 1 for every id any rank sent */
int orca_engine_apply(orca_engine_t* e, int64_t rows, void* stream);

/* Budget-bounded history for orca_engine_step_budget (north_star: "a GPU-resident index-and-gather with budget-bounded
 * eviction"; replaces the dense [n_nodes, ORCA_D] table of modules/history.py:11 under --budget B).  Device memory,
 * caller-owned:
 *   rows      [1 + 2*plan_cap + q_bound, ORCA_D]  row 0: zeros (an id that was never pushed, the padding id 0); then two
 *                                plan regions of plan_cap rows -- region `cur` holds row i = the cached embedding of
 *                                prev_plan[i], the other one is written by the running step's plan --; then the batch
 *                                region, which the layer-1 attention launch writes directly (the push moves no row)
 *   stamp     int64 [n_nodes]    id -> generation << 21 | kind << 20 | index; kind 0: evaluated in step `generation` at
 *                                position `index` (the last position wins, SURVEY H7: an atomicMax); kind 1: kept by the plan
 *                                of step `generation` as row `index` of the region that step wrote.  Valid for the pulls of
 *                                step g iff generation == g (kind 0) or g - 1 (kind 1); everything else is stale -- nothing
 *                                is ever freed or reset.  Initialise to 0.
 *   prev_plan int64 [plan_cap]   the plan the cached rows follow (whose flags the next step clears)
 *   pull_idx  int32 [3*bs*K]     work buffer: the row of every sampled neighbour of the batch
 * and three host-side words the steps advance (read them back with orca_engine_get_slot_cache): gen (the generation of the
 * next step; start at 2), cur (0 / 1), prev_n (ids in prev_plan).  plan_cap >= every plan_n, q_bound >= 3*bs + cap (< 2^20).
 * With a cache set, a step evaluates into the batch region, pulls through the stamps, makes cache_flag exactly the plan
 * (history.py:25-30) and copies the rows the plan keeps into the other plan region.  It needs a plan (plan_n >= 0): a plan
 * of the reference's planner only keeps ids that were kept before or evaluated in this batch, so every kept id has a row.
 * Differences from the dense table, both outside what the planner produces: a flagged id WITHOUT a row reads zeros (dense:
 * whatever was pushed last, however long ago), and so does the one uncached id that embedding_module.py:212's `[1:]` drops
 * when it is not the padding id.  orca_slot_cache_materialize writes the dense view (rows of the ids the plan keeps; the
 * caller zeroes `dense` first). */
typedef struct {
  float* rows;
  int64_t* stamp;
  int64_t* prev_plan;
  int32_t* pull_idx;
  int64_t plan_cap, q_bound;
  int64_t gen, prev_n;
  int32_t cur;
} orca_slot_cache;
int orca_engine_set_slot_cache(orca_engine_t* e, const orca_slot_cache* cache);    /* NULL or rows == NULL: back to hist[0] */
int orca_engine_get_slot_cache(orca_engine_t* e, orca_slot_cache* cache);          /* as the last step left it */
int orca_slot_cache_materialize(const orca_slot_cache* cache, int64_t n_nodes, float* dense, void* stream);

/* The root queries of a batch (model/tgn_model.py:99-102: nodes = [src; dst; neg], timestamps = [t; t; t]) in one launch:
 * nodes int64 [3*bs], qts double [3*bs], nodes32 = the ids as int32; *tmax (device double, may be NULL) = max t. */
int orca_build_roots(const int64_t* src, const int64_t* dst, const int64_t* neg, const double* ts, int64_t bs,
                     int64_t* nodes, double* qts, int32_t* nodes32, double* tmax, void* stream);
/* One whole batch step of Path C (modules/embedding_module.py:215-285: --reuse with --budget B; two layers, one rank)
 * issued from C++ with no host round trip: roots -> their neighbours (nbr_top .. dt_top [3*bs*K]) -> the neighbours
 * without a cached layer-1 embedding (embedding_module.py:200-212; `unc` [cap], *unc_count on the device) -> layer 1
 * over [roots; uncached] (a launch sized for 3*bs + cap queries that evaluates 3*bs + *unc_count of them) -> history
 * push, cache flags (history.py:25-30: the plan's ids when plan_n >= 0, else every evaluated id) -> layer 2 over the
 * roots -> memory persist + raw-message store (tgn_model.py:127-145, order by `train`).  out [3*bs, ORCA_D].  Uses the
 * tables and work buffers of the engine's descriptor (3*bs + cap <= q_max; `winners` is required, `eager_ids` /
 * `eager_count` enable the eager overlay below); the history is hist[0] or, when one is set, the slot cache above.
 * cache_flag uint8 [n_nodes]; bitmap as in orca_find_uncached; unc_count device int [2] (word 0 = the count, word 1
 * scratch); tmax device double [1].  More uncached ids than `cap` set bit 1 of *err (device int, may be NULL): the step then
 * evaluated only the first `cap` of them.  Train order follows the eager-overlay protocol of orca_engine_begin (below): while
 * the state is overlay-complete the step applies the GRU to the ids the previous step stored and builds no work-list;
 * otherwise it builds the work-list of the ids layer 1 reads.  The memory persist rides in the layer-1 attention launch
 * and the message store in the layer-2 launch; evaluation order runs them, and the positives' GRU, after the embedding. */
int orca_engine_step_budget(orca_engine_t* e, const int64_t* src, const int64_t* dst, const int64_t* neg,
                            const double* ts, const int64_t* eidx, int64_t bs, int train, const int64_t* plan,
                            int64_t plan_n, uint8_t* cache_flag, uint32_t* bitmap, int64_t* unc, int64_t cap,
                            int32_t* unc_count, double* tmax, int32_t* nbr_top, int32_t* eid_top, float* et_top,
                            float* dt_top, float* out, int32_t* err, void* stream);
/* Eager overlay.  While the state is "overlay complete" -- overlay[id] == GRU(messages[id], memory[id]) for every id
 * with a pending message, save those the last step stored -- a train-ordered orca_engine_begin over the whole
 * batch needs no per-batch work-list: it applies the GRU to the ids the previous step stored and nothing else.
 * A freshly zeroed state is complete; so is any state right after update_memory_in_test (no pending ids).  The
 * driver clears the flag itself when a step leaves pending ids it did not list; a caller that stores messages,
 * restores a backup or otherwise changes memory / messages / pending behind the driver's back must clear it
 * (complete = 0: also forgets the outstanding list).  orca_engine_drain applies the outstanding list now (before
 * the handle is destroyed and rebuilt, for instance). */
int orca_engine_set_overlay_complete(orca_engine_t* e, int complete);
int orca_engine_get_overlay_complete(orca_engine_t* e);
int orca_engine_drain(orca_engine_t* e, void* stream);

/* ---- offline cache planner: utils/cache.py:19-89 (read_batches) + :92-166 (MRD_numba) ------- */

/* HOST function (all pointers are host memory).  For the E training interactions (src, dst) cut into
 * batches of bs, and the sampled neighbours of every src / dst query (nbr_src, nbr_dst: int32 [E, K],
 * from orca_sample_recent at the interaction's own timestamp), writes per batch b the ids to keep
 * cached: plan_out[b*budget .. b*budget + plan_len[b]).  stats[0..1] = n_reuse, n_recompute
 * (cache.py:115,119); *avg_reuse_distance as cache.py:165.  Equal reuse distances are ordered by node
 * id (the reference's unstable argsort leaves that order undefined: SURVEY H11). */
int orca_mrd_plan_host(int64_t n_interactions, int64_t bs, int K, const int64_t* src_host, const int64_t* dst_host,
                       const int32_t* nbr_src_host, const int32_t* nbr_dst_host, int64_t n_nodes, int64_t budget,
                       int64_t* plan_out_host, int64_t* plan_len_host, int64_t* stats_host,
                       double* avg_reuse_distance_host);

/* The same plan computed ON THE DEVICE (csrc/plan_dev.hip): src, dst, nbr_src, nbr_dst, plan_out [num_batch * budget] and
 * plan_len [num_batch] are DEVICE arrays (the neighbours can stay where orca_sample_recent wrote them); stats_host [2]
 * and avg_reuse_distance_host are host scalars filled before the call returns (it synchronises the stream and
 * allocates its temporaries with hipMalloc).  Pass 1 is radix sorts over all batches at once, pass 2 one workgroup
 * walking the batches with an LDS sort of each batch's candidates; budget and the candidates of one batch
 * (budget + uncached out-of-batch neighbours + targets) are limited to 16384 (ORCA_EINVAL beyond: use the host
 * planner).  Entry for entry the plan of orca_mrd_plan_host. */
int orca_mrd_plan_device(int64_t n_interactions, int64_t bs, int K, const int64_t* src, const int64_t* dst,
                         const int32_t* nbr_src, const int32_t* nbr_dst, int64_t n_nodes, int64_t budget,
                         int64_t* plan_out, int64_t* plan_len, int64_t* stats_host, double* avg_reuse_distance_host,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORCA_HIP_H */
