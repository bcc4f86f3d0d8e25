// Time-sorted CSR in HBM + temporal neighbour sampler (replaces the Numba jitclass NeighborFinder,
// utils/util.py:91-151 of the reference).
#include <cstring>
#include <algorithm>
#include <vector>
#include "common.h"
#include "sampler.h"
#include <rocprim/device/device_radix_sort.hpp>

struct orca_csr {
  int device;
  int64_t n_rows;
  int64_t nnz;
  int64_t* indptr;  // [n_rows+1]
  int32_t* nbr;     // [nnz]
  int32_t* eid;     // [nnz]
  double* ts;       // [nnz]
  double* fence_mem;  // the fence levels of ts (sampler.h), one allocation
  orca_fences F;
};

static thread_local char g_err[512] = "";
void orca_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
ORCA_API const char* orca_last_error(void) { return g_err; }
ORCA_API int orca_version(void) { return 1; }

static void csr_free(orca_csr* c) {
  if (!c) return;
  if (c->indptr) (void)hipFree(c->indptr);
  if (c->nbr) (void)hipFree(c->nbr);
  if (c->eid) (void)hipFree(c->eid);
  if (c->ts) (void)hipFree(c->ts);
  if (c->fence_mem) (void)hipFree(c->fence_mem);
  delete c;
}

static int csr_alloc(orca_csr* c) {
  ORCA_HIP_TRY(hipMalloc(&c->indptr, (size_t)(c->n_rows + 1) * sizeof(int64_t)));
  size_t n = (size_t)std::max<int64_t>(c->nnz, 1);
  ORCA_HIP_TRY(hipMalloc(&c->nbr, n * sizeof(int32_t)));
  ORCA_HIP_TRY(hipMalloc(&c->eid, n * sizeof(int32_t)));
  ORCA_HIP_TRY(hipMalloc(&c->ts, n * sizeof(double)));
  return ORCA_OK;
}

// dst[j] = src[stride * j + stride - 1], j < n
__global__ void k_fence_level(const double* __restrict__ src, int64_t stride, double* __restrict__ dst, int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[stride * j + stride - 1];
}
// the fence levels over c->ts (sampler.h): +3 % of the timestamps' bytes
static int csr_build_fences(orca_csr* c, hipStream_t stream) {
  int64_t n[ORCA_FENCE_LEVELS], total = 0;
  int64_t prev = c->nnz;
  for (int k = 0; k < ORCA_FENCE_LEVELS; ++k) {
    n[k] = prev / (k == 0 ? ORCA_FENCE_B0 : 64);
    total += (n[k] + 15) / 16 * 16;
    prev = n[k];
  }
  for (int k = 0; k < ORCA_FENCE_LEVELS; ++k) c->F.lvl[k] = nullptr;
  if (n[0] == 0) return ORCA_OK;
  ORCA_HIP_TRY(hipMalloc(&c->fence_mem, (size_t)total * sizeof(double)));
  const double* src = c->ts;
  double* dst = c->fence_mem;
  for (int k = 0; k < ORCA_FENCE_LEVELS && n[k] > 0; ++k) {
    k_fence_level<<<(unsigned)ceil_div64(n[k], 256), 256, 0, stream>>>(src, k == 0 ? ORCA_FENCE_B0 : 64, dst, n[k]);
    c->F.lvl[k] = dst;
    src = dst;
    dst += (n[k] + 15) / 16 * 16;
  }
  ORCA_HIP_TRY(hipGetLastError());
  return ORCA_OK;
}
// (ORCA_NO_FENCES=1 in the environment: the samplers search the timestamps directly, for A/B measurements)
static orca_fences csr_F(const orca_csr* c) {
  static const bool off = getenv("ORCA_NO_FENCES") && atoi(getenv("ORCA_NO_FENCES")) != 0;
  return off ? orca_fences{} : c->F;
}
int orca_csr_fences(const orca_csr_t* c, orca_fences* out) {
  ORCA_REQUIRE(c && out, "orca_csr_fences: null argument");
  *out = csr_F(c);
  return ORCA_OK;
}

// ---------------------------------------------------------------------------------------------
// host builder: counting sort by node in edge order (src entry, then dst entry: util.py:94-96),
// then a stable per-node sort by timestamp (util.py:101) only where a list is not already sorted.
ORCA_API int orca_csr_create_host(int64_t E, const int64_t* src, const int64_t* dst, const int64_t* eidx,
                                  const double* ts, int64_t n_rows, int device, orca_csr_t** out) {
  ORCA_REQUIRE(out && n_rows > 0 && E >= 0, "orca_csr_create_host: bad arguments");
  ORCA_REQUIRE(E == 0 || (src && dst && eidx && ts), "orca_csr_create_host: null edge arrays");
  std::vector<int64_t> indptr((size_t)n_rows + 1, 0);
  for (int64_t e = 0; e < E; ++e) {
    if (src[e] < 0 || src[e] >= n_rows || dst[e] < 0 || dst[e] >= n_rows) {
      orca_set_error("orca_csr_create_host: node id out of range at edge %lld", (long long)e);
      return ORCA_ERANGE;
    }
    indptr[src[e] + 1]++;
    indptr[dst[e] + 1]++;
  }
  for (int64_t v = 0; v < n_rows; ++v) indptr[v + 1] += indptr[v];
  struct Ent { double t; int32_t nbr, eid; };
  std::vector<Ent> ent((size_t)(2 * E));
  {
    std::vector<int64_t> cur(indptr.begin(), indptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) {
      ent[cur[src[e]]++] = Ent{ts[e], (int32_t)dst[e], (int32_t)eidx[e]};
      ent[cur[dst[e]]++] = Ent{ts[e], (int32_t)src[e], (int32_t)eidx[e]};
    }
  }
  for (int64_t v = 0; v < n_rows; ++v) {
    Ent* b = ent.data() + indptr[v];
    Ent* e = ent.data() + indptr[v + 1];
    if (!std::is_sorted(b, e, [](const Ent& x, const Ent& y) { return x.t < y.t; }))
      std::stable_sort(b, e, [](const Ent& x, const Ent& y) { return x.t < y.t; });
  }
  std::vector<int32_t> h_nbr((size_t)(2 * E)), h_eid((size_t)(2 * E));
  std::vector<double> h_ts((size_t)(2 * E));
  for (int64_t i = 0; i < 2 * E; ++i) { h_nbr[i] = ent[i].nbr; h_eid[i] = ent[i].eid; h_ts[i] = ent[i].t; }

  ORCA_HIP_TRY(hipSetDevice(device));
  orca_csr* c = new orca_csr();
  memset(c, 0, sizeof(*c));
  c->device = device; c->n_rows = n_rows; c->nnz = 2 * E;
  int rc = csr_alloc(c);
  if (rc != ORCA_OK) { csr_free(c); return rc; }
  hipError_t e1 = hipMemcpy(c->indptr, indptr.data(), (size_t)(n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
  hipError_t e2 = E ? hipMemcpy(c->nbr, h_nbr.data(), (size_t)(2 * E) * sizeof(int32_t), hipMemcpyHostToDevice) : hipSuccess;
  hipError_t e3 = E ? hipMemcpy(c->eid, h_eid.data(), (size_t)(2 * E) * sizeof(int32_t), hipMemcpyHostToDevice) : hipSuccess;
  hipError_t e4 = E ? hipMemcpy(c->ts, h_ts.data(), (size_t)(2 * E) * sizeof(double), hipMemcpyHostToDevice) : hipSuccess;
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    csr_free(c);
    orca_set_error("orca_csr_create_host: upload failed");
    return ORCA_EHIP;
  }
  rc = csr_build_fences(c, nullptr);
  if (rc == ORCA_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = ORCA_EHIP;
  if (rc != ORCA_OK) { csr_free(c); return rc; }
  *out = c;
  return ORCA_OK;
}

// ---------------------------------------------------------------------------------------------
// device builder: entry i = (edge i>>1, side i&1) in the reference's append order.  A stable radix
// sort by timestamp (skipped when the edge list is already time-ordered) followed by a stable radix
// sort by owning node reproduces "append in edge order, then stable sort each list by ts".
__global__ void k_check_edges(const int64_t* src, const int64_t* dst, const double* ts, int64_t E, int64_t n_rows,
                              int* flags /* [0]=range error, [1]=unsorted */) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (src[e] < 0 || src[e] >= n_rows || dst[e] < 0 || dst[e] >= n_rows) flags[0] = 1;
  if (e + 1 < E && ts[e + 1] < ts[e]) flags[1] = 1;
}
__global__ void k_entry_ts(const double* ts, int64_t n2, double* key, uint32_t* val) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  key[i] = ts[i >> 1];
  val[i] = (uint32_t)i;
}
__global__ void k_entry_node(const int64_t* src, const int64_t* dst, const uint32_t* perm, int64_t n2, int32_t* key,
                             uint32_t* val) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n2) return;
  uint32_t i = perm ? perm[p] : (uint32_t)p;
  key[p] = (int32_t)((i & 1) ? dst[i >> 1] : src[i >> 1]);
  val[p] = i;
}
__global__ void k_fill_csr(const int64_t* src, const int64_t* dst, const int64_t* eidx, const double* ts,
                           const uint32_t* perm, int64_t n2, int32_t* nbr, int32_t* eid, double* t) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n2) return;
  uint32_t i = perm[p];
  int64_t e = i >> 1;
  nbr[p] = (int32_t)((i & 1) ? src[e] : dst[e]);
  eid[p] = (int32_t)eidx[e];
  t[p] = ts[e];
}
__global__ void k_indptr(const int32_t* sorted_node, int64_t n2, int64_t n_rows, int64_t* indptr) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > n_rows) return;
  int64_t lo = 0, hi = n2;
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if ((int64_t)sorted_node[mid] < v) lo = mid + 1; else hi = mid;
  }
  indptr[v] = lo;
}

ORCA_API int orca_csr_create_device(int64_t E, const int64_t* src, const int64_t* dst, const int64_t* eidx,
                                    const double* ts, int64_t n_rows, void* stream_, orca_csr_t** out) {
  hipStream_t stream = (hipStream_t)stream_;
  ORCA_REQUIRE(out && n_rows > 0 && n_rows < (1LL << 31) && E > 0 && 2 * E < (1LL << 32),
               "orca_csr_create_device: bad sizes (E=%lld n_rows=%lld)", (long long)E, (long long)n_rows);
  int device = 0;
  ORCA_HIP_TRY(hipGetDevice(&device));
  const int64_t n2 = 2 * E;
  const int B = 256;
  int* d_flags = nullptr;
  double *k1a = nullptr, *k1b = nullptr;
  int32_t *k2a = nullptr, *k2b = nullptr;
  uint32_t *va = nullptr, *vb = nullptr;
  void* tmp = nullptr;
  orca_csr* c = nullptr;
  int rc = ORCA_OK;
  auto cleanup = [&]() {
    (void)hipFree(d_flags); (void)hipFree(k1a); (void)hipFree(k1b); (void)hipFree(k2a); (void)hipFree(k2b);
    (void)hipFree(va); (void)hipFree(vb); (void)hipFree(tmp);
  };
#define TRY_OR_FAIL(expr)                                                                     \
  do { hipError_t _e = (expr); if (_e != hipSuccess) {                                         \
      orca_set_error("%s failed: %s", #expr, hipGetErrorString(_e)); cleanup(); csr_free(c); return ORCA_EHIP; } } while (0)

  TRY_OR_FAIL(hipMalloc(&d_flags, 2 * sizeof(int)));
  TRY_OR_FAIL(hipMemsetAsync(d_flags, 0, 2 * sizeof(int), stream));
  k_check_edges<<<(unsigned)ceil_div64(E, B), B, 0, stream>>>(src, dst, ts, E, n_rows, d_flags);
  int h_flags[2] = {0, 0};
  TRY_OR_FAIL(hipMemcpyAsync(h_flags, d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, stream));
  TRY_OR_FAIL(hipStreamSynchronize(stream));
  if (h_flags[0]) {
    orca_set_error("orca_csr_create_device: node id out of range");
    cleanup();
    return ORCA_ERANGE;
  }
  TRY_OR_FAIL(hipMalloc(&va, (size_t)n2 * sizeof(uint32_t)));
  TRY_OR_FAIL(hipMalloc(&vb, (size_t)n2 * sizeof(uint32_t)));
  uint32_t* perm1 = nullptr;  // entry order after the time sort (nullptr = identity)
  if (h_flags[1]) {
    TRY_OR_FAIL(hipMalloc(&k1a, (size_t)n2 * sizeof(double)));
    TRY_OR_FAIL(hipMalloc(&k1b, (size_t)n2 * sizeof(double)));
    k_entry_ts<<<(unsigned)ceil_div64(n2, B), B, 0, stream>>>(ts, n2, k1a, va);
    size_t tb = 0;
    TRY_OR_FAIL(rocprim::radix_sort_pairs(nullptr, tb, k1a, k1b, va, vb, (size_t)n2, 0, 64, stream));
    TRY_OR_FAIL(hipMalloc(&tmp, tb));
    TRY_OR_FAIL(rocprim::radix_sort_pairs(tmp, tb, k1a, k1b, va, vb, (size_t)n2, 0, 64, stream));
    TRY_OR_FAIL(hipStreamSynchronize(stream));
    (void)hipFree(tmp); tmp = nullptr;
    (void)hipFree(k1a); k1a = nullptr;
    (void)hipFree(k1b); k1b = nullptr;
    perm1 = vb;
  }
  TRY_OR_FAIL(hipMalloc(&k2a, (size_t)n2 * sizeof(int32_t)));
  TRY_OR_FAIL(hipMalloc(&k2b, (size_t)n2 * sizeof(int32_t)));
  uint32_t* v_in = perm1 ? va : va;   // k_entry_node writes its values into `va`
  uint32_t* v_out = perm1 ? nullptr : vb;
  if (perm1) {
    // perm1 lives in vb; write node keys + carried entry ids into va, sort into a fresh buffer
    k_entry_node<<<(unsigned)ceil_div64(n2, B), B, 0, stream>>>(src, dst, perm1, n2, k2a, va);
    TRY_OR_FAIL(hipStreamSynchronize(stream));
    v_out = vb;  // perm1 no longer needed once va holds the carried ids
  } else {
    k_entry_node<<<(unsigned)ceil_div64(n2, B), B, 0, stream>>>(src, dst, nullptr, n2, k2a, va);
  }
  int end_bit = 1;
  while ((1LL << end_bit) < n_rows) ++end_bit;
  {
    size_t tb = 0;
    TRY_OR_FAIL(rocprim::radix_sort_pairs(nullptr, tb, k2a, k2b, v_in, v_out, (size_t)n2, 0, end_bit, stream));
    TRY_OR_FAIL(hipMalloc(&tmp, tb));
    TRY_OR_FAIL(rocprim::radix_sort_pairs(tmp, tb, k2a, k2b, v_in, v_out, (size_t)n2, 0, end_bit, stream));
  }
  c = new orca_csr();
  memset(c, 0, sizeof(*c));
  c->device = device; c->n_rows = n_rows; c->nnz = n2;
  rc = csr_alloc(c);
  if (rc != ORCA_OK) { cleanup(); csr_free(c); return rc; }
  k_fill_csr<<<(unsigned)ceil_div64(n2, B), B, 0, stream>>>(src, dst, eidx, ts, v_out, n2, c->nbr, c->eid, c->ts);
  k_indptr<<<(unsigned)ceil_div64(n_rows + 1, B), B, 0, stream>>>(k2b, n2, n_rows, c->indptr);
  TRY_OR_FAIL(hipGetLastError());
  rc = csr_build_fences(c, stream);
  if (rc != ORCA_OK) { cleanup(); csr_free(c); return rc; }
  TRY_OR_FAIL(hipStreamSynchronize(stream));
  cleanup();
#undef TRY_OR_FAIL
  *out = c;
  return ORCA_OK;
}

ORCA_API int orca_csr_from_arrays_host(int64_t n_rows, const int64_t* indptr, const int32_t* nbr, const int32_t* eidx,
                                       const double* ts, int device, orca_csr_t** out) {
  ORCA_REQUIRE(out && n_rows > 0 && indptr, "orca_csr_from_arrays_host: bad arguments");
  const int64_t nnz = indptr[n_rows];
  ORCA_REQUIRE(nnz >= 0 && indptr[0] == 0 && (nnz == 0 || (nbr && eidx && ts)), "orca_csr_from_arrays_host: bad CSR");
  ORCA_HIP_TRY(hipSetDevice(device));
  orca_csr* c = new orca_csr();
  memset(c, 0, sizeof(*c));
  c->device = device; c->n_rows = n_rows; c->nnz = nnz;
  int rc = csr_alloc(c);
  if (rc != ORCA_OK) { csr_free(c); return rc; }
  hipError_t e1 = hipMemcpy(c->indptr, indptr, (size_t)(n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
  hipError_t e2 = nnz ? hipMemcpy(c->nbr, nbr, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice) : hipSuccess;
  hipError_t e3 = nnz ? hipMemcpy(c->eid, eidx, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice) : hipSuccess;
  hipError_t e4 = nnz ? hipMemcpy(c->ts, ts, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice) : hipSuccess;
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
    csr_free(c);
    orca_set_error("orca_csr_from_arrays_host: upload failed");
    return ORCA_EHIP;
  }
  rc = csr_build_fences(c, nullptr);
  if (rc == ORCA_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = ORCA_EHIP;
  if (rc != ORCA_OK) { csr_free(c); return rc; }
  *out = c;
  return ORCA_OK;
}

ORCA_API int orca_csr_destroy(orca_csr_t* csr) {
  csr_free(csr);
  return ORCA_OK;
}

ORCA_API int orca_csr_arrays(const orca_csr_t* c, int64_t* n_rows, int64_t* nnz, const int64_t** indptr,
                             const int32_t** nbr, const int32_t** eidx, const double** ts) {
  ORCA_REQUIRE(c, "orca_csr_arrays: null handle");
  if (n_rows) *n_rows = c->n_rows;
  if (nnz) *nnz = c->nnz;
  if (indptr) *indptr = c->indptr;
  if (nbr) *nbr = c->nbr;
  if (eidx) *eidx = c->eid;
  if (ts) *ts = c->ts;
  return ORCA_OK;
}

ORCA_API int orca_csr_copy_host(const orca_csr_t* c, int64_t* indptr, int32_t* nbr, int32_t* eidx, double* ts) {
  ORCA_REQUIRE(c, "orca_csr_copy_host: null handle");
  if (indptr) ORCA_HIP_TRY(hipMemcpy(indptr, c->indptr, (size_t)(c->n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (c->nnz == 0) return ORCA_OK;
  if (nbr) ORCA_HIP_TRY(hipMemcpy(nbr, c->nbr, (size_t)c->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (eidx) ORCA_HIP_TRY(hipMemcpy(eidx, c->eid, (size_t)c->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (ts) ORCA_HIP_TRY(hipMemcpy(ts, c->ts, (size_t)c->nnz * sizeof(double), hipMemcpyDeviceToHost));
  return ORCA_OK;
}

ORCA_API int orca_csr_copy_range_host(const orca_csr_t* c, int64_t begin, int64_t end, int32_t* nbr, int32_t* eidx,
                                      double* ts) {
  ORCA_REQUIRE(c && 0 <= begin && begin <= end && end <= c->nnz, "orca_csr_copy_range_host: bad range");
  const size_t n = (size_t)(end - begin);
  if (n == 0) return ORCA_OK;
  if (nbr) ORCA_HIP_TRY(hipMemcpy(nbr, c->nbr + begin, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (eidx) ORCA_HIP_TRY(hipMemcpy(eidx, c->eid + begin, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (ts) ORCA_HIP_TRY(hipMemcpy(ts, c->ts + begin, n * sizeof(double), hipMemcpyDeviceToHost));
  return ORCA_OK;
}

__global__ __launch_bounds__(256) void k_sample_recent(const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ nbr,
                                                       const int32_t* __restrict__ eid,
                                                       const double* __restrict__ ts, orca_fences F, int64_t n_rows,
                                                       const int64_t* __restrict__ nodes,
                                                       const double* __restrict__ qts, int64_t Q, int K,
                                                       int32_t* __restrict__ o_nbr, int32_t* __restrict__ o_eid,
                                                       float* __restrict__ o_et, float* __restrict__ o_dt,
                                                       int* err_flag, orca_claim claim) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= Q) return;
  sample_recent_wave(indptr, nbr, eid, ts, F, n_rows, nodes[q], qts[q], q, Q, K, o_nbr, o_eid, o_et, o_dt, err_flag, claim, lane);
}

int orca_sample_recent_claim(const orca_csr_t* c, const int64_t* nodes, const double* ts, int64_t Q, int K,
                             int32_t* out_nbr, int32_t* out_eidx, float* out_et, float* out_dt, int* err_flag,
                             const orca_claim* claim, void* stream) {
  ORCA_REQUIRE(c && Q >= 0 && K >= 1, "orca_sample_recent: bad arguments (Q=%lld K=%d)", (long long)Q, K);
  if (Q == 0) return ORCA_OK;
  ORCA_REQUIRE(nodes && ts && out_nbr && out_eidx && out_et, "orca_sample_recent: null pointer");
  const int wpb = 4;
  ORCA_LAUNCH_TIMED(k_sample_recent, (unsigned)ceil_div64(Q, wpb), wpb * 64, 0, (hipStream_t)stream,
                    c->indptr, c->nbr, c->eid, c->ts, csr_F(c), c->n_rows, nodes, ts, Q, K, out_nbr, out_eidx, out_et, out_dt,
                    err_flag, claim ? *claim : orca_claim{});
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_sample_recent(const orca_csr_t* c, const int64_t* nodes, const double* ts, int64_t Q, int K,
                                int32_t* out_nbr, int32_t* out_eidx, float* out_et, float* out_dt, int* err_flag,
                                void* stream) {
  return orca_sample_recent_claim(c, nodes, ts, Q, K, out_nbr, out_eidx, out_et, out_dt, err_flag, nullptr, stream);
}

__global__ __launch_bounds__(256) void k_find_before(const int64_t* __restrict__ indptr, const double* __restrict__ ts,
                                                     orca_fences F, int64_t n_rows, const int64_t* __restrict__ nodes,
                                                     const double* __restrict__ qts, int64_t Q, int64_t* out) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int64_t v = nodes[q];
  int64_t cnt = 0;
  if (v >= 0 && v < n_rows) {
    int64_t lo = indptr[v];
    cnt = wave_lower_bound(ts, F, lo, indptr[v + 1], qts[q], lane) - lo;
  }
  if (lane == 0) out[q] = cnt;
}

ORCA_API int orca_find_before(const orca_csr_t* c, const int64_t* nodes, const double* ts, int64_t Q, int64_t* out,
                              void* stream) {
  ORCA_REQUIRE(c && Q >= 0, "orca_find_before: bad arguments");
  if (Q == 0) return ORCA_OK;
  k_find_before<<<(unsigned)ceil_div64(Q, 4), 256, 0, (hipStream_t)stream>>>(c->indptr, c->ts, csr_F(c), c->n_rows, nodes, ts, Q, out);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

// Uniform sampling: sample_uniform_wave (sampler.h), one wave per query
__global__ __launch_bounds__(256) void k_sample_uniform(const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ nbr,
                                                        const int32_t* __restrict__ eid,
                                                        const double* __restrict__ ts, orca_fences F, int64_t n_rows,
                                                        const int64_t* __restrict__ nodes,
                                                        const double* __restrict__ qts, int64_t Q, int K,
                                                        uint64_t seed, int64_t q0, int32_t* __restrict__ o_nbr,
                                                        int32_t* __restrict__ o_eid, float* __restrict__ o_et,
                                                        float* __restrict__ o_dt, int* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= Q) return;
  sample_uniform_wave(indptr, nbr, eid, ts, F, n_rows, nodes[q], qts[q], q, Q, K, seed, q0 + q, o_nbr, o_eid, o_et, o_dt,
                      err_flag, orca_claim{}, lane);
}

ORCA_API int orca_sample_uniform(const orca_csr_t* c, const int64_t* nodes, const double* ts, int64_t Q, int K,
                                 uint64_t seed, int64_t q0, int32_t* out_nbr, int32_t* out_eidx, float* out_et,
                                 float* out_dt, int* err_flag, void* stream) {
  ORCA_REQUIRE(c && Q >= 0 && K >= 1 && K <= 64 && q0 >= 0, "orca_sample_uniform: bad arguments (Q=%lld K=%d)", (long long)Q, K);
  if (Q == 0) return ORCA_OK;
  ORCA_LAUNCH_TIMED(k_sample_uniform, (unsigned)ceil_div64(Q, 4), 256, 0, (hipStream_t)stream,
                    c->indptr, c->nbr, c->eid, c->ts, csr_F(c), c->n_rows, nodes, ts, Q, K, seed, q0, out_nbr, out_eidx, out_et,
                    out_dt, err_flag);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}
