// Backward of the GRU memory updater: torch autograd of nn.GRUCell as modules/memory_updater.py:86 calls it
// from get_updated_memory (:56-78), the only place the training loss reaches it (train.py:236-237).
// First cut in the style of tgat_bwd.hip: rows gathered into the workspace, the gate pre-activations recomputed
// and all contractions through the generic strided MFMA GEMM, one elementwise kernel for the gate derivatives.
#include "common.h"

namespace {
constexpr int D = ORCA_D, G3 = 3 * ORCA_D;

// dst[r][0..w) = table[ids ? ids[r] : r][0..w)
__global__ void k_gather_wide(const float* __restrict__ table, const int32_t* __restrict__ ids, int64_t n, int w,
                              float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * w) return;
  const int64_t r = i / w;
  dst[i] = table[(int64_t)(ids ? ids[r] : r) * w + i % w];
}

// In: gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh [n,300] (gate order r, z, n), h [n,100], d_out [n,100].
// Out (in place): gi <- d gi, gh <- d gh;  dh_direct [n,100] = d_out * z (the blend's direct path), may be NULL.
__global__ void k_gru_gates_bwd(float* __restrict__ gi, float* __restrict__ gh, const float* __restrict__ h,
                                const float* __restrict__ d_out, int64_t n, float* __restrict__ dh_direct) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  const int64_t r_ = i / D;
  const int u = (int)(i % D);
  float* a = gi + r_ * G3;
  float* b = gh + r_ * G3;
  const float r = 1.0f / (1.0f + expf(-(a[u] + b[u])));
  const float z = 1.0f / (1.0f + expf(-(a[D + u] + b[D + u])));
  const float ghn = b[2 * D + u];
  const float nn = tanhf(a[2 * D + u] + r * ghn);
  const float go = d_out[i];
  const float dn = go * (1.0f - z), dz = go * (h[i] - nn);
  const float dpn = dn * (1.0f - nn * nn);
  const float dpr = dpn * ghn * r * (1.0f - r);
  const float dpz = dz * z * (1.0f - z);
  a[u] = dpr;          b[u] = dpr;
  a[D + u] = dpz;      b[D + u] = dpz;
  a[2 * D + u] = dpn;  b[2 * D + u] = dpn * r;
  if (dh_direct) dh_direct[i] = go * z;
}
// the same from the gates the forward kept (orca_gru_update_saving: g = [n][r, z, n, h_n][100]): no recompute of the
// pre-activations.  Writes d gi, d gh [n,300].
__global__ void k_gru_gates_bwd_saved(const float* __restrict__ g, const float* __restrict__ h, const float* __restrict__ d_out,
                                      int64_t n, float* __restrict__ gi, float* __restrict__ gh, float* __restrict__ dh_direct) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  const int64_t r_ = i / D;
  const int u = (int)(i % D);
  const float* gr = g + r_ * 4 * D + u;
  const float r = gr[0], z = gr[D], nn = gr[2 * D], ghn = gr[3 * D];
  float* a = gi + r_ * G3;
  float* b = gh + r_ * G3;
  const float go = d_out[i];
  const float dn = go * (1.0f - z), dz = go * (h[i] - nn);
  const float dpn = dn * (1.0f - nn * nn);
  const float dpr = dpn * ghn * r * (1.0f - r);
  const float dpz = dz * z * (1.0f - z);
  a[u] = dpr;          b[u] = dpr;
  a[D + u] = dpz;      b[D + u] = dpz;
  a[2 * D + u] = dpn;  b[2 * D + u] = dpn * r;
  if (dh_direct) dh_direct[i] = go * z;
}
// the two gathers and k_gru_gates_bwd_saved as one launch (the saved-gates route: nothing lies between them): thread
// (r, c) copies x[ids[r]][c], and for c < 100 also h[ids[r]][c] and turns the kept gates of (r, c) into d gi, d gh
__global__ void k_gru_bwd_prep_saved(const float* __restrict__ x, const float* __restrict__ h, const int32_t* __restrict__ ids,
                                     int64_t n, int w, const float* __restrict__ g, const float* __restrict__ d_out,
                                     float* __restrict__ xg, float* __restrict__ hg, float* __restrict__ gi, float* __restrict__ gh,
                                     float* __restrict__ dh_direct) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * w) return;
  const int64_t r_ = i / w;
  const int u = (int)(i - r_ * w);
  const int64_t id = ids ? ids[r_] : r_;
  xg[i] = x[id * w + u];
  if (u >= D) return;
  const float hv = h[id * D + u];
  hg[r_ * D + u] = hv;
  const float* gr = g + r_ * 4 * D + u;
  const float r = gr[0], z = gr[D], nn = gr[2 * D], ghn = gr[3 * D];
  float* a = gi + r_ * G3;
  float* b = gh + r_ * G3;
  const float go = d_out[r_ * D + u];
  const float dn = go * (1.0f - z), dz = go * (hv - nn);
  const float dpn = dn * (1.0f - nn * nn);
  const float dpr = dpn * ghn * r * (1.0f - r);
  const float dpz = dz * z * (1.0f - z);
  a[u] = dpr;          b[u] = dpr;
  a[D + u] = dpz;      b[D + u] = dpz;
  a[2 * D + u] = dpn;  b[2 * D + u] = dpn * r;
  if (dh_direct) dh_direct[r_ * D + u] = go * z;
}
}  // namespace

namespace { constexpr int GRU_SPLITS = 16; }   // split-K pieces of the two weight-gradient GEMMs (one launch)

ORCA_API int64_t orca_gru_backward_workspace_floats(int msg_dim, int64_t n) {
  if (msg_dim < 1 || msg_dim > 3 * ORCA_D + ORCA_MAX_DE || n < 0) return -1;
  return n * (msg_dim + D + 2 * G3) + 256 + (int64_t)GRU_SPLITS * G3 * (msg_dim + 1 + D + 1) + 128;
}

ORCA_API int orca_gru_backward(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                               const float* bias_hh, const int32_t* ids, int64_t n, const float* x, const float* h,
                               const float* d_out, float* g_weight_ih, float* g_weight_hh, float* g_bias_ih,
                               float* g_bias_hh, float* d_x, float* d_h, float* workspace, void* stream_) {
  return orca_gru_backward_phase(msg_dim, weight_ih, weight_hh, bias_ih, bias_hh, ids, n, x, h, d_out, g_weight_ih, g_weight_hh,
                                 g_bias_ih, g_bias_hh, d_x, d_h, workspace, 3, stream_);
}

// phases: 1 = the gathers, the gate derivatives and the input gradients; 2 = the parameter gradients (reads what phase 1
// left in the workspace: a caller may issue it later, on another stream); 3 = both
ORCA_API int orca_gru_backward_phase(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                                     const float* bias_hh, const int32_t* ids, int64_t n, const float* x, const float* h,
                                     const float* d_out, float* g_weight_ih, float* g_weight_hh, float* g_bias_ih,
                                     float* g_bias_hh, float* d_x, float* d_h, float* workspace, int phases, void* stream_) {
  return orca_gru_backward_saved(msg_dim, weight_ih, weight_hh, bias_ih, bias_hh, ids, n, x, h, d_out, g_weight_ih, g_weight_hh,
                                 g_bias_ih, g_bias_hh, d_x, d_h, workspace, nullptr, phases, stream_);
}

// gates != NULL: rows r of what orca_gru_update_saving kept for the same id list (row r = ids[r]); the gate
// pre-activations are then not recomputed (no [n x msg_dim] x [msg_dim x 300] product in the backward)
ORCA_API int orca_gru_backward_saved(int msg_dim, const float* weight_ih, const float* weight_hh, const float* bias_ih,
                                     const float* bias_hh, const int32_t* ids, int64_t n, const float* x, const float* h,
                                     const float* d_out, float* g_weight_ih, float* g_weight_hh, float* g_bias_ih,
                                     float* g_bias_hh, float* d_x, float* d_h, float* workspace, const float* gates,
                                     int phases, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(phases >= 1 && phases <= 3, "orca_gru_backward_phase: phases=%d", phases);
  ORCA_REQUIRE(n >= 0, "orca_gru_backward: negative row count");
  if (n == 0) return ORCA_OK;
  ORCA_REQUIRE(msg_dim >= 1 && msg_dim <= 3 * ORCA_D + ORCA_MAX_DE, "orca_gru_backward: msg_dim=%d out of range", msg_dim);
  ORCA_REQUIRE(n < (1LL << 31) / 1024, "orca_gru_backward: too many rows for one call");
  ORCA_REQUIRE(weight_ih && weight_hh && bias_ih && bias_hh && x && h && d_out && workspace, "orca_gru_backward: null pointer");
  ORCA_REQUIRE(g_weight_ih && g_weight_hh && g_bias_ih && g_bias_hh, "orca_gru_backward: null gradient buffer");
  const int M = (int)n;
  float* xg = workspace;
  float* hg = xg + n * msg_dim;
  float* gi = hg + n * D;
  float* gh = gi + n * G3;
  float* part = gh + n * G3;
  part += (64 - ((part - workspace) & 63)) & 63;
  auto ew = [&](int64_t k) { return (unsigned)ceil_div64(k, 256); };
  (void)hipGetLastError();      // (a status left behind by an earlier runtime call is not this call's)
  int rc = ORCA_OK;
  const bool prep = (phases & 1) && gates && msg_dim >= D;      // one launch for the gathers and the gate derivatives
  if (prep) {
    k_gru_bwd_prep_saved<<<ew(n * msg_dim), 256, 0, s>>>(x, h, ids, n, msg_dim, gates, d_out, xg, hg, gi, gh, d_h);
  } else if (phases & 1) {
    k_gather_wide<<<ew(n * msg_dim), 256, 0, s>>>(x, ids, n, msg_dim, xg);
    k_gather_wide<<<ew(n * D), 256, 0, s>>>(h, ids, n, D, hg);
  }
  auto G1 = [&](int M_, int N_, int K_, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                float* C, int64_t scm, int64_t scn, const float* bias, int accumulate) {
    return OrcaGemm{M_, N_, K_, A, sam, sak, B, sbk, sbn, C, scm, scn, bias, 1.f, accumulate, 0, nullptr, nullptr, 0, nullptr, 1};
  };
  if ((phases & 1) && !gates) {   // gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh: one launch
    OrcaGemm g[2] = {G1(M, G3, msg_dim, xg, msg_dim, 1, weight_ih, 1, msg_dim, gi, G3, 1, bias_ih, 0),
                     G1(M, G3, D, hg, D, 1, weight_hh, 1, D, gh, G3, 1, bias_hh, 0)};
    rc |= orca_sgemm_group(s, g, 2, nullptr);
  }
  if (prep) {
  } else if ((phases & 1) && gates) k_gru_gates_bwd_saved<<<ew(n * D), 256, 0, s>>>(gates, hg, d_out, n, gi, gh, d_h);
  else if (phases & 1) k_gru_gates_bwd<<<ew(n * D), 256, 0, s>>>(gi, gh, hg, d_out, n, d_h);
  if (phases & 2) {   // weight_ih [300,msg] += dgi^T x;  weight_hh [300,100] += dgh^T h;  the biases as a column of ones: one launch
    OrcaGemm g[2] = {G1(G3, msg_dim, M, gi, 1, G3, xg, msg_dim, 1, g_weight_ih, msg_dim, 1, nullptr, 1),
                     G1(G3, D, M, gh, 1, G3, hg, D, 1, g_weight_hh, D, 1, nullptr, 1)};
    g[0].ones_out = g_bias_ih;
    g[1].ones_out = g_bias_hh;
    g[0].splits = g[1].splits = GRU_SPLITS;
    rc |= orca_sgemm_group(s, g, 2, part);
  }
  if ((phases & 1) && (d_x || d_h)) {
    OrcaGemm g[2];
    int ng = 0;
    if (d_x) g[ng++] = G1(M, msg_dim, G3, gi, G3, 1, weight_ih, msg_dim, 1, d_x, msg_dim, 1, nullptr, 0);
    if (d_h) g[ng++] = G1(M, D, G3, gh, G3, 1, weight_hh, D, 1, d_h, D, 1, nullptr, 1);
    rc |= orca_sgemm_group(s, g, ng, nullptr);
  }
  if (rc != ORCA_OK) return ORCA_EHIP;      // (orca_last_error keeps the failing callee's own message)
  if (hipGetLastError() != hipSuccess) {
    orca_set_error("orca_gru_backward: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}
