// The link decoder of a training step and its loss: model/tgn_model.py:150-157 (affinity_score = MergeLayer,
// utils/util.py:12-24: fc2(relu(fc1([x1 | x2])))) applied to (src, dst) and (src, neg), the sigmoid, and
// train.py:231-233: BCE(pos, 1) + BCE(neg, 0) -- forward AND backward in five launches.  Under torch autograd the
// same thing is ~25 launches and 0.3 ms of host time per step, a fifth of what is left of the training step.
// Not the hot path's arithmetic (two tiny dense layers); it is here because it sits between the path's forward and
// its backward.
//   fc1 is split by input half: U_a = src W1[:, :100]^T (once per interaction: the positive and the negative score
//   share it), U_b = [dst; neg] W1[:, 100:]^T;  h_pos = relu(U_a + U_b[dst] + b1), h_neg = relu(U_a + U_b[neg] + b1).
#include "common.h"

namespace {
constexpr int D = ORCA_D;
constexpr int DEC_SPLITS = 16;

// one wave per interaction i: scores, loss terms, and everything the backward GEMMs consume
//   U [3bs, D]: rows [U_a(src) ; U_b(dst) ; U_b(neg)];  H [2bs, D] = [h_pos ; h_neg];  dS [2bs] = [dL/ds_pos ; dL/ds_neg]
//   dU [3bs, D]: rows [dz_pos + dz_neg ; dz_pos ; dz_neg] (dz = dL/d pre-activation of fc1);  lossv [bs]
__global__ __launch_bounds__(256) void k_decoder_mid(const float* __restrict__ U, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2, int64_t bs,
                                                     float* __restrict__ H, float* __restrict__ dS, float* __restrict__ dU,
                                                     float* __restrict__ lossv, float* __restrict__ prob) {
  const int lane = threadIdx.x & 63, c1 = lane + 64;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= bs) return;
  const bool hi = c1 < D;
  const float ba = b1[lane], bb = hi ? b1[c1] : 0.f, wa = w2[lane], wb = hi ? w2[c1] : 0.f;
  const float ua0 = U[i * D + lane], ua1 = hi ? U[i * D + c1] : 0.f;
  float hp0 = fmaxf(ua0 + U[(bs + i) * D + lane] + ba, 0.f), hp1 = hi ? fmaxf(ua1 + U[(bs + i) * D + c1] + bb, 0.f) : 0.f;
  float hn0 = fmaxf(ua0 + U[(2 * bs + i) * D + lane] + ba, 0.f), hn1 = hi ? fmaxf(ua1 + U[(2 * bs + i) * D + c1] + bb, 0.f) : 0.f;
  const float sp = wave_sum(hp0 * wa + hp1 * wb) + b2[0], sn = wave_sum(hn0 * wa + hn1 * wb) + b2[0];
  const float pp = 1.0f / (1.0f + expf(-sp)), pn = 1.0f / (1.0f + expf(-sn));
  // torch's binary_cross_entropy clamps the logarithms at -100
  const float lp = fmaxf(logf(pp), -100.f), ln = fmaxf(logf(1.0f - pn), -100.f);
  const float inv = 1.0f / (float)bs;
  // d BCE / d s through the sigmoid: (p - y) / bs  (zero where the clamp is active is a measure-zero corner: ignored)
  const float dsp = (pp - 1.0f) * inv, dsn = pn * inv;
  if (lane == 0) {
    lossv[i] = -(lp + ln);
    dS[i] = dsp;
    dS[bs + i] = dsn;
    if (prob) { prob[i] = pp; prob[bs + i] = pn; }
  }
  const float zp0 = hp0 > 0.f ? dsp * wa : 0.f, zn0 = hn0 > 0.f ? dsn * wa : 0.f;
  H[i * D + lane] = hp0;
  H[(bs + i) * D + lane] = hn0;
  dU[i * D + lane] = zp0 + zn0;
  dU[(bs + i) * D + lane] = zp0;
  dU[(2 * bs + i) * D + lane] = zn0;
  if (hi) {
    const float zp1 = hp1 > 0.f ? dsp * wb : 0.f, zn1 = hn1 > 0.f ? dsn * wb : 0.f;
    H[i * D + c1] = hp1;
    H[(bs + i) * D + c1] = hn1;
    dU[i * D + c1] = zp1 + zn1;
    dU[(bs + i) * D + c1] = zp1;
    dU[(2 * bs + i) * D + c1] = zn1;
  }
}

struct Ws { int64_t U, H, dS, dU, lossv, part, total; };
Ws ws_layout(int64_t bs) {
  Ws w{};
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += (n + 63) / 64 * 64; return r; };
  w.U = take(3 * bs * D); w.H = take(2 * bs * D); w.dS = take(2 * bs); w.dU = take(3 * bs * D); w.lossv = take(bs);
  w.part = o;
  // four split-K problems (D x (D+1), D x D, 1 x (D+1), 1 x 1)
  o += (int64_t)DEC_SPLITS * ((int64_t)D * (D + 1) + (int64_t)D * D + (D + 1) + 1) + 64 * 4;
  w.total = o;
  return w;
}
}  // namespace

ORCA_API int64_t orca_decoder_train_workspace_floats(int64_t bs) {
  if (bs < 0) return -1;
  return ws_layout(bs).total;
}

ORCA_API int orca_decoder_train(const float* emb, int64_t bs, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                                const float* fc2_b, float* loss, float* prob, float* d_emb, float* g_fc1_w, float* g_fc1_b,
                                float* g_fc2_w, float* g_fc2_b, float* workspace, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(bs >= 1 && bs < (1LL << 30), "orca_decoder_train: bs=%lld out of range", (long long)bs);
  ORCA_REQUIRE(emb && fc1_w && fc1_b && fc2_w && fc2_b && loss && d_emb && g_fc1_w && g_fc1_b && g_fc2_w && g_fc2_b && workspace,
               "orca_decoder_train: null pointer");
  (void)hipGetLastError();      // (a status left behind by an earlier runtime call is not this call's)
  const Ws w = ws_layout(bs);
  float *U = workspace + w.U, *H = workspace + w.H, *dS = workspace + w.dS, *dU = workspace + w.dU,
        *lossv = workspace + w.lossv, *part = workspace + w.part;
  const int B = (int)bs;
  auto G1 = [&](int M_, int N_, int K_, const float* A, int64_t sam, int64_t sak, const float* Bm, int64_t sbk, int64_t sbn,
                float* C, int64_t scm, int64_t scn) {
    return OrcaGemm{M_, N_, K_, A, sam, sak, Bm, sbk, sbn, C, scm, scn, nullptr, 1.f, 0, 0, nullptr, nullptr, 0, nullptr, 1};
  };
  int rc = ORCA_OK;
  {   // U_a = src W1a^T, U_b = [dst; neg] W1b^T        (fc1_w [D, 2D] row-major: W1a = columns 0..D-1)
    OrcaGemm g[2] = {G1(B, D, D, emb, D, 1, fc1_w, 1, 2 * D, U, D, 1),
                     G1(2 * B, D, D, emb + bs * D, D, 1, fc1_w + D, 1, 2 * D, U + bs * D, D, 1)};
    rc |= orca_sgemm_group(s, g, 2, nullptr);
  }
  k_decoder_mid<<<(unsigned)ceil_div64(bs, 4), 256, 0, s>>>(U, fc1_b, fc2_w, fc2_b, bs, H, dS, dU, lossv, prob);
  {   // d src = dU_a W1a, d [dst; neg] = dU_b W1b
    OrcaGemm g[2] = {G1(B, D, D, dU, D, 1, fc1_w, 2 * D, 1, d_emb, D, 1),
                     G1(2 * B, D, D, dU + bs * D, D, 1, fc1_w + D, 2 * D, 1, d_emb + bs * D, D, 1)};
    rc |= orca_sgemm_group(s, g, 2, nullptr);
  }
  {   // dW1a += dU_a^T src (+ db1 as its ones column), dW1b += dU_b^T [dst; neg], dW2 += dS^T H (+ db2)
    // and the loss = sum of the per-interaction terms / bs: a problem that has only its ones column (deterministic: the
    // split-K pieces are added in a fixed order)
    OrcaGemm g[4] = {G1(D, D, B, dU, 1, D, emb, D, 1, g_fc1_w, 2 * D, 1),
                     G1(D, D, 2 * B, dU + bs * D, 1, D, emb + bs * D, D, 1, g_fc1_w + D, 2 * D, 1),
                     G1(1, D, 2 * B, dS, 1, 1, H, D, 1, g_fc2_w, D, 1),
                     G1(1, 0, B, lossv, 1, 1, lossv, 0, 1, nullptr, 1, 1)};
    for (auto& p : g) { p.accumulate = 1; p.splits = DEC_SPLITS; }
    g[0].ones_out = g_fc1_b;
    g[2].ones_out = g_fc2_b;
    g[3].ones_out = loss; g[3].accumulate = 0; g[3].alpha = 1.0f / (float)bs;
    rc |= orca_sgemm_group(s, g, 4, part);
  }
  if (rc != ORCA_OK) return ORCA_EHIP;      // (orca_last_error keeps the failing callee's own message)
  if (hipGetLastError() != hipSuccess) {
    orca_set_error("orca_decoder_train: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}
