// Fused TGAT attention layer (model/temporal_attention.py:24-56 of the reference) for gfx950.
//
// One workgroup (8 waves) owns TQ = 16 queries and runs the whole layer on them without touching
// HBM in between: gather -> time encoding -> projections on v_mfma_f32_16x16x4_f32 -> masked
// softmax over the K neighbours with wave shuffles -> output projection -> MergeLayer -> LayerNorm.
//
// Algebra.  nn.MultiheadAttention with one query per K keys computes, per head h,
//   s_j = (0.1*q_h) . (Wk_h x_j + bk_h),  a = softmax_j(s),  o_h = sum_j a_j (Wv_h x_j + bv_h).
// Both K-sized projections are folded through the linear maps:
//   s_j = (Wk_h^T (0.1*q_h)) . x_j + (0.1*q_h).bk_h          o_h = Wv_h (sum_j a_j x_j) + bv_h
// which is the same function (sum_j a_j = 1) at 1/10 of the multiply-adds for K = 10, and keeps the
// [K, 200+d_e] key rows out of the matrix pipe: they are read twice by the owning wave (scores,
// then the weighted sum) as plain coalesced rows.  The dense contractions that remain
// (q, Wk^T q, Wv x, out_proj, fc1, fc2) run on MFMA with M = 16 queries.
#include "common.h"

namespace {

constexpr int D = ORCA_D;          // 100
constexpr int QD = 2 * D;          // query / embed dim 200
constexpr int TQ = 16;             // queries per workgroup (one MFMA M tile)
constexpr int NW = 8;              // waves per workgroup
constexpr int ASTR = 212;          // LDS row stride of the 200-wide activation buffers (== 4 mod 16)
constexpr int MSTR = 308;          // LDS row stride of the 300-wide merger input
constexpr int MAXE = (ORCA_MAX_DE + 63) / 64;   // edge-feature registers per lane

struct Blob {   // float offsets inside the packed parameter blob
  int kdim, KP, NTk, KBk;
  int64_t pw_q, b_q, pw_kt, pw_v, b_v, pw_o, b_o, pw_1, b_1, pw_2, b_2, ln_w, ln_b, total;
};

__host__ __device__ inline Blob blob_layout(int d_e) {
  Blob L;
  L.kdim = QD + d_e;
  L.NTk = pw_tiles(L.kdim + 1);     // n-tiles of the folded key map (+1: the q.bk column)
  L.KP = L.NTk * 16;
  L.KBk = L.NTk;                    // k-blocks of the value map (reduction over the same KP columns)
  int64_t o = 0;
  L.pw_q = o;  o += pw_floats(QD, QD);
  L.b_q = o;   o += 208;
  L.pw_kt = o; o += 2 * pw_floats(L.kdim + 1, D);
  L.pw_v = o;  o += 2 * (int64_t)pw_tiles(D) * L.KBk * 256;
  L.b_v = o;   o += 208;
  L.pw_o = o;  o += pw_floats(QD, QD);
  L.b_o = o;   o += 208;
  L.pw_1 = o;  o += pw_floats(D, 3 * D);
  L.b_1 = o;   o += 112;
  L.pw_2 = o;  o += pw_floats(D, D);
  L.b_2 = o;   o += 112;
  L.ln_w = o;  o += 112;
  L.ln_b = o;  o += 112;
  L.total = o;
  return L;
}

__global__ void k_pack(float* __restrict__ dst, int KB, int n0, int n_cnt, int Kd, const float* __restrict__ src,
                       int64_t sn, int64_t sk) {
  // one thread per destination float of the n-tiles that intersect [n0, n0+n_cnt)
  int nt0 = n0 / 16;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int nt_cnt = (n0 + n_cnt + 15) / 16 - nt0;
  if (idx >= (int64_t)nt_cnt * KB * 256) return;
  int s = idx & 3;
  int lane = (idx >> 2) & 63;
  int64_t blk = idx >> 8;
  int kb = (int)(blk % KB);
  int nt = nt0 + (int)(blk / KB);
  int n = nt * 16 + (lane & 15);
  int k = kb * 16 + 4 * (lane >> 4) + s;
  if (n < n0 || n >= n0 + n_cnt) return;
  float v = (k < Kd) ? src[(int64_t)(n - n0) * sn + (int64_t)k * sk] : 0.0f;
  dst[((int64_t)nt * KB + kb) * 256 + lane * 4 + s] = v;
}

__global__ void k_copy_pad(float* dst, const float* src, int n, int n_pad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) dst[i] = (i < n) ? src[i] : 0.0f;
}

}  // namespace

int orca_pack_matrix(float* dst, int NT_total, int KB, int n0, int n_cnt, int Kd, const float* src, int64_t sn,
                     int64_t sk, hipStream_t stream) {
  (void)NT_total;
  int nt_cnt = (n0 + n_cnt + 15) / 16 - n0 / 16;
  int64_t total = (int64_t)nt_cnt * KB * 256;
  k_pack<<<(unsigned)ceil_div64(total, 256), 256, 0, stream>>>(dst, KB, n0, n_cnt, Kd, src, sn, sk);
  return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
}

static int copy_pad(float* dst, const float* src, int n, int n_pad, hipStream_t s) {
  k_copy_pad<<<(n_pad + 255) / 256, 256, 0, s>>>(dst, src, n, n_pad);
  return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
}

ORCA_API int64_t orca_tgat_blob_floats(int d_e) {
  if (d_e < 1 || d_e > ORCA_MAX_DE) return -1;
  return blob_layout(d_e).total;
}

ORCA_API int orca_tgat_pack(int d_e, const float* q_w, const float* k_w, const float* v_w, const float* in_b,
                            const float* o_w, const float* o_b, const float* fc1_w, const float* fc1_b,
                            const float* fc2_w, const float* fc2_b, const float* ln_w, const float* ln_b,
                            float* blob, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(d_e >= 1 && d_e <= ORCA_MAX_DE, "orca_tgat_pack: d_e=%d outside [1,%d]", d_e, ORCA_MAX_DE);
  ORCA_REQUIRE(q_w && k_w && v_w && in_b && o_w && o_b && fc1_w && fc1_b && fc2_w && fc2_b && ln_w && ln_b && blob,
               "orca_tgat_pack: null pointer");
  Blob L = blob_layout(d_e);
  const int kdim = L.kdim;
  ORCA_HIP_TRY(hipMemsetAsync(blob, 0, (size_t)L.total * sizeof(float), s));
  int rc = ORCA_OK;
  // q projection: out n = row of q_proj_weight [200,200]
  rc |= orca_pack_matrix(blob + L.pw_q, pw_tiles(QD), pw_tiles(QD), 0, QD, QD, q_w, QD, 1, s);
  rc |= copy_pad(blob + L.b_q, in_b, QD, 208, s);
  const int KB100 = pw_tiles(D);
  for (int h = 0; h < 2; ++h) {
    // folded key map of head h: out n = key column c (< kdim), reduction k = d (< 100):  Wk[h*100+d][c]
    float* kt = blob + L.pw_kt + (int64_t)h * pw_floats(kdim + 1, D);
    rc |= orca_pack_matrix(kt, L.NTk, KB100, 0, kdim, D, k_w + (int64_t)h * D * kdim, 1, kdim, s);
    // extra output column n = kdim: bk[h*100+d]  (gives the (0.1 q_h).bk_h score offset)
    rc |= orca_pack_matrix(kt, L.NTk, KB100, kdim, 1, D, in_b + QD + h * D, 0, 1, s);
    // value map of head h: out n = d (< 100), reduction k = c (< kdim):  Wv[h*100+d][c]
    float* pv = blob + L.pw_v + (int64_t)h * KB100 * L.KBk * 256;
    rc |= orca_pack_matrix(pv, KB100, L.KBk, 0, D, kdim, v_w + (int64_t)h * D * kdim, kdim, 1, s);
  }
  rc |= copy_pad(blob + L.b_v, in_b + 2 * QD, QD, 208, s);
  rc |= orca_pack_matrix(blob + L.pw_o, pw_tiles(QD), pw_tiles(QD), 0, QD, QD, o_w, QD, 1, s);
  rc |= copy_pad(blob + L.b_o, o_b, QD, 208, s);
  rc |= orca_pack_matrix(blob + L.pw_1, pw_tiles(D), pw_tiles(3 * D), 0, D, 3 * D, fc1_w, 3 * D, 1, s);
  rc |= copy_pad(blob + L.b_1, fc1_b, D, 112, s);
  rc |= orca_pack_matrix(blob + L.pw_2, pw_tiles(D), pw_tiles(D), 0, D, D, fc2_w, D, 1, s);
  rc |= copy_pad(blob + L.b_2, fc2_b, D, 112, s);
  rc |= copy_pad(blob + L.ln_w, ln_w, D, 112, s);
  rc |= copy_pad(blob + L.ln_b, ln_b, D, 112, s);
  if (rc != ORCA_OK) {
    orca_set_error("orca_tgat_pack: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}

namespace {

// Epilogue helper: lane holds C[m = 4*(lane>>4)+r][n = tile*16 + (lane&15)] in acc[r].
template <typename F>
__device__ __forceinline__ void for_acc(const f32x4& acc, int tile, int lane, F f) {
  int n = tile * 16 + (lane & 15);
  int m0 = 4 * (lane >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) f(m0 + r, n, acc[r]);
}

// C[16 x N] = A * PW (+bias) for the n-tiles owned by this wave, NT tiles at a time.
template <typename F>
__device__ __forceinline__ void gemm_phase(const float* A, int a_str, const f32x4* pw, int KB, int NTtot, int wave,
                                           int lane, F epilogue) {
  const float* a_row = A + (lane & 15) * a_str + 4 * (lane >> 4);
  for (int t0 = wave; t0 < NTtot; t0 += NW * 4) {
    int tile[4];
    int nt = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      tile[t] = t0 + t * NW;
      if (tile[t] < NTtot) nt = t + 1;
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    wave_gemm16<4>(a_row, pw, KB, tile, nt, acc, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < nt) for_acc(acc[t], tile[t], lane, epilogue);
  }
}

__global__ __launch_bounds__(NW * 64) void k_tgat_forward(const float* __restrict__ blob, int d_e,
                                                          orca_tgat_inputs in, int64_t Q, int K,
                                                          float* __restrict__ out, float* __restrict__ out_w) {
  extern __shared__ __align__(16) float lds[];
  const Blob L = blob_layout(d_e);
  const int kdim = L.kdim;
  const int KPS = L.KP + 4;                       // row stride of the q~ / x-bar buffer
  float* bufQ = lds;                              // [TQ][ASTR]  query vector, later o (concat heads)
  float* bufP = bufQ + TQ * ASTR;                 // [TQ][ASTR]  0.1*q, later relu(fc1)
  float* bufM = bufP + TQ * ASTR;                 // [TQ][MSTR]  [attn_out(200) | src(100)]
  float* bufT = bufM + TQ * MSTR;                 // [TQ][2][KPS] q~ then x-bar
  float* bufE = bufT + TQ * 2 * KPS;              // [NW][K][D]  time encodings of the wave's current query
  __shared__ int s_invalid[TQ];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t q0 = (int64_t)blockIdx.x * TQ;
  const f32x4* blob4 = reinterpret_cast<const f32x4*>(blob);

  // ---- P0: stage the query vectors [src | src_time] and the merger's copy of src --------------
  for (int idx = tid; idx < TQ * ASTR; idx += NW * 64) {
    int m = idx / ASTR, c = idx - m * ASTR;
    int64_t q = q0 + m;
    float v = 0.0f;
    if (q < Q && c < QD) {
      if (c < D) {
        int64_t row = in.src_idx ? (int64_t)in.src_idx[q] : q;
        const float* tab = (in.src_flags && in.src_flags[row]) ? in.src_overlay : in.src_table;
        v = tab[row * D + c];
        bufM[m * MSTR + QD + c] = v;
      } else {
        int cc = c - D;
        v = in.dt ? cosf(0.0f * in.time_w[cc] + in.time_b[cc]) : in.src_time[q * D + cc];
      }
    } else if (c < D) {
      bufM[m * MSTR + QD + c] = 0.0f;
    }
    bufQ[idx] = v;
    bufP[idx] = 0.0f;
  }
  for (int idx = tid; idx < TQ * (MSTR - 3 * D); idx += NW * 64) {   // zero the merger row padding
    int m = idx / (MSTR - 3 * D), c = idx - m * (MSTR - 3 * D);
    bufM[m * MSTR + 3 * D + c] = 0.0f;
  }
  __syncthreads();

  // ---- P1: q = Wq [src|time] + bq, scaled by sqrt(1/head_dim) ----------------------------------
  {
    const float* bq = blob + L.b_q;
    gemm_phase(bufQ, ASTR, blob4 + L.pw_q / 4, pw_tiles(QD), pw_tiles(QD), wave, lane,
               [&](int m, int n, float v) {
                 if (n < ASTR) bufP[m * ASTR + n] = (n < QD) ? (v + bq[n]) * 0.1f : 0.0f;
               });
  }
  __syncthreads();

  // ---- P2: q~_h = Wk_h^T (0.1 q_h)  (column kdim carries (0.1 q_h).bk_h) -----------------------
  for (int h = 0; h < 2; ++h) {
    const f32x4* pw = blob4 + (L.pw_kt + (int64_t)h * pw_floats(kdim + 1, D)) / 4;
    float* dstT = bufT + h * KPS;
    gemm_phase(bufP + h * D, ASTR, pw, pw_tiles(D), L.NTk, wave, lane,
               [&](int m, int n, float v) { dstT[m * 2 * KPS + n] = v; });
  }
  __syncthreads();

  // ---- P3: per query (one wave each): scores, masked softmax, weighted key sum ------------------
  for (int m = wave; m < TQ; m += NW) {
    const int64_t q = q0 + m;
    if (q >= Q) {
      if (lane == 0) s_invalid[m] = 1;
      for (int c = lane; c < 2 * KPS; c += 64) bufT[m * 2 * KPS + c] = 0.0f;
      continue;
    }
    float* te = bufE + wave * K * D;
    // lane j (< K) owns neighbour slot j
    int64_t nrow = 0, erow = 0;
    float dtj = 0.0f;
    bool masked = true;
    if (lane < K) {
      int64_t p = q * K + lane;
      nrow = in.nbr_idx ? (int64_t)in.nbr_idx[p] : p;
      erow = in.edge_idx ? (int64_t)in.edge_idx[p] : p;
      masked = in.nbr_ids ? (in.nbr_ids[p] == 0) : (in.mask[p] != 0);
      if (in.dt) dtj = in.dt[p];
    }
    const bool invalid = (__ballot(lane < K && !masked) == 0ull);   // temporal_attention.py:46
    if (invalid && lane == 0) {
      masked = false;                                               // :47
      if (!in.nbr_ids) in.mask[q * K] = 0;
    }
    if (lane == 0) s_invalid[m] = invalid ? 1 : 0;
    const float* ntab = in.nbr_table;
    if (lane < K && in.nbr_flags && in.nbr_flags[nrow]) ntab = in.nbr_overlay;
    const float* nptr = ntab + nrow * D;
    const float* eptr = in.edge_table + erow * d_e;
    // time encodings of the K slots -> LDS (computed once, used by both passes)
    for (int base = 0; base < K * D; base += 64) {   // uniform trip count: every lane takes part in the shuffle
      int idx = base + lane;
      int j = idx / D;
      if (j > K - 1) j = K - 1;
      int c = idx - j * D;
      float d = __shfl(dtj, j, 64);
      if (idx < K * D) te[idx] = in.dt ? cosf(d * in.time_w[c] + in.time_b[c]) : in.nbr_time[(q * K + j) * D + c];
    }
    __builtin_amdgcn_wave_barrier();
    const float* qt0 = bufT + m * 2 * KPS;
    const float* qt1 = qt0 + KPS;
    // this lane's columns of q~ (both heads): nbr part c, c+64; edge part; time part c, c+64
    const int c1 = lane + 64;
    float qn0a = qt0[lane], qn1a = qt1[lane];
    float qn0b = c1 < D ? qt0[c1] : 0.f, qn1b = c1 < D ? qt1[c1] : 0.f;
    float qe0[MAXE], qe1[MAXE];
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      int c = lane + 64 * i;
      qe0[i] = c < d_e ? qt0[D + c] : 0.f;
      qe1[i] = c < d_e ? qt1[D + c] : 0.f;
    }
    const int tb = D + d_e;
    float qt0a = qt0[tb + lane], qt1a = qt1[tb + lane];
    float qt0b = c1 < D ? qt0[tb + c1] : 0.f, qt1b = c1 < D ? qt1[tb + c1] : 0.f;
    const float off0 = qt0[kdim], off1 = qt1[kdim];
    // pass 1: scores
    float s0 = 0.f, s1 = 0.f;
    for (int j = 0; j < K; ++j) {
      const float* np = (const float*)__shfl((unsigned long long)nptr, j, 64);
      const float* ep = (const float*)__shfl((unsigned long long)eptr, j, 64);
      float xa = np[lane];
      float xb = c1 < D ? np[c1] : 0.f;
      float p0 = xa * qn0a + xb * qn0b;
      float p1 = xa * qn1a + xb * qn1b;
#pragma unroll
      for (int i = 0; i < MAXE; ++i) {
        int c = lane + 64 * i;
        float xe = c < d_e ? ep[c] : 0.f;
        p0 += xe * qe0[i];
        p1 += xe * qe1[i];
      }
      float ta = te[j * D + lane];
      float tbv = c1 < D ? te[j * D + c1] : 0.f;
      p0 += ta * qt0a + tbv * qt0b;
      p1 += ta * qt1a + tbv * qt1b;
      p0 = wave_sum(p0);
      p1 = wave_sum(p1);
      if (lane == j) { s0 = p0 + off0; s1 = p1 + off1; }
    }
    // masked softmax over the K slots (lanes 0..K-1)
    const bool live = lane < K && !masked;
    float m0 = wave_max(live ? s0 : -INFINITY), m1 = wave_max(live ? s1 : -INFINITY);
    float e0 = live ? expf(s0 - m0) : 0.f, e1 = live ? expf(s1 - m1) : 0.f;
    float a0 = e0 / wave_sum(e0), a1 = e1 / wave_sum(e1);
    if (out_w && lane < K) out_w[q * K + lane] = invalid ? 0.f : (a0 + a1) * 0.5f;   // :51-53
    // pass 2: x-bar_h = sum_j a_hj x_j   (rows re-read; they are L1/L2 resident)
    float an0 = 0.f, an1 = 0.f, bn0 = 0.f, bn1 = 0.f, at0 = 0.f, at1 = 0.f, bt0 = 0.f, bt1 = 0.f;
    float ae0[MAXE], ae1[MAXE];
#pragma unroll
    for (int i = 0; i < MAXE; ++i) { ae0[i] = 0.f; ae1[i] = 0.f; }
    for (int j = 0; j < K; ++j) {
      const float w0 = __shfl(a0, j, 64), w1 = __shfl(a1, j, 64);
      const float* np = (const float*)__shfl((unsigned long long)nptr, j, 64);
      const float* ep = (const float*)__shfl((unsigned long long)eptr, j, 64);
      float xa = np[lane];
      float xb = c1 < D ? np[c1] : 0.f;
      an0 += w0 * xa; an1 += w1 * xa; bn0 += w0 * xb; bn1 += w1 * xb;
#pragma unroll
      for (int i = 0; i < MAXE; ++i) {
        int c = lane + 64 * i;
        float xe = c < d_e ? ep[c] : 0.f;
        ae0[i] += w0 * xe;
        ae1[i] += w1 * xe;
      }
      float ta = te[j * D + lane];
      float tbv = c1 < D ? te[j * D + c1] : 0.f;
      at0 += w0 * ta; at1 += w1 * ta; bt0 += w0 * tbv; bt1 += w1 * tbv;
    }
    float* xb0 = bufT + m * 2 * KPS;
    float* xb1 = xb0 + KPS;
    xb0[lane] = an0; xb1[lane] = an1;
    if (c1 < D) { xb0[c1] = bn0; xb1[c1] = bn1; }
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      int c = lane + 64 * i;
      if (c < d_e) { xb0[D + c] = ae0[i]; xb1[D + c] = ae1[i]; }
    }
    xb0[tb + lane] = at0; xb1[tb + lane] = at1;
    if (c1 < D) { xb0[tb + c1] = bt0; xb1[tb + c1] = bt1; }
    if (lane == 0) { xb0[kdim] = 0.f; xb1[kdim] = 0.f; }
  }
  __syncthreads();

  // ---- P4: o_h = Wv_h x-bar_h + bv_h  ->  bufQ[:, h*100 + d] -------------------------------------
  {
    const float* bv = blob + L.b_v;
    const int NT100 = pw_tiles(D);
    const float* a_base = bufT + (lane & 15) * 2 * KPS + 4 * (lane >> 4);
    for (int item = wave; item < 2 * NT100; item += NW) {
      int h = item / NT100, t = item - h * NT100;
      const f32x4* pw = blob4 + (L.pw_v + (int64_t)h * NT100 * L.KBk * 256) / 4;
      int tile[1] = {t};
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      wave_gemm16<1>(a_base + h * KPS, pw, L.KBk, tile, 1, acc, lane);
      for_acc(acc[0], t, lane, [&](int m, int n, float v) {
        if (n < D) bufQ[m * ASTR + h * D + n] = v + bv[h * D + n];
      });
    }
  }
  __syncthreads();

  // ---- P5: attn = Wo o + bo, zeroed for all-padded neighbourhoods (:52) -> bufM[:, 0:200] -------
  {
    const float* bo = blob + L.b_o;
    gemm_phase(bufQ, ASTR, blob4 + L.pw_o / 4, pw_tiles(QD), pw_tiles(QD), wave, lane,
               [&](int m, int n, float v) {
                 if (n < QD) bufM[m * MSTR + n] = s_invalid[m] ? 0.0f : (v + bo[n]);
               });
  }
  __syncthreads();

  // ---- P6: h1 = relu(fc1 [attn | src] + b1) -> bufP[:, 0:112] -----------------------------------
  {
    const float* b1 = blob + L.b_1;
    gemm_phase(bufM, MSTR, blob4 + L.pw_1 / 4, pw_tiles(3 * D), pw_tiles(D), wave, lane,
               [&](int m, int n, float v) { bufP[m * ASTR + n] = (n < D) ? fmaxf(v + b1[n], 0.0f) : 0.0f; });
  }
  __syncthreads();

  // ---- P7: y = fc2 h1 + b2 -> bufQ[:, 0:100] -----------------------------------------------------
  {
    const float* b2 = blob + L.b_2;
    gemm_phase(bufP, ASTR, blob4 + L.pw_2 / 4, pw_tiles(D), pw_tiles(D), wave, lane,
               [&](int m, int n, float v) {
                 if (n < D) bufQ[m * ASTR + n] = v + b2[n];
               });
  }
  __syncthreads();

  // ---- P8: LayerNorm(100), eps 1e-5, biased variance (:55) ---------------------------------------
  {
    const float* g = blob + L.ln_w;
    const float* b = blob + L.ln_b;
    for (int m = wave; m < TQ; m += NW) {
      const int64_t q = q0 + m;
      if (q >= Q) continue;
      const int c1 = lane + 64;
      float ya = bufQ[m * ASTR + lane];
      float yb = c1 < D ? bufQ[m * ASTR + c1] : 0.f;
      float mean = wave_sum(ya + yb) * (1.0f / D);
      float da = ya - mean, db = c1 < D ? yb - mean : 0.f;
      float var = wave_sum(da * da + db * db) * (1.0f / D);
      float rstd = 1.0f / sqrtf(var + 1e-5f);
      out[q * D + lane] = da * rstd * g[lane] + b[lane];
      if (c1 < D) out[q * D + c1] = db * rstd * g[c1] + b[c1];
    }
  }
}

}  // namespace

ORCA_API int orca_tgat_forward(const float* blob, int d_e, const orca_tgat_inputs* in, int64_t Q, int K, float* out,
                               float* out_weights, void* stream) {
  ORCA_REQUIRE(Q >= 0, "orca_tgat_forward: Q=%lld < 0", (long long)Q);
  if (Q == 0) return ORCA_OK;   // empty batch (an empty tensor has no address)
  ORCA_REQUIRE(blob && in && out, "orca_tgat_forward: null pointer");
  ORCA_REQUIRE(d_e >= 1 && d_e <= ORCA_MAX_DE, "orca_tgat_forward: d_e=%d outside [1,%d]", d_e, ORCA_MAX_DE);
  ORCA_REQUIRE(K >= 1 && K <= ORCA_MAX_K, "orca_tgat_forward: K=%d outside [1,%d]", K, ORCA_MAX_K);
  ORCA_REQUIRE(in->src_table && in->nbr_table && in->edge_table, "orca_tgat_forward: null feature table");
  ORCA_REQUIRE(in->nbr_ids || in->mask, "orca_tgat_forward: need nbr_ids or mask");
  ORCA_REQUIRE(in->dt ? (in->time_w && in->time_b) : (in->src_time && in->nbr_time),
               "orca_tgat_forward: need (dt, time_w, time_b) or (src_time, nbr_time)");
  ORCA_REQUIRE(!in->src_flags || in->src_overlay, "orca_tgat_forward: src_flags without src_overlay");
  ORCA_REQUIRE(!in->nbr_flags || in->nbr_overlay, "orca_tgat_forward: nbr_flags without nbr_overlay");
  Blob L = blob_layout(d_e);
  size_t lds_bytes = (size_t)(2 * TQ * ASTR + TQ * MSTR + TQ * 2 * (L.KP + 4) + NW * K * D) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_tgat_forward, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    attr_set = true;
  }
  ORCA_REQUIRE(lds_bytes <= 160 * 1024 - 256, "orca_tgat_forward: LDS budget exceeded (%zu B)", lds_bytes);
  k_tgat_forward<<<(unsigned)ceil_div64(Q, TQ), NW * 64, lds_bytes, (hipStream_t)stream>>>(blob, d_e, *in, Q, K, out,
                                                                                           out_weights);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}
