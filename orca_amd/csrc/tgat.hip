// Fused TGAT attention layer (model/temporal_attention.py:24-56 of the reference) for gfx950.
//
// One workgroup (16 waves) owns TQ = 16 queries -- one MFMA M tile -- and runs the whole layer on them
// without touching HBM in between: gather -> time encoding -> projections on v_mfma_f32_16x16x4_f32 ->
// masked softmax over the K neighbours with wave shuffles -> output projection -> MergeLayer -> LayerNorm.
// In the per-query stages wave m owns query m; in the GEMM stages the 16 waves split the N tiles
// (WaveGemm).  At the batch shape of the benchmark (4095 queries) that is one workgroup per CU: there is no
// second tile to overlap stages with, so every stage requests the next stage's weights before its barrier
// and the (K = 10, d_e = 172) specialisation unrolls every chain branch-free.
//
// Algebra.  nn.MultiheadAttention with one query per K keys computes, per head h,
//   s_j = (0.1*q_h) . (Wk_h x_j + bk_h),  a = softmax_j(s),  o_h = sum_j a_j (Wv_h x_j + bv_h).
// Both K-sized projections are folded through the linear maps:
//   s_j = (Wk_h^T (0.1*q_h)) . x_j + (0.1*q_h).bk_h          o_h = Wv_h (sum_j a_j x_j) + bv_h
// which is the same function (sum_j a_j = 1) at 1/10 of the multiply-adds for K = 10, and keeps the
// [K, 200+d_e] key rows out of the matrix pipe: they are read twice by the owning wave (scores,
// then the weighted sum) as plain coalesced rows.
// The linear tail is folded too.  Between the weighted key sum x-bar_h and the merger's ReLU everything is
// linear:  fc1([Wo (Wv x-bar + bv) + bo | src]) = sum_h (W1a Wo_h Wv_h) x-bar_h + W1b src + b1 + W1a (Wo bv + bo),
// so the value map, the output projection and fc1 are ONE matrix, multiplied out once in fp64 when the weights
// are packed (orca_tgat_pack): a [100 x (2*k_dim + 1 + 100)] map instead of three stages with two barriers
// and 1.7x its multiply-adds.  A query whose neighbours are all padding has attn = 0
// (temporal_attention.py:52): its x-bar and the indicator column that carries the constant are zeroed.
// The dense contractions that remain (q, Wk^T q, the folded tail, fc2) run on MFMA with M = 16 queries.
#include "common.h"

namespace {

constexpr int D = ORCA_D;          // 100
constexpr int QD = 2 * D;          // query / embed dim 200
constexpr int TQ = 16;             // queries per workgroup (one MFMA M tile)
constexpr int NW = 16;             // waves per workgroup: one per query in the per-query stage
constexpr int ASTR = 212;          // LDS row stride of the 200-wide activation buffers (== 4 mod 16)
constexpr int HSTR = 116;          // LDS row stride of the 112-wide hidden activation of the merger
constexpr int MAXE = (ORCA_MAX_DE + 63) / 64;   // edge-feature registers per lane

struct Blob {   // float offsets inside the packed parameter blob
  int kdim, KP, NTk, KBk, KBT;
  int64_t pw_q, b_q, pw_kt, pw_t, b_1, pw_2, b_2, ln_w, ln_b, pw_qk, b_qk, qk_flag, scr_u, scr_f, scr_t, scr_a, total;
};

__host__ __device__ constexpr Blob blob_layout(int d_e) {
  Blob L{};
  L.kdim = QD + d_e;
  L.NTk = pw_tiles(L.kdim + 1);     // n-tiles of the folded key map (+1: the q.bk column)
  L.KP = L.NTk * 16;
  L.KBk = L.NTk;                    // k-blocks of one head's x-bar (the same KP columns)
  // folded tail: reduction over [x-bar_0 (KP) | x-bar_1 (KP) | src (112)], padded to a multiple of 4 k-blocks so
  // that it splits evenly over 2 or 4 waves
  L.KBT = (2 * L.KBk + pw_tiles(D) + 3) / 4 * 4;
  int64_t o = 0;
  L.pw_q = o;  o += pw_floats(QD, QD);
  L.b_q = o;   o += 208;
  L.pw_kt = o; o += 2 * pw_floats(L.kdim + 1, D);
  L.pw_t = o;  o += (int64_t)pw_tiles(D) * L.KBT * 256;
  L.b_1 = o;   o += 112;
  L.pw_2 = o;  o += pw_floats(D, D);
  L.b_2 = o;   o += 112;
  L.ln_w = o;  o += 112;
  L.ln_b = o;  o += 112;
  // gather mode only (orca_tgat_pack_gather): the q projection folded into the key map, per head
  // [kdim+1 x 100] over src plus a constant column vector; qk_flag != 0 once they are filled in
  L.pw_qk = o; o += 2 * pw_floats(L.kdim + 1, D);
  L.b_qk = o;  o += 2 * L.KP;
  L.qk_flag = o; o += 16;
  // scratch of the pack routines (the fp64 products of the folds), kept inside the blob so that packing neither
  // allocates nor synchronises; the kernel never reads it.  Offsets are even: the fp64 parts stay 8-byte aligned.
  L.scr_u = o; o += 2 * (2 * D * D + 3 * QD);                      // double [2*D*D + 3*QD]
  L.scr_f = o; o += (int64_t)D * L.KBT * 16;                        // float  [D, KBT*16]
  L.scr_t = o; o += 2 * QD;                                         // double [QD]
  L.scr_a = o; o += 2 * (int64_t)(L.kdim + 1) * D + ((L.kdim + 1) * D & 1);   // float [2, kdim+1, D]
  L.total = o;
  return L;
}

// ---- folding the linear tail (setup time, fp64 accumulation) ---------------------------------------------
// U[h][n][d] = sum_j W1[n][j] Wo[j][h*100 + d]            (fc1_w [100, 300], o_w [200, 200], row-major)
// U[2*D*D + h*QD + j] = sum_{k in head h} Wo[j][k] bv[k]  (the out-projection of head h's value bias: it reaches the
//                        output scaled by the sum of that head's attention weights -- 1 unless dropout is on)
// U[2*D*D + 2*QD + j] = bo[j]
// The products themselves run on the fp64 matrix pipe, one wave per 16 x 16 tile: as one thread per element with a
// 100- or 200-step loop these folds were latency chains of 12-23 us per kernel (a re-pack follows every optimizer
// step).  C tile (rows m0.., columns n0..) of A [M x K] B [K x N] on v_mfma_f64_16x16x4f64: lane (r = lane & 15,
// kq = lane >> 4) supplies A(m0 + r, k0 + kq) and B(k0 + kq, n0 + r) and receives C(m0 + 4 i + kq, n0 + r), i < 4.
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int K, class GA, class GB, class ST>
__device__ __forceinline__ void fold_tile(int m0, int n0, int M, int N, GA A, GB B, ST store) {
  const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
  const int m = m0 + r < M ? m0 + r : M - 1, n = n0 + r < N ? n0 + r : N - 1;     // clamped, never skipped
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  // (every operand of up to 25 steps is requested before the first product, and the scheduler is told to leave it so:
  // with a loop left over, or with the loads sunk to their products, each load is waited for where it is issued -- 29 us
  // for the K = 200 product instead of 4)
  constexpr int STEPS = K / 4, CH = 25;
  static_assert(K % 4 == 0, "whole MFMA steps");
#pragma unroll
  for (int c0 = 0; c0 < STEPS; c0 += CH) {
    double a[CH], b[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (c0 + i < STEPS) { a[i] = A(m, 4 * (c0 + i) + kq); b[i] = B(4 * (c0 + i) + kq, n); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < CH; ++i)
      if (c0 + i < STEPS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (m0 + 4 * i + kq < M && n0 + r < N) store(m0 + 4 * i + kq, n0 + r, acc[i]);
}
constexpr int FOLD_T100 = (D + 15) / 16;     // 16-wide tiles of a 100-long index
// tile w of U[h][n][d] = sum_j W1[n][j] Wo[j][h*100 + d]
__device__ __forceinline__ void fold_u_tile(int w, const float* __restrict__ fc1_w, const float* __restrict__ o_w,
                                            double* __restrict__ U) {
  if (w >= 2 * FOLD_T100 * FOLD_T100) return;
  const int h = w / (FOLD_T100 * FOLD_T100), mt = (w / FOLD_T100) % FOLD_T100, nt = w % FOLD_T100;
  fold_tile<QD>(mt * 16, nt * 16, D, D,
                [&](int n, int j) { return (double)fc1_w[n * 3 * D + j]; },
                [&](int j, int d) { return (double)o_w[j * QD + h * D + d]; },
                [&](int n, int d, double v) { U[(h * D + n) * D + d] = v; });
}
// the rest of U: i >= 2*D*D
__device__ __forceinline__ void fold_u_rest(int r, const float* __restrict__ o_w, const float* __restrict__ o_b,
                                            const float* __restrict__ bv, double* __restrict__ U) {
  if (r >= 3 * QD) return;
  const int part = r / QD, j = r % QD;
  if (part < 2) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) t += (double)o_w[j * QD + part * D + k] * (double)bv[part * D + k];
    U[2 * D * D + r] = t;
  } else {
    U[2 * D * D + r] = (double)o_b[j];
  }
}
// F [100, KBT*16]: columns h*KP + c (c < kdim) = sum_d U[h][n][d] Wv[h*100 + d][c];  column h*KP + kdim = the
// constant W1a Wo_h bv_h of head h (multiplied by that head's weight sum);  columns 2*KP + k (k < 100) =
// W1[n][200 + k];  column 2*KP + 100 = W1a bo (multiplied by 1 for a valid query);  zero elsewhere.
__device__ __forceinline__ void fold_f_tile(int w, const double* __restrict__ U, const float* __restrict__ v_w, int kdim,
                                            int KP, int width, float* __restrict__ F) {
  const int ntc = (kdim + 15) / 16;
  if (w >= 2 * FOLD_T100 * ntc) return;
  const int h = w / (FOLD_T100 * ntc), mt = (w / ntc) % FOLD_T100, nt = w % ntc;
  fold_tile<D>(mt * 16, nt * 16, D, kdim,
               [&](int n, int d) { return U[(h * D + n) * D + d]; },
               [&](int d, int c) { return (double)v_w[(int64_t)(h * D + d) * kdim + c]; },
               [&](int n, int c, double v) { F[n * width + h * KP + c] = (float)v; });
}
// every element of F outside the two [100 x kdim] products and the three dot-product columns (fold_f_dot)
__device__ __forceinline__ void fold_f_rest(int i, const double* __restrict__ U, const float* __restrict__ fc1_w, int kdim,
                                            int KP, int width, float* __restrict__ F) {
  if (i >= D * width) return;
  const int n = i / width, col = i % width;
  float v = 0.0f;
  if (col < 2 * KP) {
    const int h = col / KP, c = col - h * KP;
    if (c <= kdim) return;                     // (fold_f_tile, fold_f_dot)
  } else if (col < 2 * KP + D) {
    v = fc1_w[n * 3 * D + QD + (col - 2 * KP)];
  } else if (col == 2 * KP + D) {
    return;                                    // (fold_f_dot)
  }
  F[i] = v;
}
// The three columns of F that are dot products over j < 200 of row n of W1 with a vector of U: head h's value-bias
// column (which = h) and the out-projection-bias column (which = 2).  One wave per (n, which), the lanes stride j:
// as one thread per element these were 300 threads walking 200 dependent loads each -- 12 of stage 2's 18.6 us.
__device__ __forceinline__ void fold_f_dot(int w, const double* __restrict__ U, const float* __restrict__ fc1_w, int kdim,
                                           int KP, int width, float* __restrict__ F) {
  if (w >= 3 * D) return;
  const int n = w / 3, which = w % 3, lane = threadIdx.x & 63;
  const double* u = U + 2 * D * D + which * QD;
  double acc = 0.0;
#pragma unroll
  for (int j0 = 0; j0 < QD; j0 += 64) {
    const int j = j0 + lane < QD ? j0 + lane : QD - 1;
    const double t = (double)fc1_w[n * 3 * D + j] * u[j];
    acc += j0 + lane < QD ? t : 0.0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) F[n * width + (which < 2 ? which * KP + kdim : 2 * KP + D)] = (float)acc;
}

// ---- folding the q projection into the key map (gather mode: the source's time encoding is the constant
// cos(time_b), embedding_module.py:126) ---------------------------------------------------------------------
// t[j] = bq[j] + sum_i Wq[j][100 + i] cos(time_b[i])          (q_w [200, 200] row-major, j < 200)
__device__ __forceinline__ void fold_qt(int j, const float* __restrict__ q_w, const float* __restrict__ in_b,
                                        const float* __restrict__ time_b, double* __restrict__ t) {
  if (j >= QD) return;
  double acc = (double)in_b[j];
#pragma unroll
  for (int i = 0; i < D; ++i)
    acc += (double)q_w[j * QD + D + i] * (double)orca_cosf(orca_time_arg(0.0f, 0.0f, time_b[i]));
  t[j] = acc;
}
// A [2][kdim+1][100]: A[h][c][s] = 0.1 sum_d Wk'[h][d][c] Wq[h*100 + d][s], with Wk'[h][d][kdim] = bk[h*100 + d];
// a [2][KP]: a[h][c] = 0.1 sum_d Wk'[h][d][c] t[h*100 + d]    (zero beyond kdim) -- column s = 100 of the same product
__device__ __forceinline__ void fold_qk_tile(int w, const float* __restrict__ q_w, const float* __restrict__ k_w,
                                             const float* __restrict__ in_b, const double* __restrict__ t, int kdim, int KP,
                                             float* __restrict__ A, float* __restrict__ a) {
  const int mtc = (kdim + 1 + 15) / 16;
  if (w >= 2 * mtc * FOLD_T100) return;
  const int h = w / (mtc * FOLD_T100), mt = (w / FOLD_T100) % mtc, nt = w % FOLD_T100;
  const double scale = (double)0.1f;
  // (one unconditional load per operand: the choice is made on the address, or after both loads -- a load under a
  // lane-dependent branch is waited for where it stands, and 25 such steps in a row were 12 of this launch's 18 us)
  fold_tile<D>(mt * 16, nt * 16, kdim + 1, D + 1,
               [&](int c, int d) {
                 const float* p = c < kdim ? k_w + (int64_t)(h * D + d) * kdim + c : in_b + QD + h * D + d;
                 return (double)*p;
               },
               [&](int d, int s_) {
                 const float qv = q_w[(int64_t)(h * D + d) * QD + (s_ < D ? s_ : D - 1)];
                 const double tv = t[h * D + d];
                 return s_ < D ? (double)qv : tv;
               },
               [&](int c, int s_, double v) {
                 if (s_ < D) A[((int64_t)h * (kdim + 1) + c) * D + s_] = (float)(scale * v);
                 else a[h * KP + c] = (float)(scale * v);
               });
}
// The folds of one or several layers, both kinds at once (a re-pack after every optimizer step was eight fold launches
// per layer pair): stage 1 is U and t of every layer in one launch, stage 2 F and A.  Block roles by index.
constexpr int FOLD_LAYERS = 4;
struct FoldLayers {
  int n, kdim, KP, width, tail, gather;      // tail: the folded tail (orca_tgat_pack); gather: the q fold (orca_tgat_pack_gather)
  const float *q_w[FOLD_LAYERS], *k_w[FOLD_LAYERS], *v_w[FOLD_LAYERS], *in_b[FOLD_LAYERS], *o_w[FOLD_LAYERS],
      *o_b[FOLD_LAYERS], *fc1_w[FOLD_LAYERS];
  const float* time_b;
  double* U[FOLD_LAYERS];
  float* F[FOLD_LAYERS];
  double* t[FOLD_LAYERS];
  float* A[FOLD_LAYERS];
  float* a[FOLD_LAYERS];
};
constexpr int FOLD_BU = (2 * FOLD_T100 * FOLD_T100 + 3) / 4, FOLD_BR = (3 * QD + 255) / 256;
__global__ __launch_bounds__(256) void k_fold_stage1(FoldLayers L) {
  const int per = FOLD_BU + FOLD_BR + 1, l = blockIdx.x / per, local = blockIdx.x % per;
  if (local < FOLD_BU) {
    if (L.tail) fold_u_tile(local * 4 + (threadIdx.x >> 6), L.fc1_w[l], L.o_w[l], L.U[l]);
  } else if (local < FOLD_BU + FOLD_BR) {
    if (L.tail) fold_u_rest((local - FOLD_BU) * 256 + threadIdx.x, L.o_w[l], L.o_b[l], L.in_b[l] + 2 * QD, L.U[l]);
  } else if (L.gather) {
    fold_qt(threadIdx.x, L.q_w[l], L.in_b[l], L.time_b, L.t[l]);
  }
}
constexpr int FOLD_BD = (3 * D + 3) / 4;      // fold_f_dot: a wave per dot product
__global__ __launch_bounds__(256) void k_fold_stage2(FoldLayers L, int bft, int bfr, int bk) {
  const int per = bft + bfr + FOLD_BD + bk, l = blockIdx.x / per, local = blockIdx.x % per;
  if (local < bft) {
    if (L.tail) fold_f_tile(local * 4 + (threadIdx.x >> 6), L.U[l], L.v_w[l], L.kdim, L.KP, L.width, L.F[l]);
  } else if (local < bft + bfr) {
    if (L.tail) fold_f_rest((local - bft) * 256 + threadIdx.x, L.U[l], L.fc1_w[l], L.kdim, L.KP, L.width, L.F[l]);
  } else if (local < bft + bfr + FOLD_BD) {
    if (L.tail) fold_f_dot((local - bft - bfr) * 4 + (threadIdx.x >> 6), L.U[l], L.fc1_w[l], L.kdim, L.KP, L.width, L.F[l]);
  } else if (L.gather) {
    fold_qk_tile((local - bft - bfr - FOLD_BD) * 4 + (threadIdx.x >> 6), L.q_w[l], L.k_w[l], L.in_b[l], L.t[l], L.kdim, L.KP, L.A[l], L.a[l]);
  }
}
// launches both stages for the layers described by F (host side)
int fold_launch(const FoldLayers& F, hipStream_t s) {
  const int bft = (2 * FOLD_T100 * ((F.kdim + 15) / 16) + 3) / 4, bfr = (D * F.width + 255) / 256,
            bk = (2 * ((F.kdim + 1 + 15) / 16) * FOLD_T100 + 3) / 4;
  k_fold_stage1<<<(unsigned)(F.n * (FOLD_BU + FOLD_BR + 1)), 256, 0, s>>>(F);
  k_fold_stage2<<<(unsigned)(F.n * (bft + bfr + FOLD_BD + bk)), 256, 0, s>>>(F, bft, bfr, bk);
  return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
}
__global__ void k_pack(float* __restrict__ dst, int KB, int n0, int n_cnt, int Kd, const float* __restrict__ src,
                       int64_t sn, int64_t sk, int tail1) {
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
  if (tail1 && kb == KB - 1) k = s == 0 ? kb * 16 + (lane >> 4) : Kd;   // WaveGemm TAIL1: four columns, all in step 0
  if (n < n0 || n >= n0 + n_cnt) return;
  float v = (k < Kd) ? src[(int64_t)(n - n0) * sn + (int64_t)k * sk] : 0.0f;
  dst[((int64_t)nt * KB + kb) * 256 + lane * 4 + s] = v;
}

struct PackGroup {
  int n;
  int first[ORCA_PACK_GROUP_MAX + 1];     // first block of each job
  OrcaPackJob j[ORCA_PACK_GROUP_MAX];
};
__global__ __launch_bounds__(256) void k_pack_group(PackGroup grp) {
  int p = 0;
#pragma unroll
  for (int i = 1; i < ORCA_PACK_GROUP_MAX; ++i) p += (i < grp.n && (int)blockIdx.x >= grp.first[i]) ? 1 : 0;
  p = __builtin_amdgcn_readfirstlane(p);
  const OrcaPackJob j = grp.j[p];
  const int64_t idx = (int64_t)((int)blockIdx.x - grp.first[p]) * 256 + threadIdx.x;
  if (j.kind == 1) {
    if (idx < j.n_pad) j.dst[idx] = idx < j.n ? j.src[idx] : 0.0f;
    return;
  }
  if (j.kind == 2) {
    if (idx == 0) *j.dst = 1.0f;
    return;
  }
  // one thread per destination float of the n-tiles that intersect [n0, n0+n_cnt)   (k_pack)
  const int nt0 = j.n0 / 16;
  const int nt_cnt = (j.n0 + j.n_cnt + 15) / 16 - nt0;
  if (idx >= (int64_t)nt_cnt * j.KB * 256) return;
  const int s = idx & 3;
  const int lane = (idx >> 2) & 63;
  const int64_t blk = idx >> 8;
  const int kb = (int)(blk % j.KB);
  const int nt = nt0 + (int)(blk / j.KB);
  const int n = nt * 16 + (lane & 15);
  int k = kb * 16 + 4 * (lane >> 4) + s;
  if (j.tail1 && kb == j.KB - 1) k = s == 0 ? kb * 16 + (lane >> 4) : j.Kd;
  if (n < j.n0 || n >= j.n0 + j.n_cnt) return;
  const float v = (k < j.Kd) ? j.src[(int64_t)(n - j.n0) * j.sn + (int64_t)k * j.sk] : 0.0f;
  j.dst[((int64_t)nt * j.KB + kb) * 256 + lane * 4 + s] = v;
}

}  // namespace

int orca_pack_group(const OrcaPackJob* jobs, int n, hipStream_t stream) {
  if (n <= 0) return ORCA_OK;
  if (n > ORCA_PACK_GROUP_MAX) {
    orca_set_error("orca_pack_group: %d jobs in one group (at most %d)", n, ORCA_PACK_GROUP_MAX);
    return ORCA_EINVAL;
  }
  PackGroup grp{};
  grp.n = n;
  int first = 0;
  for (int i = 0; i < n; ++i) {
    grp.first[i] = first;
    grp.j[i] = jobs[i];
    int64_t total = 1;
    if (jobs[i].kind == 0) total = (int64_t)((jobs[i].n0 + jobs[i].n_cnt + 15) / 16 - jobs[i].n0 / 16) * jobs[i].KB * 256;
    else if (jobs[i].kind == 1) total = jobs[i].n_pad;
    first += (int)ceil_div64(total > 0 ? total : 1, 256);
  }
  grp.first[n] = first;
  k_pack_group<<<(unsigned)first, 256, 0, stream>>>(grp);
  return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
}

int orca_pack_matrix(float* dst, int NT_total, int KB, int n0, int n_cnt, int Kd, const float* src, int64_t sn,
                     int64_t sk, hipStream_t stream, bool tail1) {
  (void)NT_total;
  int nt_cnt = (n0 + n_cnt + 15) / 16 - n0 / 16;
  int64_t total = (int64_t)nt_cnt * KB * 256;
  k_pack<<<(unsigned)ceil_div64(total, 256), 256, 0, stream>>>(dst, KB, n0, n_cnt, Kd, src, sn, sk, tail1 ? 1 : 0);
  return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
}

namespace {
// the pack jobs of one layer's blob (after its folds): 11 for the dense boundary, 3 more for gather mode
int tgat_pack_jobs(OrcaPackJob* jobs, const Blob& L, const float* q_w, const float* k_w, const float* in_b,
                   const float* fc1_b, const float* fc2_w, const float* fc2_b, const float* ln_w, const float* ln_b,
                   float* blob) {
  const int kdim = L.kdim, width = L.KBT * 16, KB100 = pw_tiles(D);
  const float* F = blob + L.scr_f;
  int nj = 0;
  // q projection: out n = row of q_proj_weight [200,200]
  jobs[nj++] = orca_pack_job(blob + L.pw_q, pw_tiles(QD), 0, QD, QD, q_w, QD, 1);
  jobs[nj++] = orca_copy_pad_job(blob + L.b_q, in_b, QD, 208);
  for (int h = 0; h < 2; ++h) {
    // folded key map of head h: out n = key column c (< kdim), reduction k = d (< 100):  Wk[h*100+d][c]
    float* kt = blob + L.pw_kt + (int64_t)h * pw_floats(kdim + 1, D);
    jobs[nj++] = orca_pack_job(kt, KB100, 0, kdim, D, k_w + (int64_t)h * D * kdim, 1, kdim, true);
    // extra output column n = kdim: bk[h*100+d]  (gives the (0.1 q_h).bk_h score offset)
    jobs[nj++] = orca_pack_job(kt, KB100, kdim, 1, D, in_b + QD + h * D, 0, 1, true);
  }
  jobs[nj++] = orca_pack_job(blob + L.pw_t, L.KBT, 0, D, width, F, width, 1);
  jobs[nj++] = orca_copy_pad_job(blob + L.b_1, fc1_b, D, 112);
  jobs[nj++] = orca_pack_job(blob + L.pw_2, pw_tiles(D), 0, D, D, fc2_w, D, 1);
  jobs[nj++] = orca_copy_pad_job(blob + L.b_2, fc2_b, D, 112);
  jobs[nj++] = orca_copy_pad_job(blob + L.ln_w, ln_w, D, 112);
  jobs[nj++] = orca_copy_pad_job(blob + L.ln_b, ln_b, D, 112);
  return nj;
}
int tgat_pack_gather_jobs(OrcaPackJob* jobs, const Blob& L, float* blob) {
  const float* A = blob + L.scr_a;
  for (int h = 0; h < 2; ++h)
    jobs[h] = orca_pack_job(blob + L.pw_qk + (int64_t)h * pw_floats(L.kdim + 1, D), pw_tiles(D), 0, L.kdim + 1, D,
                            A + (int64_t)h * (L.kdim + 1) * D, D, 1, true);
  jobs[2] = orca_flag_job(blob + L.qk_flag);
  return 3;
}
}  // namespace

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
  // the folded tail (see the head of this file): multiplied out in fp64 in scratch memory first
  const int width = L.KBT * 16;
  {
    FoldLayers F{};
    F.n = 1; F.kdim = kdim; F.KP = L.KP; F.width = width; F.tail = 1; F.gather = 0;
    F.q_w[0] = q_w; F.k_w[0] = k_w; F.v_w[0] = v_w; F.in_b[0] = in_b; F.o_w[0] = o_w; F.o_b[0] = o_b; F.fc1_w[0] = fc1_w;
    F.U[0] = reinterpret_cast<double*>(blob + L.scr_u);
    F.F[0] = blob + L.scr_f;
    rc |= fold_launch(F, s);
  }
  OrcaPackJob jobs[ORCA_PACK_GROUP_MAX];
  int nj = tgat_pack_jobs(jobs, L, q_w, k_w, in_b, fc1_b, fc2_w, fc2_b, ln_w, ln_b, blob);
  rc |= orca_pack_group(jobs, nj, s);
  if (rc != ORCA_OK) {
    orca_set_error("orca_tgat_pack: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}

ORCA_API int orca_tgat_pack_gather(int d_e, const float* q_w, const float* k_w, const float* in_b,
                                   const float* time_b, float* blob, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(d_e >= 1 && d_e <= ORCA_MAX_DE, "orca_tgat_pack_gather: d_e=%d outside [1,%d]", d_e, ORCA_MAX_DE);
  ORCA_REQUIRE(q_w && k_w && in_b && time_b && blob, "orca_tgat_pack_gather: null pointer");
  Blob L = blob_layout(d_e);
  const int kdim = L.kdim;
  int rc = ORCA_OK;
  hipError_t e0 = hipMemsetAsync(blob + L.b_qk, 0, sizeof(float) * 2 * L.KP, s);
  {
    FoldLayers F{};
    F.n = 1; F.kdim = kdim; F.KP = L.KP; F.width = L.KBT * 16; F.tail = 0; F.gather = 1; F.time_b = time_b;
    F.q_w[0] = q_w; F.k_w[0] = k_w; F.in_b[0] = in_b;
    F.t[0] = reinterpret_cast<double*>(blob + L.scr_t);
    F.A[0] = blob + L.scr_a;
    F.a[0] = blob + L.b_qk;
    rc |= fold_launch(F, s);
  }
  OrcaPackJob jobs[3];
  tgat_pack_gather_jobs(jobs, L, blob);
  rc |= orca_pack_group(jobs, 3, s);
  ORCA_HIP_TRY(e0);
  if (rc != ORCA_OK) {
    orca_set_error("orca_tgat_pack_gather: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}

ORCA_API int orca_tgat_pack_layers(int n_layers, int d_e, const orca_tgat_params* P, const float* time_b,
                                   float* const* blobs, int blobs_packed_before, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(d_e >= 1 && d_e <= ORCA_MAX_DE, "orca_tgat_pack_layers: d_e=%d outside [1,%d]", d_e, ORCA_MAX_DE);
  ORCA_REQUIRE(n_layers >= 0 && (n_layers == 0 || (P && blobs)), "orca_tgat_pack_layers: null pointer");
  const Blob L = blob_layout(d_e);
  const int width = L.KBT * 16;
  for (int l0 = 0; l0 < n_layers; l0 += 2) {        // two layers per group of launches (the pack launch holds 32 jobs)
    const int n = n_layers - l0 < 2 ? n_layers - l0 : 2;
    FoldLayers F{};
    F.n = n; F.kdim = L.kdim; F.KP = L.KP; F.width = width; F.tail = 1; F.gather = time_b ? 1 : 0; F.time_b = time_b;
    OrcaPackJob jobs[ORCA_PACK_GROUP_MAX];
    static_assert(ORCA_PACK_GROUP_MAX >= 2 * 14, "two layers' jobs in one launch");
    int nj = 0;
    for (int i = 0; i < n; ++i) {
      const orca_tgat_params& p = P[l0 + i];
      float* blob = blobs[l0 + i];
      ORCA_REQUIRE(p.q_w && p.k_w && p.v_w && p.in_b && p.o_w && p.o_b && p.fc1_w && p.fc1_b && p.fc2_w && p.fc2_b && p.ln_w &&
                   p.ln_b && blob, "orca_tgat_pack_layers: null pointer (layer %d)", l0 + i);
      if (!blobs_packed_before) {     // the padding of the blob is written once; a re-pack leaves it alone
        ORCA_HIP_TRY(hipMemsetAsync(blob, 0, (size_t)L.total * sizeof(float), s));
      }
      F.q_w[i] = p.q_w; F.k_w[i] = p.k_w; F.v_w[i] = p.v_w; F.in_b[i] = p.in_b; F.o_w[i] = p.o_w; F.o_b[i] = p.o_b;
      F.fc1_w[i] = p.fc1_w;
      F.U[i] = reinterpret_cast<double*>(blob + L.scr_u);
      F.F[i] = blob + L.scr_f;
      F.t[i] = reinterpret_cast<double*>(blob + L.scr_t);
      F.A[i] = blob + L.scr_a;
      F.a[i] = blob + L.b_qk;
      nj += tgat_pack_jobs(jobs + nj, L, p.q_w, p.k_w, p.in_b, p.fc1_b, p.fc2_w, p.fc2_b, p.ln_w, p.ln_b, blob);
      if (time_b) nj += tgat_pack_gather_jobs(jobs + nj, L, blob);
    }
    if (fold_launch(F, s) != ORCA_OK || orca_pack_group(jobs, nj, s) != ORCA_OK) {
      orca_set_error("orca_tgat_pack_layers: kernel launch failed");
      return ORCA_EHIP;
    }
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

// One wave's share of a projection: NT 16x16 output tiles of C = A(16 x 16*KB) * PW, the B operands streamed
// from L2 G k-blocks ahead in a register ring.  prefetch() only needs the weights, so it is issued BEFORE the
// barrier that publishes the A operand: the first group's L2 round trip hides behind the barrier wait.
// KBC > 0 = compile-time k-block count: the whole chain is unrolled and BRANCH-FREE (the compiler's
// s_waitcnt placement gives up at control flow -- with run-time trip counts every step ended in vmcnt(0),
// i.e. no prefetch at all); KBC = 0 is the generic run-time fallback.  A job is either active with all NT
// tiles live (ALL) or carries a run-time live count nt.
// TAIL1: the reduction length is 16*(KBC-1) + 4 (100 here): the last k-block then holds its four columns in step 0
// of the four lane groups (k = 16*(KBC-1) + (lane >> 4), see k_pack) and costs one MFMA per tile instead of four;
// the A operand of that step is read through a_tail.
template <int NT, int G, int KBC, bool ALL, bool TAIL1 = false>
struct WaveGemm {
  const f32x4* bp[NT];
  f32x4 ring[G][NT];
  float bias[NT];      // bias of this lane's output column n = tile*16 + (lane & 15), fetched with the weights
  int tile[NT];
  int KB, nt;

  __device__ __forceinline__ void prefetch(const f32x4* __restrict__ pw, int KB_, int first_tile, int tile_stride,
                                           int n_tiles, int lane, const float* __restrict__ bias_vec = nullptr) {
    KB = KBC ? KBC : KB_;
    nt = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tile[t] = first_tile + t * tile_stride;
      if (first_tile >= 0 && tile[t] < n_tiles) nt = t + 1;
    }
    if (ALL && nt) nt = NT;
    if (nt == 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tl = (ALL || t < nt) ? tile[t] : tile[0];        // dead tiles alias a live one: loads stay valid
      bp[t] = pw + (int64_t)tl * KB * 64 + lane;
      bias[t] = bias_vec ? bias_vec[tl * 16 + (lane & 15)] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[g][t] = bp[t][(g < KB ? g : KB - 1) * 64];
    __builtin_amdgcn_sched_barrier(0);
  }

  // a_row: LDS pointer to this lane's A row (row l&15) already offset by 4*(l>>4) floats.
  // KBN: the k-blocks this call multiplies (<= KBC: the same prefetched ring serves waves whose pieces differ in length --
  // two objects, one per length, were merged by the compiler through register copies behind a wait for every load).
  template <int KBN = KBC, typename F>
  __device__ __forceinline__ void run(const float* a_row, int lane, F epilogue, const float* a_tail = nullptr) {
    static_assert(KBN <= KBC && (KBC == 0 || KBN >= G), "a piece no longer than the ring was sized for");
    if (nt == 0) return;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KBC > 0) {
#pragma unroll
      for (int kb = 0; kb < KBN; ++kb) {
        const int g = kb % G;
        const f32x4 a = *reinterpret_cast<const f32x4*>(a_row + kb * 16);
        f32x4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = ring[g][t];
        if (kb + G < KBN) {
          if (TAIL1 && kb + G == KBN - 1) {
            // one word per lane is all the tail step uses; as a dwordx4 whose upper words are dead the three loads
            // were given overlapping registers, and the next LDS read into one of them waited for vmcnt(0)
#pragma unroll
            for (int t = 0; t < NT; ++t) ring[g][t][0] = *reinterpret_cast<const float*>(&bp[t][(kb + G) * 64]);
          } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) ring[g][t] = bp[t][(kb + G) * 64];
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the refill ahead of the MFMAs (the scheduler sinks it otherwise)
        if constexpr (TAIL1) {
          if (kb == KBN - 1) {
            const float at = *a_tail;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, b[t][0], acc[t], 0, 0, 0);
            continue;
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t], 0, 0, 0);
      }
    } else {
      for (int kb0 = 0; kb0 < KB; kb0 += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int kb = kb0 + g;
          if (kb < KB) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(a_row + kb * 16);
            f32x4 b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = ring[g][t];
            const int kn = kb + G < KB ? kb + G : KB - 1;          // clamped: no branch around the loads
#pragma unroll
            for (int t = 0; t < NT; ++t) ring[g][t] = bp[t][kn * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (ALL || t < nt) {
        const float bb = bias[t];
        for_acc(acc[t], tile[t], lane, [&](int m, int n, float v) { epilogue(m, n, v + bb); });
      }
  }
};

constexpr int KC = 10;   // neighbour rows a wave keeps in registers at a time

// The K key rows [nbr(100) | edge(d_e) | time(100)] of one query, spread over the 64 lanes: lane l holds
// columns l and l+64 of the nbr and time parts and columns l, l+64, l+128 of the edge part.
template <int ME>     // ME = edge-feature registers per lane and slot: ceil(d_e / 64), or MAXE for a run-time width
struct KeyRows {
  float na[KC], nb[KC], e[KC][ME];
};
struct KeyTimes {
  float ta[KC], tb[KC];
};

struct QuerySlots {   // lane j (< K) describes neighbour slot j of this wave's query
  int nrow, erow, ovl;
  float dt;
  bool masked;
};

// Issues the global loads of slots j0 .. j0+KC-1 (all in flight at once; nothing here waits on them).
// own: bit j set = this wave handles slot j0 + j (all ones for a wave that has its query to itself; with several waves per
// query each takes every NQW-th slot and the others' registers stay zero)
// GIVEN_TIMES: the launch may be one whose time features are given (dense boundary, in.dt == nullptr); false for the
// folded-q instantiations, which the host only takes with in.dt (the per-slot test and its loads are then not compiled)
template <int ME, bool GIVEN_TIMES = true>
__device__ __forceinline__ void load_key_rows(KeyRows<ME>& R, KeyTimes& T, const orca_tgat_inputs& in, const QuerySlots& S,
                                              int64_t q, int j0, int K, int d_e, int lane, unsigned own = ~0u,
                                              int jlo = 0, int jhi = KC) {
  const int c1 = lane + 64;
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    if (j < jlo || j >= jhi) continue;       // (constants at the call: the slots of this call)
    if (j0 + j < K && ((own >> j) & 1u)) {   // wave-uniform
      const int nr = __builtin_amdgcn_readlane(S.nrow, j0 + j);
      const int er = __builtin_amdgcn_readlane(S.erow, j0 + j);
      const int ov = __builtin_amdgcn_readlane(S.ovl, j0 + j);
      const float* np = (ov ? in.nbr_overlay : in.nbr_table) + (int64_t)nr * D;
      const float* ep = in.edge_table + (int64_t)er * d_e;
      // Columns past the row's width load a clamped (finite) element instead of being skipped: a lane-conditional load
      // is a branch, and behind a branch every load is waited for where it is issued -- the ten slots became ten
      // dependent round trips.  The surplus values meet zero weights (q~ is zero there) and are never stored.
      R.na[j] = np[lane];
      R.nb[j] = np[c1 < D ? c1 : 0];
#pragma unroll
      for (int i = 0; i < ME; ++i) {
        const int c = lane + 64 * i;
        R.e[j][i] = ep[c < d_e ? c : 0];
      }
      if (GIVEN_TIMES && !in.dt) {     // dense boundary: time features are given
        const float* tp = in.nbr_time + (q * K + j0 + j) * D;
        T.ta[j] = tp[lane];
        T.tb[j] = tp[c1 < D ? c1 : 0];
      }
    } else {
      R.na[j] = R.nb[j] = 0.f;
#pragma unroll
      for (int i = 0; i < ME; ++i) R.e[j][i] = 0.f;
      if (GIVEN_TIMES && !in.dt) T.ta[j] = T.tb[j] = 0.f;
    }
  }
}

// cos(dt*w + b) for the chunk's slots (model/time_encoding.py:26-27), kept in registers
// (wa, ba / wb, bb: time_w, time_b of this lane's columns lane and lane + 64 -- lanes past the width carry the values of
// column 0 and compute a surplus value: no lane-conditional code)
template <bool GIVEN_TIMES = true>
__device__ __forceinline__ void encode_key_times(KeyTimes& T, const orca_tgat_inputs& in, const QuerySlots& S, int j0,
                                                 int K, int lane, float wa, float ba, float wb, float bb, unsigned own = ~0u,
                                                 int jlo = 0, int jhi = KC) {
  if (GIVEN_TIMES && !in.dt) return;
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    if (j < jlo || j >= jhi) continue;       // (constants at the call: the slots of this call)
    if (j0 + j < K && ((own >> j) & 1u)) {
      const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S.dt), j0 + j));
      T.ta[j] = orca_cosf(orca_time_arg(d, wa, ba));
      T.tb[j] = orca_cosf(orca_time_arg(d, wb, bb));
    } else {
      T.ta[j] = T.tb[j] = 0.f;
    }
  }
}

// ---- 2 x KC wave-wide sums in 22 shuffles instead of 2*KC*6 -------------------------------------------
// Recursive halving: at xor-distance d each lane keeps one half of its value list and adds the partner's
// copy of that half, so the list shrinks 20 -> 10 -> 5 -> 3 -> 2 -> 1 while every value keeps summing over
// twice as many lanes.  The full sum of (head h, slot j) ends in the two lanes holder(h, j) | {0, 1}.
__host__ __device__ constexpr int holder_lane(int h, int j) {
  int b4 = j >= 5, a = j - 5 * b4;
  int b3 = a >= 3, b = a - 3 * b3;
  int b2 = b == 2, c = b2 ? 0 : b;
  return 32 * h + 16 * b4 + 8 * b3 + 4 * b2 + 2 * c;
}
// slot held by this lane (ignoring the head bit), or -1 for a padding lane
__device__ __forceinline__ int held_slot(int lane) {
  const int b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1, b1 = (lane >> 1) & 1;
  if ((b2 && b1) || (b3 && b2)) return -1;
  return 5 * b4 + 3 * b3 + (b2 ? 2 : b1);
}
__device__ __forceinline__ float reduce_2xKC(const float (&p0)[KC], const float (&p1)[KC], int lane) {
  static_assert(KC == 10, "the halving schedule below is written for 10 slots");
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
  // (every exchange is a vector instruction -- common.h, lane exchanges -- and the two widest steps need no select: the
  // swap leaves `kept` and `partner's copy` in the two registers)
  float w[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) { float a = p0[j], b = p1[j]; lanes_swap32(a, b); w[j] = a + b; }
  float x[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) { float a = w[j], b = w[5 + j]; lanes_swap16(a, b); x[j] = a + b; }
  float y[3];
  y[0] = (b3 ? x[3] : x[0]) + lanes_xor8(b3 ? x[0] : x[3]);
  y[1] = (b3 ? x[4] : x[1]) + lanes_xor8(b3 ? x[1] : x[4]);
  y[2] = (b3 ? 0.f : x[2]) + lanes_xor8(b3 ? x[2] : 0.f);
  float z[2];
  z[0] = (b2 ? y[2] : y[0]) + lanes_xor4(b2 ? y[0] : y[2]);
  z[1] = (b2 ? 0.f : y[1]) + lanes_xor4(b2 ? y[1] : 0.f);
  float u = (b1 ? z[1] : z[0]) + lanes_xor2(b1 ? z[0] : z[1]);
  return u + lanes_xor1(u);
}
__device__ __forceinline__ float half_max(float v) {   // over the 32-lane half this lane is in
  v = lanes_max_xor16(v);
  v = fmaxf(v, lanes_xor8(v));
  v = fmaxf(v, lanes_xor4(v));
  v = fmaxf(v, lanes_xor2(v));
  return fmaxf(v, lanes_xor1(v));
}
__device__ __forceinline__ float half_sum(float v) {
  v = lanes_add_xor16(v);
  v += lanes_xor8(v);
  v += lanes_xor4(v);
  v += lanes_xor2(v);
  return v + lanes_xor1(v);
}

// KT = compile-time neighbour count (10, the reference's default n_degree: every per-slot predicate and
// the second-chunk code fold away) or 0 = run-time K <= ORCA_MAX_K.
// DE = compile-time edge width (172, the Wikipedia / Reddit / benchmark width: every k-block count of the
// projection chains becomes a constant) or 0 = run-time d_e <= ORCA_MAX_DE.
// NQW = waves per query.  1: a wave has its query to itself and the workgroup holds TQ = 16 queries -- the form for a batch
// that fills the chip (one workgroup per CU at 4095 queries).  4: the workgroup holds 4 queries (still one MFMA M-tile in
// the GEMM stages: twelve of its rows idle), and the four waves of a query split its K slots in the per-query stages
// (time encodings, key rows, scores, weighted sums; partial results meet in LDS) -- the form for the reference's own batch
// size, where 600 queries are 38 workgroups on 256 CUs and what a step costs is ONE workgroup's latency: the per-query
// stages are a quarter as long, and four times as many CUs stream the weights.  K <= KC, gather mode with a folded q, no
// saving of intermediates; sums over the slots are taken in another order than NQW = 1 (same bars, not the same bits).
// PROF: which of the optional inputs and fused writes a launch carries, as a compile-time fact (tgat_profile() on the host
// derives it from the arguments; 0 = anything goes).  The kernel tests some twenty pointers of orca_tgat_inputs /
// orca_tgat_side at run time; with the answers known the compiler drops a sixth of the instructions of the bench shape's
// kernel.  A profile is 16 + a set of bits: 1 = gather mode over the memory table with ids, flags and overlays of source
// and neighbours (a first layer; without it: dense source rows and a flag-less neighbour table, a later layer), 2 = commits
// the flagged source rows (persist), 4 = pushes its winners' rows, 8 = stores the raw messages.  All profiles: neighbour,
// edge and id arrays and computed times given; no saved intermediates, attention weights, source rows out, dropout or
// device-side query count.  Instantiated: 16|7 the step driver's first layer of several, 16|3 the only layer of a
// one-layer model, 16|8 its last layer, 16|1 and 16|0 the first and later layers of the owner-computes (multi-rank) step.
template <int KT, int DE, bool QK, int NQW = 1, int PROF = 0>
__global__ __launch_bounds__(NW * 64) void k_tgat_forward(const float* __restrict__ blob, int d_e_rt,
                                                          orca_tgat_inputs in, int64_t Q, int K_rt,
                                                          float* __restrict__ out, float* __restrict__ out_w,
                                                          orca_tgat_side side, long long* __restrict__ phase_clock) {
  extern __shared__ __align__(16) float lds[];
  // optional profiling aid (orca_debug_tgat_phase_clock): wave 0 of each workgroup stamps the 100 MHz
  // real-time counter at every phase boundary (words 0..15 of its 32) and the shader clock beside it (16..31)
#ifdef ORCA_WAVE_STAMPS   // diagnostic build only (tools/build_diag.sh): every wave's shader clock around the GEMM stages,
                          // behind the workgroups' 32 words each: [workgroup][wave][8]
  auto wstamp = [&](int i) {
    if (phase_clock && (threadIdx.x & 63) == 0)
      phase_clock[(int64_t)gridDim.x * 32 + ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 8 + i] =
          (long long)__builtin_amdgcn_s_memtime();
  };
#else
  auto wstamp = [](int) {};
#endif
  int phase_no = 0;
  auto stamp = [&]() {
    if (phase_clock && threadIdx.x == 0) {
      phase_clock[(int64_t)blockIdx.x * 32 + phase_no] = wall_clock64();
      phase_clock[(int64_t)blockIdx.x * 32 + 16 + phase_no] = (long long)__builtin_amdgcn_s_memtime();
    }
    ++phase_no;
  };
  stamp();
  if constexpr (PROF != 0) {        // (what tgat_profile() has checked on the host)
    static_assert((PROF & ~15) == 16, "16 + profile bits");
    __builtin_assume(in.nbr_idx != nullptr); __builtin_assume(in.edge_idx != nullptr); __builtin_assume(in.nbr_ids != nullptr);
    __builtin_assume(in.dt != nullptr); __builtin_assume(side.src_rows_out == nullptr); __builtin_assume(side.save_qt == nullptr);
    __builtin_assume(side.save_a == nullptr); __builtin_assume(side.save_xbar == nullptr); __builtin_assume(out_w == nullptr);
    __builtin_assume(!(in.dropout_p > 0.f)); __builtin_assume(side.q_extra == nullptr);
    if constexpr (PROF & 1) {
      __builtin_assume(in.src_idx != nullptr); __builtin_assume(in.src_flags != nullptr); __builtin_assume(in.nbr_flags != nullptr);
    } else {
      __builtin_assume(in.src_idx == nullptr); __builtin_assume(in.src_flags == nullptr); __builtin_assume(in.nbr_flags == nullptr);
    }
    if constexpr (PROF & 2) __builtin_assume(side.persist_memory != nullptr); else __builtin_assume(side.persist_memory == nullptr);
    if constexpr (PROF & 4) __builtin_assume(side.push_table != nullptr); else __builtin_assume(side.push_table == nullptr);
    if constexpr (PROF & 8) __builtin_assume(side.store_bs > 0); else __builtin_assume(side.store_bs == 0);
  }
  if (side.q_extra) {               // block-uniform, ahead of every barrier
    const int64_t lim = side.q_base + *side.q_extra;
    Q = lim < Q ? lim : Q;
    if ((int64_t)blockIdx.x * (TQ / NQW) >= Q) return;
  }
  if (side.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *side.zero_word = 0;
  const int K = KT ? KT : K_rt;
  const int d_e = DE ? DE : d_e_rt;
  constexpr Blob LC = blob_layout(DE ? DE : 1);   // compile-time layout when DE is fixed
  const Blob L = DE ? LC : blob_layout(d_e);
  constexpr int KBT_C = DE ? LC.KBT : 0;         // k-blocks of the folded tail (0 = run time)
  constexpr bool FOLD_ALL = DE && LC.NTk == 24;  // every (head, slot) wave of P2 owns exactly 3 live tiles
  constexpr int ME = DE ? (DE + 63) / 64 : MAXE;  // edge-feature registers per lane and slot
  const int kdim = L.kdim;
  const int KPS = L.KP;                           // columns of one head's q~ / x-bar
  const int RS = L.KBT * 16 + 4;                  // row stride of the tail input [x-bar_0 | x-bar_1 | src | 0]
  float* bufQ = lds;                              // [TQ][ASTR]  query vector; tail partials; fc2 output
  float* bufP = bufQ + TQ * ASTR;                 // [TQ][ASTR]  0.1*q; tail partials (continued)
  float* bufM = bufP + TQ * ASTR;                 // [TQ][HSTR]  relu(fc1)
  float* bufT = bufM + TQ * HSTR;                 // [TQ][RS]    q~ then x-bar (both heads), src, zero padding

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;                      // wave m owns query q0 + m in the per-query stages
  constexpr int TQW = TQ / NQW;                    // queries of this workgroup
  static_assert(NQW == 1 || (KT != 0 && KT <= KC && QK), "several waves per query: compile-time K <= KC, folded q");
  const int64_t q0 = (int64_t)blockIdx.x * TQW;
  const int qm = wave / NQW;                       // this wave's query (row of the M-tile) ...
  // ... and its part of the slots: qr, qr + NQW, ... (a scalar; spelled as a constant for NQW = 1, or the default form pays
  // for branches that can never be taken)
  const int qr = NQW == 1 ? 0 : __builtin_amdgcn_readfirstlane(wave % NQW);
  const bool lead = NQW == 1 || qr == 0;           // the wave that does what is done once per query
  constexpr int NSL = (KC + NQW - 1) / NQW;        // slots of a wave when several share a query
  // LDS row this wave fills in the staging stage: its query's row, or (not the lead) one of the idle rows, with zeros
  const int lrow = NQW == 1 ? wave : (lead ? qm : TQW + wave - qm - 1);
  const f32x4* blob4 = reinterpret_cast<const f32x4*>(blob);
  const int c1 = lane + 64;
  const int a_off = (lane & 15), a_k = 4 * (lane >> 4);
  const int NT200 = pw_tiles(QD), NT100 = pw_tiles(D);

  // QK: gather mode with a blob completed by orca_tgat_pack_gather: q~ comes straight from src (no P1)
  constexpr bool qk = QK;
  const int h2 = wave & 1;                        // P2: wave (h2, slot = wave >> 1)
  WaveGemm<1, 7, 13, true> g1;
  WaveGemm<3, 3, 7, FOLD_ALL, true> g2;      // reduction length 100 = 6 k-blocks + 4 columns

  // ---- T0: this wave's query: neighbour slots, source row, L2 touches, time encodings ------------
  // The stage is a chain of dependent gathers, and what it costs is the number of round trips on that chain.  Its loads
  // are therefore issued LEVEL BY LEVEL -- everything that depends on the query number alone, then everything that depends
  // on those values, then the row touches -- and unconditionally: an optional array that is absent is replaced by a valid
  // address (the weight blob) and its value discarded, dead waves and lanes load clamped positions.  Written with an `if`
  // per optional input, every load was waited for where it was issued: ten serial round trips instead of three.
  const int64_t q = q0 + qm;
  const bool live_q = q < Q;
  const int64_t qc = live_q ? q : Q - 1;
  const bool slot = live_q && lane < K;
  const int64_t pc = qc * K + (lane < K ? lane : 0);
  const int32_t* any32 = reinterpret_cast<const int32_t*>(blob);
  const uint8_t* any8 = reinterpret_cast<const uint8_t*>(blob);
  const int cb = c1 < D ? c1 : 0;                 // (column c1 is stored only where c1 < D)
  // level 1
  const int nr1 = (in.nbr_idx ? in.nbr_idx : any32)[in.nbr_idx ? pc : 0];
  const int er1 = (in.edge_idx ? in.edge_idx : any32)[in.edge_idx ? pc : 0];
  const int id1 = (in.nbr_ids ? in.nbr_ids : any32)[in.nbr_ids ? pc : 0];
  const uint8_t mk1 = (in.nbr_ids ? any8 : in.mask)[in.nbr_ids ? 0 : pc];
  const float dt1 = (in.dt ? in.dt : blob)[in.dt ? pc : 0];
  const int sr1 = (in.src_idx ? in.src_idx : any32)[in.src_idx ? qc : 0];
  const float* twp = in.dt ? in.time_w : blob;
  const float* tbp = in.dt ? in.time_b : blob;
  const float tw_a = twp[lane], tb_a = tbp[lane], tw_b = twp[cb], tb_b = tbp[cb];
  // (fused raw-message store of position q, common.h: its batch arrays are read with the slots)
  const int64_t st_bs = side.store_bs;
  const bool st_on = st_bs > 0, st_q = st_on && lead && q < 2 * st_bs;
  const int64_t sq = st_q ? q : 0, sp = st_q ? (q < st_bs ? q + st_bs : q - st_bs) : 0, se = st_q ? (q < st_bs ? q : q - st_bs) : 0;
  const int64_t* st_nodes = st_on ? side.store_nodes : reinterpret_cast<const int64_t*>(blob);
  const uint8_t st_w = (st_on ? side.store_winner : any8)[sq];
  const int64_t st_id = st_nodes[sq], st_partner = st_nodes[sp];
  const int64_t st_eidx = (st_on ? side.store_eidx : reinterpret_cast<const int64_t*>(blob))[se];
  const double st_t = (st_on ? side.store_ts : reinterpret_cast<const double*>(blob))[sq];
  QuerySlots S{0, 0, 0, 0.f, true};
  if (slot) {
    S.nrow = in.nbr_idx ? nr1 : (int)pc;
    S.erow = in.edge_idx ? er1 : (int)pc;
    S.masked = in.nbr_ids ? (id1 == 0) : (mk1 != 0);
    if (in.dt) S.dt = dt1;
  }
  const int64_t row = in.src_idx ? (int64_t)sr1 : qc;
  // level 2: the overlay flags of the neighbours and of the source, BOTH candidate source rows (blended bit-wise below:
  // flag -> table -> row would be one more round trip), the persist's timestamp
  const uint8_t nf1 = (in.nbr_flags ? in.nbr_flags : any8)[in.nbr_flags ? S.nrow : 0];
  const uint8_t sf1 = (in.src_flags ? in.src_flags : any8)[in.src_flags ? row : 0];
  const float* s_ovl = in.src_flags ? in.src_overlay : in.src_table;
  const float ta_ = in.src_table[row * D + lane], tb_ = in.src_table[row * D + cb];
  const float oa_ = s_ovl[row * D + lane], ob_ = s_ovl[row * D + cb];
  const float pts1 = (side.persist_memory ? side.persist_msg_ts : blob)[side.persist_memory ? row : 0];
  const bool st_do = st_q && st_w && st_id >= 0;
  constexpr int MME = DE ? ME : 1;      // (a run-time edge width keeps message_store_wave's own loads: MAXE registers spill)
  float m_lu = 0.f, m_sa = 0.f, m_sb = 0.f, m_da = 0.f, m_db = 0.f, m_e[MME];
#pragma unroll
  for (int i = 0; i < MME; ++i) m_e[i] = 0.f;
  if (DE != 0 && st_on) {    // launch-uniform; a wave without a message to store reads the rows of node 0
    const int64_t idc = st_do ? st_id : 0, pac = st_do ? st_partner : 0, ec = st_do ? st_eidx : 0;
    m_lu = side.store_last_update[idc];
    m_sa = side.store_memory[idc * D + lane]; m_sb = side.store_memory[idc * D + cb];
    m_da = side.store_memory[pac * D + lane]; m_db = side.store_memory[pac * D + cb];
#pragma unroll
    for (int i = 0; i < MME; ++i) {
      const int c = lane + 64 * i;
      m_e[i] = in.edge_table[ec * d_e + (c < d_e ? c : 0)];
    }
  }
  // The time encodings need only dt (level 1): the first EARLY slots' are computed HERE, under the round trip of level 2 --
  // behind the touches of level 3, which wait for it, the stage was two round trips and then twenty cosines per lane
  // (2.2 us of the SIMD's vector time) end to end.
  // (Same box, per step: none early 75.4 us, five slots 74.8, six 74.5, seven 74.2, eight 74.3, all ten 75.3.)
  KeyTimes T;
  float lta[NSL], ltb[NSL];                      // (several waves per query: the encodings of this wave's slots)
  constexpr int EARLY = NQW == 1 ? 7 : 0;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NQW == 1) {
    if (live_q) encode_key_times<!QK>(T, in, S, 0, K, lane, tw_a, tb_a, tw_b, tb_b, ~0u, 0, EARLY);
  } else {
#pragma unroll
    for (int jl = 0; jl < NSL; ++jl) {
      const int j = qr + NQW * jl;
      lta[jl] = ltb[jl] = 0.f;
      if (live_q && j < K) {
        const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S.dt), j));
        lta[jl] = orca_cosf(orca_time_arg(d, tw_a, tb_a));
        ltb[jl] = orca_cosf(orca_time_arg(d, tw_b, tb_b));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (in.nbr_flags && slot) S.ovl = nf1 ? 1 : 0;
  // source row [src(100) | src_time(100)] of query m = wave and the merger's copy of src
  float sa = 0.f, sb = 0.f, sta = 0.f, stb = 0.f;
  if (live_q) {
    const unsigned m = (in.src_flags && sf1) ? ~0u : 0u;
    sa = __uint_as_float((__float_as_uint(oa_) & m) | (__float_as_uint(ta_) & ~m));
    sb = __uint_as_float((__float_as_uint(ob_) & m) | (__float_as_uint(tb_) & ~m));
    if (lead && side.persist_memory && q < side.persist_n && m != 0u) {   // fused memory persist (common.h)
      side.persist_memory[row * D + lane] = sa;
      if (c1 < D) side.persist_memory[row * D + c1] = sb;
      if (lane == 0) side.persist_last_update[row] = pts1;
    }
    if (lead && side.src_rows_out) {
      side.src_rows_out[q * D + lane] = sa;
      if (c1 < D) side.src_rows_out[q * D + c1] = sb;
    }
    if constexpr (qk) {
      // the source's time encoding is a constant already inside the folded map: nothing to compute
    } else if (in.dt) {    // embedding_module.py:126: the source's own time encoding is cos(0*w + b)
      sta = orca_cosf(orca_time_arg(0.0f, tw_a, tb_a));
      stb = c1 < D ? orca_cosf(orca_time_arg(0.0f, tw_b, tb_b)) : 0.f;
    } else {
      sta = in.src_time[q * D + lane];
      stb = c1 < D ? in.src_time[q * D + c1] : 0.f;
    }
  }
  // level 3: touch every 128-B line of the K neighbour and edge rows now (one dword per line, two VGPRs held):
  // the lines travel HBM -> L2 while the projection phases run, and the row loads of P3 then hit L2
  float touch0 = 0.f, touch1 = 0.f;
  if (live_q) {
    const int nl = (D * 4 + 127) / 128, el = (d_e * 4 + 127) / 128;   // lines per nbr / edge row
    const int per = nl + el, total = K * per;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int idx = lane + 64 * r;
      if constexpr (NQW == 1) {
        if (idx < total) {
          const int j = idx / per, ln = idx - j * per;
          const int nr = __shfl(S.nrow, j, 64), er = __shfl(S.erow, j, 64), ov = __shfl(S.ovl, j, 64);
          const float* p = ln < nl ? (ov ? in.nbr_overlay : in.nbr_table) + (int64_t)nr * D + ln * 32
                                   : in.edge_table + (int64_t)er * d_e + (ln - nl) * 32;
          const float v = *p;
          if (r == 0) touch0 = v; else touch1 = v;
        }
      } else {
        const int j = idx < total ? idx / per : 0, ln = idx - j * per;
        // (every lane takes part in the shuffles: a pulled-from lane must be active, and this wave's slots are not 0..)
        const int nr = __shfl(S.nrow, j, 64), er = __shfl(S.erow, j, 64), ov = __shfl(S.ovl, j, 64);
        if (idx < total && j % NQW == qr) {
          const float* p = ln < nl ? (ov ? in.nbr_overlay : in.nbr_table) + (int64_t)nr * D + ln * 32
                                   : in.edge_table + (int64_t)er * d_e + (ln - nl) * 32;
          const float v = *p;
          if (r == 0) touch0 = v; else touch1 = v;
        }
      }
    }
  }
  // the message itself (model/tgn_model.py:172-191 + modules/memory.py:25-28, as message_store_wave builds it), while
  // the touches are in flight
  if (DE == 0 && st_do) {
    message_store_wave(st_id, st_partner, st_eidx, st_t, side.store_memory, side.store_last_update, in.edge_table, d_e,
                       in.time_w, in.time_b, side.store_messages, side.store_msg_ts, side.store_pending, lane);
  } else if (st_do) {
    float* mrow = side.store_messages + st_id * (3 * D + d_e);
    const float tf = (float)st_t;
    const float delta = tf - m_lu;
    mrow[lane] = m_sa;
    mrow[D + lane] = m_da;
    mrow[2 * D + d_e + lane] = orca_cosf(orca_time_arg(delta, tw_a, tb_a));
    if (c1 < D) {
      mrow[c1] = m_sb;
      mrow[D + c1] = m_db;
      mrow[2 * D + d_e + c1] = orca_cosf(orca_time_arg(delta, tw_b, tb_b));
    }
#pragma unroll
    for (int i = 0; i < MME; ++i) {
      const int c = lane + 64 * i;
      if (c < d_e) mrow[2 * D + c] = m_e[i];
    }
    if (lane == 0) {
      side.store_msg_ts[st_id] = tf;
      side.store_pending[st_id] = 1;
    }
  }
  // The first weights of the next GEMM stage are requested HERE, behind the gathers of this stage and ahead of its
  // cosines: loads return in order, so 144 KB of weights requested at kernel entry stood between every dependent
  // level of the gathers above and its consumer.
  if constexpr (qk)
    g2.prefetch(blob4 + (L.pw_qk + (int64_t)h2 * pw_floats(kdim + 1, D)) / 4, NT100, wave >> 1, NW / 2, L.NTk, lane,
                blob + L.b_qk + h2 * KPS);
  else
    g1.prefetch(blob4 + L.pw_q / 4, NT200, wave < NT200 ? wave : -1, NW, NT200, lane, blob + L.b_q);
  // the other slots' time encodings: all VALU, computed while the loads above are in flight
  if constexpr (NQW == 1) {
    if (live_q) encode_key_times<!QK>(T, in, S, 0, K, lane, tw_a, tb_a, tw_b, tb_b, ~0u, EARLY, KC);
  }
  if (!lead) { sa = sb = sta = stb = 0.f; }       // (an idle row of the M-tile)
  {
    const int m = lrow;
    bufQ[m * ASTR + lane] = sa;
    bufQ[m * ASTR + D + lane] = sta;
    float* trow = bufT + m * RS + 2 * KPS;        // the merger's copy of src, behind the two x-bar blocks
    trow[lane] = sa;
    if (c1 < D) {
      bufQ[m * ASTR + c1] = sb;
      bufQ[m * ASTR + D + c1] = stb;
      trow[c1] = sb;
    }
    if (lane < ASTR - QD) bufQ[m * ASTR + QD + lane] = 0.f;
    for (int c = D + lane; c < RS - 2 * KPS; c += 64) trow[c] = 0.f;
    for (int c = lane; c < ASTR; c += 64) bufP[m * ASTR + c] = 0.f;
  }
  __syncthreads();
  stamp();

  // ---- P1: q = Wq [src|time] + bq, scaled by sqrt(1/head_dim) -> bufP ---------------------------
  if constexpr (!qk) {
    g1.run(bufQ + a_off * ASTR + a_k, lane, [&](int m, int n, float v) {
      if (n < ASTR) bufP[m * ASTR + n] = (n < QD) ? v * 0.1f : 0.0f;
    });
    g2.prefetch(blob4 + (L.pw_kt + (int64_t)h2 * pw_floats(kdim + 1, D)) / 4, NT100, wave >> 1, NW / 2, L.NTk, lane);
    __syncthreads();
  }
  stamp();
  // ---- P2: q~_h = Wk_h^T (0.1 q_h)  (column kdim carries (0.1 q_h).bk_h): wave (h = w&1, slot = w>>1)
  //      owns tiles slot, slot+8, slot+16 of head h -> bufT.  With the folded map the operand is src itself
  //      (the columns 100..111 of bufQ hold finite time values that meet zero weights).
  // The first key rows of P3 are requested ahead of the barrier that ends the key fold: every wave leaves that barrier at
  // the same moment and would otherwise wait for its first row with nothing else to do; the rest follow behind the
  // barrier, under the arithmetic on these.  (Same box, per step: none 76.29 us, 2 slots 75.77, 3 slots 75.53, 5 slots
  // 75.50, all ten 76.23; a slot requested ahead of the fold's MFMA loop as well: no gain, 3 spills with two.  The
  // several-waves-per-query form keeps its loads behind the barrier: ahead of it config 2's step is 56.8 instead of 56.3 us.)
  constexpr int PRE = (NQW == 1 && QK && KT != 0 && KT <= KC) ? 3 : 0;
  KeyRows<ME> R;
  {
    float* dstT = bufT + h2 * KPS;
    const float* a2 = (qk ? bufQ : bufP + h2 * D) + a_off * ASTR;
    wstamp(0);
    g2.run(a2 + a_k, lane, [&](int m, int n, float v) { dstT[m * RS + n] = v; }, a2 + 96 + (lane >> 4));
    wstamp(1);
  }
  if (PRE && live_q) load_key_rows<ME, !QK>(R, T, in, S, q, 0, K, d_e, lane, ~0u, 0, PRE);
  wstamp(2);
  __syncthreads();
  wstamp(3);
  stamp();

  // ---- P3: this wave's query: scores, masked softmax, weighted key sum ---------------------------
  asm volatile("" ::"v"(touch0), "v"(touch1));   // the touches are complete (and kept) up to here
  if constexpr (NQW == 1)
  {
    const int m = wave;
    float* xb0 = bufT + m * RS;
    float* xb1 = xb0 + KPS;
    if (!live_q) {
      for (int c = lane; c < 2 * KPS; c += 64) xb0[c] = 0.0f;
    } else {
      load_key_rows<ME, !QK>(R, T, in, S, q, 0, K, d_e, lane, ~0u, PRE, KC);
      unsigned long long livebits = __ballot(lane < K && !S.masked);
      const bool invalid = livebits == 0ull;                           // temporal_attention.py:46
      if (invalid) {
        livebits = 1ull;                                                 // :47 un-masks slot 0
        if (!in.nbr_ids && lane == 0) in.mask[q * K] = 0;
      }
      // this lane's columns of q~ (both heads): nbr part c, c+64; edge part; time part c, c+64
      const int tb = D + d_e;
      const float qn0a = xb0[lane], qn1a = xb1[lane];
      const float qn0b = c1 < D ? xb0[c1] : 0.f, qn1b = c1 < D ? xb1[c1] : 0.f;
      float qe0[ME], qe1[ME];
#pragma unroll
      for (int i = 0; i < ME; ++i) {
        const int c = lane + 64 * i;
        qe0[i] = c < d_e ? xb0[D + c] : 0.f;
        qe1[i] = c < d_e ? xb1[D + c] : 0.f;
      }
      const float qt0a = xb0[tb + lane], qt1a = xb1[tb + lane];
      const float qt0b = c1 < D ? xb0[tb + c1] : 0.f, qt1b = c1 < D ? xb1[tb + c1] : 0.f;
      const float off = (lane & 32) ? xb1[kdim] : xb0[kdim];
      // (orca_tgat_forward_saving) this wave's row of q~ / x-bar, LDS -> [Q][2][kdim], three 16-byte stores per lane
      auto save_row = [&](float* dst) {
        float* g = dst + q * 2 * kdim;
        if ((kdim & 3) == 0) {
          const int n4 = kdim >> 2;
          for (int i = lane; i < 2 * n4; i += 64) {
            const int h = i >= n4 ? 1 : 0, c = i - h * n4;
            *reinterpret_cast<f32x4*>(g + h * kdim + 4 * c) = *reinterpret_cast<const f32x4*>(xb0 + h * KPS + 4 * c);
          }
        } else {
          for (int i = lane; i < 2 * kdim; i += 64) {
            const int h = i >= kdim ? 1 : 0, c = i - h * kdim;
            g[h * kdim + c] = xb0[h * KPS + c];
          }
        }
      };
      if (side.save_qt) save_row(side.save_qt);      // wave-uniform
      // per-lane partial dot products of the chunk's rows with q~, reduced 20 at a time; afterwards the
      // lanes holder_lane(h, j) | {0,1} hold score(h, j0 + j)
      auto scores = [&]() {
        float p0[KC], p1[KC];
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          p0[j] = R.na[j] * qn0a + R.nb[j] * qn0b + T.ta[j] * qt0a + T.tb[j] * qt0b;
          p1[j] = R.na[j] * qn1a + R.nb[j] * qn1b + T.ta[j] * qt1a + T.tb[j] * qt1b;
#pragma unroll
          for (int i = 0; i < ME; ++i) {
            p0[j] += R.e[j][i] * qe0[i];
            p1[j] += R.e[j][i] * qe1[i];
          }
        }
        return reduce_2xKC(p0, p1, lane) + off;
      };
      const int hs = held_slot(lane);
      float sc[2];
      sc[0] = scores();
      if (K > KC) {
        load_key_rows<ME, !QK>(R, T, in, S, q, KC, K, d_e, lane);
        encode_key_times<!QK>(T, in, S, KC, K, lane, tw_a, tb_a, tw_b, tb_b);
        sc[1] = scores();
      } else {
        sc[1] = 0.f;
      }
      // masked softmax per head = per 32-lane half; every real slot sits in two lanes of its half
      const bool lv0 = hs >= 0 && ((livebits >> hs) & 1ull);
      const bool lv1 = K > KC && hs >= 0 && ((livebits >> (hs + KC)) & 1ull);
      const float mx = half_max(fmaxf(lv0 ? sc[0] : -INFINITY, lv1 ? sc[1] : -INFINITY));
      const float e0 = lv0 ? expf(sc[0] - mx) : 0.f, e1 = lv1 ? expf(sc[1] - mx) : 0.f;
      // an all-padded neighbourhood has attn = 0 (:52): its weights, hence x-bar, are zeroed (out_w too, below)
      const float inv = invalid ? 0.f : 1.0f / (half_sum(e0 + e1) * 0.5f);
      float a_c0 = e0 * inv, a_c1 = e1 * inv;            // attention weight of (head = lane>>5, slot hs [+KC])
      if (side.save_a) {             // the softmax output itself, before dropout: [q][head][slot]
        const int jj = lane < K ? (lane < KC ? lane : lane - KC) : 0;
        int hl = 0;
#pragma unroll
        for (int j = 0; j < KC; ++j) hl = (jj == j) ? holder_lane(0, j) : hl;
        float w0 = __shfl(a_c0, hl, 64), w1 = __shfl(a_c0, hl + 32, 64);
        if (K > KC) {
          const float v0 = __shfl(a_c1, hl, 64), v1 = __shfl(a_c1, hl + 32, 64);
          if (lane >= KC) { w0 = v0; w1 = v1; }
        }
        if (lane < K) {
          side.save_a[(q * 2 + 0) * K + lane] = w0;
          side.save_a[(q * 2 + 1) * K + lane] = w1;
        }
        if (lane == 0) side.save_invalid[q] = invalid ? 1.f : 0.f;
      }
      float wsum = invalid ? 0.f : 1.f;                  // sum of this head's weights: 1 without dropout
      if (in.dropout_p > 0.f) {                          // nn.MultiheadAttention's dropout on the softmax output
        const float sc_ = 1.0f / (1.0f - in.dropout_p);
        const int hh = hs < 0 ? 0 : hs;
        a_c0 *= orca_dropout_mult(in.dropout_seed, in.dropout_q0 + q, lane >> 5, hh, in.dropout_p, sc_);
        a_c1 *= orca_dropout_mult(in.dropout_seed, in.dropout_q0 + q, lane >> 5, hh + KC, in.dropout_p, sc_);
        wsum = half_sum(a_c0 + a_c1) * 0.5f;             // (a_c* are zero outside the real slots)
      }
      if (out_w) {                                                       // :51-53 (head average); wave-uniform
        const int jj = lane < K ? (lane < KC ? lane : lane - KC) : 0;
        int hl = 0;
#pragma unroll
        for (int j = 0; j < KC; ++j) hl = (jj == j) ? holder_lane(0, j) : hl;
        // every lane takes part in the shuffles (a pulled-from lane must be active)
        float w0 = __shfl(a_c0, hl, 64), w1 = __shfl(a_c0, hl + 32, 64);
        if (K > KC) {
          const float v0 = __shfl(a_c1, hl, 64), v1 = __shfl(a_c1, hl + 32, 64);
          if (lane >= KC) { w0 = v0; w1 = v1; }
        }
        if (lane < K) out_w[q * K + lane] = invalid ? 0.f : (w0 + w1) * 0.5f;
      }
      // x-bar_h = sum_j a_hj x_j
      float an0 = 0.f, an1 = 0.f, bn0 = 0.f, bn1 = 0.f, at0 = 0.f, at1 = 0.f, bt0 = 0.f, bt1 = 0.f;
      float ae0[ME], ae1[ME];
#pragma unroll
      for (int i = 0; i < ME; ++i) { ae0[i] = 0.f; ae1[i] = 0.f; }
      auto accumulate = [&](float a_c) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_c), holder_lane(0, j)));
          const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_c), holder_lane(1, j)));
          an0 += w0 * R.na[j]; an1 += w1 * R.na[j]; bn0 += w0 * R.nb[j]; bn1 += w1 * R.nb[j];
          at0 += w0 * T.ta[j]; at1 += w1 * T.ta[j]; bt0 += w0 * T.tb[j]; bt1 += w1 * T.tb[j];
#pragma unroll
          for (int i = 0; i < ME; ++i) { ae0[i] += w0 * R.e[j][i]; ae1[i] += w1 * R.e[j][i]; }
        }
      };
      if (K > KC) {          // the registers hold slots KC..K-1: use them, then fetch slots 0..KC-1 again (L2 hits)
        accumulate(a_c1);
        load_key_rows<ME, !QK>(R, T, in, S, q, 0, K, d_e, lane);
        encode_key_times<!QK>(T, in, S, 0, K, lane, tw_a, tb_a, tw_b, tb_b);
      }
      accumulate(a_c0);
      xb0[lane] = an0; xb1[lane] = an1;
      if (c1 < D) { xb0[c1] = bn0; xb1[c1] = bn1; }
#pragma unroll
      for (int i = 0; i < ME; ++i) {
        const int c = lane + 64 * i;
        if (c < d_e) { xb0[D + c] = ae0[i]; xb1[D + c] = ae1[i]; }
      }
      xb0[tb + lane] = at0; xb1[tb + lane] = at1;
      if (c1 < D) { xb0[tb + c1] = bt0; xb1[tb + c1] = bt1; }
      // constant columns of the tail: each head's value-bias term times its weight sum, the out-projection bias once
      const float ws1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wsum), 32));
      if (lane == 0) { xb0[kdim] = wsum; xb1[kdim] = ws1; xb0[2 * KPS + D] = invalid ? 0.f : 1.f; }
      if (side.save_xbar) {
        if (lane == 0) { side.save_wsum[q * 2] = wsum; side.save_wsum[q * 2 + 1] = ws1; }
        save_row(side.save_xbar);      // (same wave wrote the row: LDS operations of a wave complete in order)
      }
    }
  }
  else {
    // Several waves per query: each takes its slots through the same arithmetic as above -- key rows, time encodings,
    // scores, weighted sums -- and the parts meet in LDS twice: the scores before the softmax (every wave of the query
    // then computes the same weights), the partial weighted sums after it (the lead wave adds them in wave order).
    float* xb0 = bufT + qm * RS;
    float* xb1 = xb0 + KPS;
    float* xpart = bufT + TQ * RS;                    // [TQ waves][2 * KPS] partial x-bar rows
    float* scx = xpart + TQ * 2 * KPS;                // [TQW][2][KC] scores
    const int tb = D + d_e;
    const int hs = held_slot(lane);
    float lna[NSL], lnb[NSL], le[NSL][ME];
    unsigned long long livebits = 1ull;
    bool invalid = false;
    if (live_q) {
#pragma unroll
      for (int jl = 0; jl < NSL; ++jl) {
        const int j = qr + NQW * jl;
        lna[jl] = lnb[jl] = 0.f;
#pragma unroll
        for (int i = 0; i < ME; ++i) le[jl][i] = 0.f;
        if (j < K) {      // wave-uniform; clamped columns as in load_key_rows
          const int nr = __builtin_amdgcn_readlane(S.nrow, j), er = __builtin_amdgcn_readlane(S.erow, j);
          const int ov = __builtin_amdgcn_readlane(S.ovl, j);
          const float* np = (ov ? in.nbr_overlay : in.nbr_table) + (int64_t)nr * D;
          const float* ep = in.edge_table + (int64_t)er * d_e;
          lna[jl] = np[lane];
          lnb[jl] = np[c1 < D ? c1 : 0];
#pragma unroll
          for (int i = 0; i < ME; ++i) {
            const int c = lane + 64 * i;
            le[jl][i] = ep[c < d_e ? c : 0];
          }
        }
      }
      livebits = __ballot(lane < K && !S.masked);
      invalid = livebits == 0ull;                                        // temporal_attention.py:46
      if (invalid) livebits = 1ull;                                      // :47 un-masks slot 0
      const float qn0a = xb0[lane], qn1a = xb1[lane];
      const float qn0b = c1 < D ? xb0[c1] : 0.f, qn1b = c1 < D ? xb1[c1] : 0.f;
      float qe0[ME], qe1[ME];
#pragma unroll
      for (int i = 0; i < ME; ++i) {
        const int c = lane + 64 * i;
        qe0[i] = c < d_e ? xb0[D + c] : 0.f;
        qe1[i] = c < d_e ? xb1[D + c] : 0.f;
      }
      const float qt0a = xb0[tb + lane], qt1a = xb1[tb + lane];
      const float qt0b = c1 < D ? xb0[tb + c1] : 0.f, qt1b = c1 < D ? xb1[tb + c1] : 0.f;
      const float off = (lane & 32) ? xb1[kdim] : xb0[kdim];
      // this lane's partial dot products of the wave's slots, placed at their slots of the 2 x KC reduction
      float l0[NSL], l1[NSL];
#pragma unroll
      for (int jl = 0; jl < NSL; ++jl) {
        l0[jl] = lna[jl] * qn0a + lnb[jl] * qn0b + lta[jl] * qt0a + ltb[jl] * qt0b;
        l1[jl] = lna[jl] * qn1a + lnb[jl] * qn1b + lta[jl] * qt1a + ltb[jl] * qt1b;
#pragma unroll
        for (int i = 0; i < ME; ++i) {
          l0[jl] += le[jl][i] * qe0[i];
          l1[jl] += le[jl][i] * qe1[i];
        }
      }
      float p0[KC], p1[KC];
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        p0[j] = 0.f; p1[j] = 0.f;
#pragma unroll
        for (int jl = 0; jl < NSL; ++jl)
          if (j == qr + NQW * jl) { p0[j] = l0[jl]; p1[j] = l1[jl]; }     // (uniform)
      }
      const float sc_own = reduce_2xKC(p0, p1, lane) + off;
      if (hs >= 0 && !(lane & 1) && hs % NQW == qr) scx[(qm * 2 + (lane >> 5)) * KC + hs] = sc_own;
    }
    __syncthreads();            // (also: every wave of the query has read q~ before the lead overwrites the row below)
    float an0 = 0.f, an1 = 0.f, bn0 = 0.f, bn1 = 0.f, at0 = 0.f, at1 = 0.f, bt0 = 0.f, bt1 = 0.f;
    float ae0[ME], ae1[ME];
#pragma unroll
    for (int i = 0; i < ME; ++i) { ae0[i] = 0.f; ae1[i] = 0.f; }
    float wsum = 0.f;
    if (live_q) {
      const float sc0 = scx[(qm * 2 + (lane >> 5)) * KC + (hs < 0 ? 0 : hs)];
      // masked softmax per head = per 32-lane half; every real slot sits in two lanes of its half
      const bool lv0 = hs >= 0 && ((livebits >> hs) & 1ull);
      const float mx = half_max(lv0 ? sc0 : -INFINITY);
      const float e0 = lv0 ? expf(sc0 - mx) : 0.f;
      const float inv = invalid ? 0.f : 1.0f / (half_sum(e0) * 0.5f);
      float a_c0 = e0 * inv;
      wsum = invalid ? 0.f : 1.f;
      if (in.dropout_p > 0.f) {
        const float sc_ = 1.0f / (1.0f - in.dropout_p);
        a_c0 *= orca_dropout_mult(in.dropout_seed, in.dropout_q0 + q, lane >> 5, hs < 0 ? 0 : hs, in.dropout_p, sc_);
        wsum = half_sum(a_c0) * 0.5f;
      }
      if (out_w) {                                                       // :51-53 (head average); wave-uniform
        const int jj = lane < K ? lane : 0;
        int hl = 0;
#pragma unroll
        for (int j = 0; j < KC; ++j) hl = (jj == j) ? holder_lane(0, j) : hl;
        const float w0 = __shfl(a_c0, hl, 64), w1 = __shfl(a_c0, hl + 32, 64);
        if (lead && lane < K) out_w[q * K + lane] = invalid ? 0.f : (w0 + w1) * 0.5f;
      }
      // this wave's slots: the weight of slot j sits in lanes holder_lane(h, j) -- picked by a chain of uniform selects
#pragma unroll
      for (int jl = 0; jl < NSL; ++jl) {
        float w0 = 0.f, w1 = 0.f;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          if (j == qr + NQW * jl) {
            w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_c0), holder_lane(0, j)));
            w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_c0), holder_lane(1, j)));
          }
        }
        an0 += w0 * lna[jl]; an1 += w1 * lna[jl]; bn0 += w0 * lnb[jl]; bn1 += w1 * lnb[jl];
        at0 += w0 * lta[jl]; at1 += w1 * lta[jl]; bt0 += w0 * ltb[jl]; bt1 += w1 * ltb[jl];
#pragma unroll
        for (int i = 0; i < ME; ++i) { ae0[i] += w0 * le[jl][i]; ae1[i] += w1 * le[jl][i]; }
      }
    }
    {
      float* xp0 = xpart + wave * 2 * KPS;
      float* xp1 = xp0 + KPS;
      xp0[lane] = an0; xp1[lane] = an1;
      if (c1 < D) { xp0[c1] = bn0; xp1[c1] = bn1; }
#pragma unroll
      for (int i = 0; i < ME; ++i) {
        const int c = lane + 64 * i;
        if (c < d_e) { xp0[D + c] = ae0[i]; xp1[D + c] = ae1[i]; }
      }
      xp0[tb + lane] = at0; xp1[tb + lane] = at1;
      if (c1 < D) { xp0[tb + c1] = bt0; xp1[tb + c1] = bt1; }
    }
    __syncthreads();
    if (lead) {
      const float* xs = xpart + (qm * NQW) * 2 * KPS;
      auto total = [&](int col) {
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int r = 0; r < NQW; ++r) { v0 += xs[r * 2 * KPS + col]; v1 += xs[r * 2 * KPS + KPS + col]; }
        xb0[col] = v0; xb1[col] = v1;
      };
      total(lane);
      if (c1 < D) total(c1);
#pragma unroll
      for (int i = 0; i < ME; ++i) {
        const int c = lane + 64 * i;
        if (c < d_e) total(D + c);
      }
      total(tb + lane);
      if (c1 < D) total(tb + c1);
      // constant columns of the tail: each head's value-bias term times its weight sum, the out-projection bias once
      const float ws1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wsum), 32));
      if (lane == 0) { xb0[kdim] = wsum; xb1[kdim] = ws1; xb0[2 * KPS + D] = (invalid || !live_q) ? 0.f : 1.f; }
    }
  }
  // ---- P4: h1 = relu(F [x-bar_0 | x-bar_1 | src] + b1), F = the folded value map / out-proj / fc1 ----------
  // 7 n-tiles x KBT k-blocks cut into 16 equal-cost pieces, 4 per SIMD: waves 0..11 take half the reduction of
  // tiles 0..5, waves 12..15 a quarter of tile 6.  Partial tiles meet in LDS (the q / 0.1q buffers are dead by
  // now) and are summed in a fixed order: no atomics, the result does not depend on timing.
  float* part = bufQ;                               // [16 pieces][16 x 16]
  const bool big = wave < 12;
  const int t4 = big ? (wave >> 1) : 6;
  const int kb4 = big ? (wave & 1) * (L.KBT / 2) : (wave - 12) * (L.KBT / 4);
  const f32x4* pw4 = blob4 + L.pw_t / 4 + ((int64_t)t4 * L.KBT + kb4) * 64;
  // Small operands of the last stages, requested ahead of the tail's weight stream (behind it, i.e. in front of the barrier
  // that ends the tail, their issue waited for room in a memory pipe that the other waves' streams keep full: the stage
  // grew by what the later ones lost).  The winner flag and the id of the fused history push (P8): read where they are
  // used they were two dependent round trips at the very end of every workgroup.  Without a push the loads go to a valid
  // address and are ignored (no branch around a load: it would be waited for on the spot).
  const bool pushing = side.push_table != nullptr;
  const int64_t pq = (pushing && live_q && lead) ? q : 0;
  const uint8_t push_win = (pushing ? side.push_winner : reinterpret_cast<const uint8_t*>(blob))[pq];
  const int64_t push_id = (pushing ? side.push_ids : reinterpret_cast<const int64_t*>(blob))[pq];
  // The fc1 bias of this thread's two elements of the sum of the tail's partial tiles: read inside that loop it was an L2
  // round trip between two barriers.
  static_assert(TQ * 112 <= 2 * NW * 64, "two elements per thread");
  const int e_a = tid, e_b = tid + NW * 64;
  const int n_a = e_a % 112, n_b = e_b % 112;
  const float b1_a = blob[L.b_1 + n_a], b1_b = blob[L.b_1 + n_b];      // (the bias vector is padded to 112 zeros-filled columns)
  WaveGemm<1, 8, KBT_C / 2, true> g4;              // ONE ring for both piece lengths (see WaveGemm::run)
  g4.prefetch(pw4, big ? L.KBT / 2 : L.KBT / 4, 0, 1, 1, lane);
  __syncthreads();
  stamp();
  {
    const float* a4 = bufT + a_off * RS + a_k + kb4 * 16;
    float* dst = part + wave * 256;
    auto put = [&](int m, int n, float v) { dst[m * 16 + n] = v; };
    wstamp(4);
    if constexpr (KBT_C > 0) {
      if (big) g4.template run<KBT_C / 2>(a4, lane, put);
      else     g4.template run<KBT_C / 4>(a4, lane, put);
    } else {
      g4.run(a4, lane, put);
    }
    wstamp(5);
  }
  WaveGemm<1, 7, 7, true> g7;
  g7.prefetch(blob4 + L.pw_2 / 4, NT100, wave < NT100 ? wave : -1, NW, NT100, lane, blob + L.b_2);
  // LayerNorm affine parameters (P8), requested with the fc2 weights (at kernel entry they were one more round trip -- and
  // a spilled register -- ahead of the gathers)
  const float ln_ga = blob[L.ln_w + lane], ln_ba = blob[L.ln_b + lane];
  const float ln_gb = blob[c1 < D ? L.ln_w + c1 : L.ln_w], ln_bb = blob[c1 < D ? L.ln_b + c1 : L.ln_b];
  wstamp(6);
  __syncthreads();
  wstamp(7);
  stamp();
  auto sum_parts = [&](int e, int n, float b1) {
    const int m = e / 112, t = n >> 4, o = m * 16 + (n & 15);
    float v = t < 6 ? part[(2 * t) * 256 + o] + part[(2 * t + 1) * 256 + o]
                    : (part[12 * 256 + o] + part[13 * 256 + o]) + (part[14 * 256 + o] + part[15 * 256 + o]);
    bufM[m * HSTR + n] = n < D ? fmaxf(v + b1, 0.0f) : 0.0f;
  };
  sum_parts(e_a, n_a, b1_a);
  if (e_b < TQ * 112) sum_parts(e_b, n_b, b1_b);
  // ---- P7: y = fc2 h1 + b2 -> bufQ[:, 0:100] -----------------------------------------------------
  __syncthreads();
  stamp();
  {
    g7.run(bufM + a_off * HSTR + a_k, lane, [&](int m, int n, float v) {
      if (n < D) bufQ[m * ASTR + n] = v;
    });
  }
  __syncthreads();
  stamp();

  // ---- P8: LayerNorm(100), eps 1e-5, biased variance (:55): wave m normalises query m ------------
  if (live_q && lead) {
    const int m = qm;
    float ya = bufQ[m * ASTR + lane];
    float yb = c1 < D ? bufQ[m * ASTR + c1] : 0.f;
    float mean = wave_sum(ya + yb) * (1.0f / D);
    float da = ya - mean, db = c1 < D ? yb - mean : 0.f;
    float var = wave_sum(da * da + db * db) * (1.0f / D);
    float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float oa = da * rstd * ln_ga + ln_ba, ob = db * rstd * ln_gb + ln_bb;
    out[q * D + lane] = oa;
    if (c1 < D) out[q * D + c1] = ob;
    if (pushing && push_win) {                      // fused history push of the winner rows (common.h)
      side.push_table[push_id * D + lane] = oa;
      if (c1 < D) side.push_table[push_id * D + c1] = ob;
    }
  }
  stamp();
}

}  // namespace

static long long* g_phase_clock = nullptr;
// Profiling aid, not part of the product surface: buf = device int64 [n_workgroups * 32] (or NULL to stop; a diagnostic build writes n_workgroups * 16 * 8 more words behind them);
// later orca_tgat_forward launches stamp their phase boundaries into it.
ORCA_API int orca_debug_tgat_phase_clock(void* buf) {
  g_phase_clock = (long long*)buf;
  return ORCA_OK;
}

namespace {
__global__ void k_dropout_mask(uint64_t seed, int64_t q0, int64_t Q, int K, float p, float* __restrict__ mult) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q * ORCA_HEADS * K) return;
  const int j = (int)(i % K), h = (int)((i / K) % ORCA_HEADS);
  mult[i] = orca_dropout_mult(seed, q0 + i / (K * ORCA_HEADS), h, j, p, 1.0f / (1.0f - p));
}
}  // namespace

ORCA_API int orca_tgat_dropout_mask(uint64_t seed, int64_t q0, int64_t Q, int K, float p, float* mult, void* stream) {
  ORCA_REQUIRE(Q >= 0 && K >= 1 && K <= ORCA_MAX_K && p >= 0.f && p < 1.f && (Q == 0 || mult), "orca_tgat_dropout_mask: bad arguments");
  if (Q == 0) return ORCA_OK;
  k_dropout_mask<<<(unsigned)ceil_div64(Q * ORCA_HEADS * K, 256), 256, 0, (hipStream_t)stream>>>(seed, q0, Q, K, p, mult);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int64_t orca_tgat_saved_floats(int d_e, int64_t Q, int K) {
  if (d_e < 1 || d_e > ORCA_MAX_DE || Q < 0 || K < 1 || K > ORCA_MAX_K) return -1;
  return orca_saved(d_e, Q, K).total;
}
ORCA_API int orca_tgat_forward_saving(const float* blob, int d_e, const orca_tgat_inputs* in, int64_t Q, int K, float* out,
                                      float* out_weights, float* saved, void* stream) {
  ORCA_REQUIRE(saved, "orca_tgat_forward_saving: null buffer");
  ORCA_REQUIRE(d_e >= 1 && d_e <= ORCA_MAX_DE && K >= 1 && K <= ORCA_MAX_K && Q >= 0, "orca_tgat_forward_saving: bad sizes");
  const orca_saved_layout L = orca_saved(d_e, Q, K);
  orca_tgat_side side{};
  side.save_qt = saved + L.qt;
  side.save_a = saved + L.a;
  side.save_xbar = saved + L.xbar;
  side.save_wsum = saved + L.wsum;
  side.save_invalid = saved + L.invalid;
  return orca_tgat_forward_side(blob, d_e, in, Q, K, out, out_weights, &side, stream);
}

ORCA_API int orca_tgat_forward(const float* blob, int d_e, const orca_tgat_inputs* in, int64_t Q, int K, float* out,
                               float* out_weights, void* stream) {
  return orca_tgat_forward_side(blob, d_e, in, Q, K, out, out_weights, nullptr, stream);
}

// the PROF of k_tgat_forward a launch may take (0: none): every fact that instantiation assumes is checked here
static int tgat_profile(const orca_tgat_inputs& in, const orca_tgat_side& side, const float* out_weights) {
  static const bool off = getenv("ORCA_TGAT_PROFILES") && getenv("ORCA_TGAT_PROFILES")[0] == '0';      // (A/B switch)
  if (off) return 0;
  if (!in.nbr_idx || !in.edge_idx || !in.nbr_ids || !in.dt || side.src_rows_out || side.save_qt || side.save_a ||
      side.save_xbar || out_weights || in.dropout_p > 0.f || side.q_extra || side.store_bs < 0)
    return 0;
  const bool all = in.src_idx && in.src_flags && in.src_overlay && in.nbr_flags && in.nbr_overlay;
  const bool none = !in.src_idx && !in.src_flags && !in.nbr_flags;
  if (!all && !none) return 0;
  const int bits = (all ? 1 : 0) | (side.persist_memory ? 2 : 0) | (side.push_table ? 4 : 0) | (side.store_bs > 0 ? 8 : 0);
  return (bits == 7 || bits == 3 || bits == 8 || bits == 1 || bits == 0) ? 16 | bits : 0;      // the instantiated ones
}
int orca_tgat_forward_side(const float* blob, int d_e, const orca_tgat_inputs* in, int64_t Q, int K, float* out,
                           float* out_weights, const orca_tgat_side* side_, void* stream) {
  const orca_tgat_side side = side_ ? *side_ : orca_tgat_side{};
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
  ORCA_REQUIRE(in->nbr_idx || Q * K < (1LL << 31), "orca_tgat_forward: dense mode limited to Q*K < 2^31");
  Blob L = blob_layout(d_e);
  size_t lds_bytes = (size_t)(2 * TQ * ASTR + TQ * HSTR + TQ * (L.KBT * 16 + 4)) * sizeof(float);
  static bool attr_set[64] = {};
  if (orca_first_use_on_device(attr_set)) {
    const int lim = 160 * 1024 - 256;
    const void* fns[] = {(const void*)k_tgat_forward<10, 172, false>, (const void*)k_tgat_forward<10, 172, true>,
                         (const void*)k_tgat_forward<10, 0, false>,   (const void*)k_tgat_forward<10, 0, true>,
                         (const void*)k_tgat_forward<0, 0, false>,    (const void*)k_tgat_forward<0, 0, true>,
                         (const void*)k_tgat_forward<10, 1, false>,   (const void*)k_tgat_forward<10, 1, true>,
                         (const void*)k_tgat_forward<20, 1, false>,   (const void*)k_tgat_forward<20, 1, true>,
                         (const void*)k_tgat_forward<10, 172, true, 4>, (const void*)k_tgat_forward<10, 1, true, 4>,
#define ORCA_PROFS_OF(KT_, DE_, NQW_)                                                                            \
  (const void*)k_tgat_forward<KT_, DE_, true, NQW_, 16 | 7>, (const void*)k_tgat_forward<KT_, DE_, true, NQW_, 16 | 3>, \
  (const void*)k_tgat_forward<KT_, DE_, true, NQW_, 16 | 8>, (const void*)k_tgat_forward<KT_, DE_, true, NQW_, 16 | 1>, \
  (const void*)k_tgat_forward<KT_, DE_, true, NQW_, 16 | 0>
                         ORCA_PROFS_OF(10, 172, 1), ORCA_PROFS_OF(10, 1, 1), ORCA_PROFS_OF(20, 1, 1),
                         ORCA_PROFS_OF(10, 172, 4), ORCA_PROFS_OF(10, 1, 4)};
#undef ORCA_PROFS_OF
    for (const void* f : fns) ORCA_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
  }
  ORCA_REQUIRE(lds_bytes <= 160 * 1024 - 256, "orca_tgat_forward: LDS budget exceeded (%zu B)", lds_bytes);
  ORCA_REQUIRE(!in->q_folded || in->dt, "orca_tgat_forward: q_folded needs gather-mode time inputs (dt)");
  ORCA_REQUIRE(in->dropout_p >= 0.f && in->dropout_p < 1.f, "orca_tgat_forward: dropout_p=%g outside [0,1)", (double)in->dropout_p);
  hipStream_t st = (hipStream_t)stream;
  // (q-folded launches of the compile-time shapes take the instantiation of their profile, see k_tgat_forward)
  const int prof = (in->q_folded & 1) ? tgat_profile(*in, side, out_weights) : 0;
#define ORCA_TGAT_LAUNCH_P1(KT_, DE_, NQW_, P_, GRID_, LDS_)                                                      \
  ORCA_LAUNCH_TIMED((k_tgat_forward<KT_, DE_, true, NQW_, P_>), GRID_, NW * 64, LDS_, st, blob, d_e, *in, Q, K, out, out_weights, \
                    side, g_phase_clock)
#define ORCA_TGAT_LAUNCH_PROF(KT_, DE_, NQW_, GRID_, LDS_)                                                        \
  do {                                                                                                          \
    switch (prof) {                                                                                             \
      case 16 | 7: ORCA_TGAT_LAUNCH_P1(KT_, DE_, NQW_, 16 | 7, GRID_, LDS_); break;                             \
      case 16 | 3: ORCA_TGAT_LAUNCH_P1(KT_, DE_, NQW_, 16 | 3, GRID_, LDS_); break;                             \
      case 16 | 8: ORCA_TGAT_LAUNCH_P1(KT_, DE_, NQW_, 16 | 8, GRID_, LDS_); break;                             \
      case 16 | 1: ORCA_TGAT_LAUNCH_P1(KT_, DE_, NQW_, 16 | 1, GRID_, LDS_); break;                             \
      case 16 | 0: ORCA_TGAT_LAUNCH_P1(KT_, DE_, NQW_, 16 | 0, GRID_, LDS_); break;                             \
      default: ORCA_TGAT_LAUNCH_P1(KT_, DE_, NQW_, 0, GRID_, LDS_); break;                                      \
    }                                                                                                           \
  } while (0)
  if (ORCA_TGAT_WAVES(in->q_folded) == 4 && K == 10 && (d_e == 172 || d_e == 1) && (in->q_folded & 1) && !side.save_qt) {
    const size_t lds4 = lds_bytes + (size_t)(TQ * 2 * L.KP + TQ * 2 * KC) * sizeof(float);     // partial x-bar rows, scores
    ORCA_REQUIRE(lds4 <= 160 * 1024 - 256, "orca_tgat_forward: LDS budget exceeded (%zu B)", lds4);
    const unsigned grid4 = (unsigned)ceil_div64(Q, TQ / 4);
    if (d_e == 172) ORCA_TGAT_LAUNCH_PROF(10, 172, 4, grid4, lds4);
    else            ORCA_TGAT_LAUNCH_PROF(10, 1, 4, grid4, lds4);
    ORCA_LAUNCH_CHECK();
    return ORCA_OK;
  }
  const unsigned grid = (unsigned)ceil_div64(Q, TQ);
  if ((in->q_folded & 1) && prof != 0 && ((K == 10 && (d_e == 172 || d_e == 1)) || (K == 20 && d_e == 1))) {
    if (K == 10 && d_e == 172) ORCA_TGAT_LAUNCH_PROF(10, 172, 1, grid, lds_bytes);
    else if (K == 10)          ORCA_TGAT_LAUNCH_PROF(10, 1, 1, grid, lds_bytes);
    else                       ORCA_TGAT_LAUNCH_PROF(20, 1, 1, grid, lds_bytes);
    ORCA_LAUNCH_CHECK();
    return ORCA_OK;
  }
#undef ORCA_TGAT_LAUNCH_PROF
#undef ORCA_TGAT_LAUNCH_P1
#define ORCA_TGAT_LAUNCH(KT_, DE_)                                                                              \
  do {                                                                                                          \
    if (in->q_folded & 1)                                                                                       \
      ORCA_LAUNCH_TIMED((k_tgat_forward<KT_, DE_, true>), grid, NW * 64, lds_bytes, st, blob, d_e, *in, Q, K, out, \
                        out_weights, side, g_phase_clock);                                                      \
    else                                                                                                        \
      ORCA_LAUNCH_TIMED((k_tgat_forward<KT_, DE_, false>), grid, NW * 64, lds_bytes, st, blob, d_e, *in, Q, K, out, \
                        out_weights, side, g_phase_clock);                                                      \
  } while (0)
  // compile-time shapes: the reference's default (n_degree 10, 172-d Wikipedia / Reddit features), and the zero-feature
  // datasets (train.py:139-142: d_e = 1) at n_degree 10 (BASELINE config 3) and 20 (config 5)
  if (K == 10 && d_e == 172)
    ORCA_TGAT_LAUNCH(10, 172);
  else if (K == 10 && d_e == 1)
    ORCA_TGAT_LAUNCH(10, 1);
  else if (K == 20 && d_e == 1)
    ORCA_TGAT_LAUNCH(20, 1);
  else if (K == 10)
    ORCA_TGAT_LAUNCH(10, 0);
  else
    ORCA_TGAT_LAUNCH(0, 0);
#undef ORCA_TGAT_LAUNCH
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}
