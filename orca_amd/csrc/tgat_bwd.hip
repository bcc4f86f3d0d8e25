// Backward pass of the TGAT attention layer (autograd of model/temporal_attention.py:24-56 as train.py:236-237
// runs it): gradients of the raw state_dict tensors and of the layer's inputs.
//
// First cut, built for correctness before speed: the forward intermediates are recomputed from the raw
// (unpacked, unfolded) parameters, every dense contraction -- recompute, data gradients and the weight
// gradients, which reduce over the queries -- goes through ONE generic strided fp32 MFMA GEMM
// (v_mfma_f32_16x16x4_f32, 64x64 tiles staged in LDS, deterministic split-K), and the per-query parts (key
// gathers, scores, softmax and their derivatives, LayerNorm) are wave-per-query kernels.  The algebra of the
// forward kernel is kept where it removes work: scores and the weighted key sum act on the raw key rows
// (q~_h = Wk_h^T 0.1 q_h, x-bar_h = sum_j a_hj x_j), so the K x k_dim key matrix is never projected.
// Intermediates live in a caller-provided workspace (orca_tgat_backward_workspace_floats).
#include "common.h"
#include <algorithm>
#include <vector>

namespace {

constexpr int D = ORCA_D;
constexpr int QD = 2 * D;
constexpr int MAXC = (QD + ORCA_MAX_DE + 63) / 64;   // key columns per lane

// ---- generic strided GEMM ---------------------------------------------------------------------------------
// C(m,n) = alpha * (sum_k A(m,k) B(k,n) + bias[n]) [+ C(m,n)] [relu];  X(i,j) = X[i*s0 + j*s1].
// With part != NULL block z writes its raw partial sums of k in [z*kchunk, (z+1)*kchunk) to part[z][m][n].
struct GemmArgs {
  int M, N, K;             // N counts the virtual ones column when ones_out is set
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t scm, scn;
  const float* bias;
  float alpha;
  int accumulate, relu, kchunk;
  float* part;
  const float* gate;
  const float* rowzero;
  int rowzero_n;
  float* ones_out;
  int gx, gy, gz;          // tiles along n, m and k
};
struct GemmGroup {
  int n;
  int first[ORCA_GEMM_GROUP_MAX + 1];   // first block of each problem (GEMM launch) / first 256-element block (sums)
  GemmArgs g[ORCA_GEMM_GROUP_MAX];
};

__device__ __forceinline__ void gemm_store(const GemmArgs& g, int m, int n, float acc) {
  const int Nr = g.ones_out ? g.N - 1 : g.N;      // real columns
  if (n == Nr) {                                  // the ones column: a bias gradient
    const float v = g.alpha * acc;
    g.ones_out[m] = g.accumulate ? g.ones_out[m] + v : v;
    return;
  }
  float v = g.alpha * (acc + (g.bias ? g.bias[n] : 0.f));
  float* c = g.C + (int64_t)m * g.scm + (int64_t)n * g.scn;
  if (g.accumulate) v += *c;
  if (g.relu) v = fmaxf(v, 0.f);
  if (g.gate && !(g.gate[(int64_t)m * g.scm + (int64_t)n * g.scn] > 0.f)) v = 0.f;
  if (g.rowzero && n < g.rowzero_n && g.rowzero[m] != 0.f) v = 0.f;
  *c = v;
}

// TM = rows of the block tile (64, or 32 when the grid would otherwise leave CUs idle); the tile is TM x 64 and the
// reduction advances GEMM_BK = 64 at a time, the next slab's global loads in flight while the current one is
// multiplied.  These problems are small (a few hundred blocks of a few slabs): a block's time is its chain of
// load -> LDS -> barrier round trips, so the slab is deep (16-deep slabs: 15 us per launch, 64-deep: see DESIGN).
constexpr int GEMM_BK = 32;
template <int TM>
__global__ __launch_bounds__(256) void k_sgemm(GemmGroup grp) {
  constexpr int WR = TM >= 32 ? 2 : 1;         // the four waves as WR x (4 / WR) over the TM x 64 tile
  constexpr int IM = TM / (16 * WR);           // 16-row MFMA tiles per wave along m
  constexpr int JN = 64 / (16 * (4 / WR));     // 16-column MFMA tiles per wave along n
  constexpr int EA = TM * GEMM_BK / 256;       // A elements per thread and slab
  constexpr int EB = 64 * GEMM_BK / 256;       // B elements per thread and slab
  __shared__ float As[GEMM_BK][TM + 4], Bs[GEMM_BK][68];
  int p = 0;
#pragma unroll
  for (int i = 1; i < ORCA_GEMM_GROUP_MAX; ++i) p += (i < grp.n && (int)blockIdx.x >= grp.first[i]) ? 1 : 0;
  p = __builtin_amdgcn_readfirstlane(p);
  const GemmArgs g = grp.g[p];
  const int local = (int)blockIdx.x - grp.first[p];
  const int bx = local % g.gx, by = (local / g.gx) % g.gy, bz = local / (g.gx * g.gy);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = by * TM, n0 = bx * 64;
  const int kbeg = bz * g.kchunk;
  const int kend = kbeg + g.kchunk < g.K ? kbeg + g.kchunk : g.K;
  const int Nr = g.ones_out ? g.N - 1 : g.N;
  const int wm = WR == 2 ? (wave >> 1) * (TM / 2) : 0, wn = WR == 2 ? (wave & 1) * 32 : wave * 16;
  f32x4 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool a_kfast = g.sak == 1, b_nfast = g.sbn == 1;
  float ra[EA], rb[EB];
  auto a_index = [&](int e, int& m, int& k) {
    const int idx = t + 256 * e;
    if (a_kfast) { m = idx / GEMM_BK; k = idx % GEMM_BK; } else { m = idx % TM; k = idx / TM; }
  };
  auto b_index = [&](int e, int& n, int& kk) {
    const int idx = t + 256 * e;
    if (b_nfast) { n = idx & 63; kk = idx >> 6; } else { kk = idx % GEMM_BK; n = idx / GEMM_BK; }
  };
  // running pointers and row / column masks: the 64-bit strided address of every element is computed once, not per slab
  const float* pa[EA];
  const float* pb[EB];
  int ka[EA], kb[EB];        // k offset inside the slab; -1: the row / column is out of range (loads nothing)
  int onesb = 0;             // bit e: element e of B lies in the virtual ones column
#pragma unroll
  for (int e = 0; e < EA; ++e) {
    int m, k;
    a_index(e, m, k);
    ka[e] = m0 + m < g.M ? k : -1;
    pa[e] = g.A + (int64_t)(m0 + m < g.M ? m0 + m : 0) * g.sam + (int64_t)(kbeg + k) * g.sak;
  }
#pragma unroll
  for (int e = 0; e < EB; ++e) {
    int n, kk;
    b_index(e, n, kk);
    kb[e] = n0 + n < g.N ? kk : -1;
    if (n0 + n >= Nr && n0 + n < g.N) onesb |= 1 << e;
    pb[e] = g.B + (int64_t)(kbeg + kk) * g.sbk + (int64_t)(n0 + n < Nr ? n0 + n : 0) * g.sbn;
  }
  const int64_t step_a = (int64_t)GEMM_BK * g.sak, step_b = (int64_t)GEMM_BK * g.sbk;
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      ra[e] = (ka[e] >= 0 && k0 + ka[e] < kend) ? *pa[e] : 0.f;
      pa[e] += step_a;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const bool in = kb[e] >= 0 && k0 + kb[e] < kend;
      rb[e] = in ? (((onesb >> e) & 1) ? 1.f : *pb[e]) : 0.f;
      pb[e] += step_b;
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      int m, k;
      a_index(e, m, k);
      As[k][m] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      int n, kk;
      b_index(e, n, kk);
      Bs[kk][n] = rb[e];
    }
    __syncthreads();
    if (k0 + GEMM_BK < kend) fetch(k0 + GEMM_BK);
    const int ksteps = (kend - k0 < GEMM_BK ? kend - k0 + 3 : GEMM_BK) / 4;     // rows past kend hold zeros: skip them
    for (int kk = 0; kk < ksteps; ++kk) {
      const int kr = kk * 4 + (lane >> 4);
      float a[IM], b[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) a[i] = As[kr][wm + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < JN; ++j) b[j] = Bs[kr][wn + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + 4 * (lane >> 4) + r, n = n0 + wn + j * 16 + (lane & 15);
        if (m >= g.M || n >= g.N) continue;
        if (g.part) g.part[((int64_t)bz * g.M + m) * g.N + n] = acc[i][j][r];
        else gemm_store(g, m, n, acc[i][j][r]);
      }
}

// ---- the same GEMM without LDS and without barriers, for activations times weights ----------------------------------
// The backward's data GEMMs are C [M x N] = A [M x K] B with M = the queries (thousands), N and K a few hundred, A
// row-major.  Here a wave owns a 16-row x 32-column tile outright: lane (r = lane & 15, kq = lane >> 4) loads the
// float4 A[m0 + r][k0 + 4 kq ..] -- the four k-steps of a 16-deep block -- and, for each of its two column tiles, the
// matching four values of B (one float4 when B is k-contiguous, i.e. a weight matrix used transposed; four dwords
// when it is n-contiguous).  MFMA step e then consumes element e of both: the k index 4 kq + e is a permutation of
// the block's 16, the same on both operands, which is all a sum over k needs.  The next block's loads are issued
// before the current block's MFMAs; nothing is shared between waves except through L1 / L2 (the four waves of a
// workgroup read the same 16 rows of A).  Needs 16-byte aligned float4 rows (K, row strides and bases multiples of 4
// floats): orca_sgemm_group checks and otherwise takes k_sgemm.  About 10 % faster than k_sgemm on these shapes
// (8-19 us a launch).
template <bool B_KFAST>
__global__ __launch_bounds__(256) void k_sgemm_rows(GemmGroup grp) {
  int p = 0;
#pragma unroll
  for (int i = 1; i < ORCA_GEMM_GROUP_MAX; ++i) p += (i < grp.n && (int)blockIdx.x >= grp.first[i]) ? 1 : 0;
  p = __builtin_amdgcn_readfirstlane(p);
  const GemmArgs g = grp.g[p];
  const int local = (int)blockIdx.x - grp.first[p];
  const int bx = local % g.gx, by = local / g.gx;        // gx column groups of 128, gy row tiles of 16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int m0 = by * 16, n0 = bx * 128 + wave * 32;
  if (n0 >= g.N) return;
  const int mr = m0 + r < g.M ? m0 + r : g.M - 1;        // clamped, never skipped (the store is masked)
  const float* arow = g.A + (int64_t)mr * g.sam + 4 * kq;
  const float* bcol[2];
  bool nok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 16 * j + r;
    nok[j] = n < g.N;
    const int nc = nok[j] ? n : 0;
    bcol[j] = B_KFAST ? g.B + (int64_t)nc * g.sbn + 4 * kq : g.B + (int64_t)(4 * kq) * g.sbk + nc;
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // Loads are unconditional from clamped addresses and masked by a multiplication (the data is finite): behind a
  // select the compiler puts the load under a branch, and a load behind a branch is waited for where it is issued.
  // The next block's operands are requested before the current block's MFMAs and pinned there (sched_barrier).
  const float nf0 = nok[0] ? 1.f : 0.f, nf1 = nok[1] ? 1.f : 0.f;
  struct Ops { f32x4 a, b0, b1; };
  auto load = [&](int k0) {
    Ops o;
    const int kc = k0 + 4 * kq < g.K ? k0 : 0;
    o.a = *reinterpret_cast<const f32x4*>(arow + kc);
    if (B_KFAST) {
      o.b0 = *reinterpret_cast<const f32x4*>(bcol[0] + kc);
      o.b1 = *reinterpret_cast<const f32x4*>(bcol[1] + kc);
    } else {
      const float* p0 = bcol[0] + (int64_t)kc * g.sbk;
      const float* p1 = bcol[1] + (int64_t)kc * g.sbk;
      o.b0 = f32x4{p0[0], p0[g.sbk], p0[2 * g.sbk], p0[3 * g.sbk]};
      o.b1 = f32x4{p1[0], p1[g.sbk], p1[2 * g.sbk], p1[3 * g.sbk]};
    }
    return o;
  };
  Ops raw = load(0);
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    const float kf = k0 + 4 * kq < g.K ? 1.f : 0.f;
    const f32x4 a = raw.a * kf, b0 = raw.b0 * nf0, b1 = raw.b1 * nf1;      // (the wait for this block's operands)
    raw = load(k0 + 16);                                                      // (past K: the clamped block, unused)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 4 * kq + i, n = n0 + 16 * j + r;
      if (m < g.M && n < g.N) gemm_store(g, m, n, acc[j][i]);
    }
}

// ---- weight gradients: C [M x N] = A^T B, a reduction over the queries -------------------------------------------
// A [K x M] and B [K x N] are both row-major with the reduction index k slowest (activations and their gradients, one
// row per query).  No LDS staging: lane (c = lane & 15, kq = lane >> 4) loads V consecutive floats of row k0 + kq of A
// and four of B -- V m-tiles and four n-tiles of 16 whose rows / columns are interleaved (tile t holds the rows
// m0 + V i + t) -- so one pair of fully coalesced vector loads feeds 4 V MFMAs of a (16 V) x 64 wave tile, and nothing
// is shared between waves.  M is cut into 64-row pieces (V = 4) and a remainder of 32 (V = 2) and / or 16 (V = 1):
// 100 rows are 64 + 32 + 16, not two thirds-empty 64 x 64 tiles.  The four waves of a workgroup take a quarter of the
// block's k range each and meet in LDS in a fixed order (deterministic); the block-level split-K partials then go
// through k_splitk_reduce as before.  The virtual ones column (bias gradients) is a select on B.
constexpr int WG_SPLITS = 8;
__device__ __forceinline__ int wgrad_m_items(int M) {
  const int r = M & 63;
  return (M >> 6) + (r == 0 ? 0 : (r > 32 && r <= 48) ? 2 : 1);
}
inline int wgrad_m_items_host(int M) {
  const int r = M & 63;
  return (M >> 6) + (r == 0 ? 0 : (r > 32 && r <= 48) ? 2 : 1);
}
template <int V>
__device__ __forceinline__ void wgrad_tile(const GemmArgs& g, int m0, int n0, int bz, f32x4* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int Nr = g.ones_out ? g.N - 1 : g.N;
  const int kbeg = bz * g.kchunk;
  const int kend = kbeg + g.kchunk < g.K ? kbeg + g.kchunk : g.K;
  const int sub = (g.kchunk / 4 + 3) / 4 * 4;            // this wave's share of the block's k range
  const int kb = kbeg + wave * sub;
  const int ke = kb + sub < kend ? kb + sub : kend;
  const int ma = m0 + V * c, nb = n0 + 4 * c;
  const bool mok = ma < g.M, nok = nb < Nr, one = nb == Nr;
  const float* ap = g.A + (mok ? ma : 0);
  const float* bp = g.B + (nok ? nb : 0);
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[V][4];
#pragma unroll
  for (int t = 0; t < V; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = zero4;
  // Out-of-range lanes load from clamped addresses and are masked ARITHMETICALLY (times 0 or 1: the data is finite): a
  // select lets the compiler sink the load under a branch, and behind a branch every load is waited for on the spot --
  // the loop then alternates between waiting and multiplying instead of overlapping them.  Rows m >= M need no mask
  // at all: their accumulators are never stored.
  const float nokf = nok ? 1.f : 0.f, onef = one ? 1.f : 0.f;
  struct Ops { float a[V]; f32x4 b; };
  auto load = [&](int k0) {          // the raw operands of step k0 (requested, not yet looked at)
    Ops o;
    const int k = k0 + kq;
    const int kc = k < ke ? k : kbeg;                    // clamped, never skipped
    const float* pa = ap + (int64_t)kc * g.sak;
    if constexpr (V == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(pa); o.a[0] = v[0]; o.a[1] = v[1]; o.a[2] = v[2]; o.a[3] = v[3]; }
    else if constexpr (V == 2) { const float2 v = *reinterpret_cast<const float2*>(pa); o.a[0] = v.x; o.a[1] = v.y; }
    else o.a[0] = *pa;
    o.b = *reinterpret_cast<const f32x4*>(bp + (int64_t)kc * g.sbk);
    return o;
  };
  auto mask = [&](Ops& o, int k0) {
    const float kokf = k0 + kq < ke ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < V; ++t) o.a[t] *= kokf;
    o.b = o.b * nokf;
    o.b[0] += onef;
  };
  auto mma = [&](const Ops& o) {
#pragma unroll
    for (int t = 0; t < V; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[t], o.b[u], acc[t][u], 0, 0, 0);
  };
  if (kb < ke) {
    Ops r0 = load(kb), r1 = load(kb + 4);
    for (int k0 = kb; k0 < ke; k0 += 8) {
      Ops c0 = r0, c1 = r1;
      mask(c0, k0);                      // (the wait for this step's operands, requested one iteration ago)
      mask(c1, k0 + 4);
      r0 = load(k0 + 8);                 // the next iteration's: in flight under the 8 V MFMAs below
      r1 = load(k0 + 12);
      __builtin_amdgcn_sched_barrier(0);
      mma(c0);
      mma(c1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // waves 2, 3 -> LDS; waves 0, 1 add them; wave 1 -> LDS; wave 0 adds it and stores the block's partial sums
  auto put = [&](f32x4* dst) {
#pragma unroll
    for (int t = 0; t < V; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[(t * 4 + u) * 64 + lane] = acc[t][u];
  };
  auto add = [&](const f32x4* src) {
#pragma unroll
    for (int t = 0; t < V; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] += src[(t * 4 + u) * 64 + lane];
  };
  if (wave >= 2) put(red + (wave - 2) * 1024);
  __syncthreads();
  if (wave < 2) add(red + wave * 1024);
  __syncthreads();
  if (wave == 1) put(red);
  __syncthreads();
  if (wave == 0) {
    add(red);
#pragma unroll
    for (int t = 0; t < V; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + V * (4 * kq + r) + t, n = n0 + 4 * c + u;
          if (m < g.M && n < g.N) g.part[((int64_t)bz * g.M + m) * g.N + n] = acc[t][u][r];
        }
  }
}
// The tiles of a launch differ in cost (V = 4, 2, 1) and the whole launch is resident at once -- about one round of
// workgroups -- so the ORDER of the blocks decides how evenly the CUs are loaded: the host lists the (problem, m piece,
// n block) tiles heaviest first, and every tile's k pieces are consecutive blocks.
constexpr int WG_ITEMS_MAX = 448;
struct WgItems {
  int n, gz;
  uint16_t code[WG_ITEMS_MAX];      // problem << 8 | m piece << 4 | n block
};
__global__ __launch_bounds__(256, 4) void k_wgrad(GemmGroup grp, WgItems items) {
  __shared__ f32x4 red[2 * 1024];
  const int item = (int)blockIdx.x / items.gz, bz = (int)blockIdx.x % items.gz;
  const int code = items.code[item];
  const int p = code >> 8, mi = (code >> 4) & 15, bx = code & 15;
  const GemmArgs g = grp.g[p];
  // the mi-th piece of M: full 64-row pieces first, then the remainder as 32 + 16, 32 or 16 rows
  const int full = g.M >> 6, r = g.M & 63;
  int m0 = mi * 64, V = 4;
  if (mi >= full && r <= 48) {
    if (r > 32) { V = mi == full ? 2 : 1; m0 = full * 64 + (mi == full ? 0 : 32); }
    else V = r > 16 ? 2 : 1;
  }
  if (V == 4) wgrad_tile<4>(g, m0, bx * 64, bz, red);
  else if (V == 2) wgrad_tile<2>(g, m0, bx * 64, bz, red);
  else wgrad_tile<1>(g, m0, bx * 64, bz, red);
}

// the split-K sums of every problem of a group that has them
__global__ __launch_bounds__(256) void k_splitk_reduce(GemmGroup grp) {
  int p = 0;
#pragma unroll
  for (int i = 1; i < ORCA_GEMM_GROUP_MAX; ++i) p += (i < grp.n && (int)blockIdx.x >= grp.first[i]) ? 1 : 0;
  p = __builtin_amdgcn_readfirstlane(p);
  const GemmArgs g = grp.g[p];
  const int64_t i = (int64_t)((int)blockIdx.x - grp.first[p]) * 256 + threadIdx.x;
  const int64_t MN = (int64_t)g.M * g.N;
  if (i >= MN) return;
  // at most ORCA_GEMM_SPLITS partials, clamped and masked instead of skipped: no control flow, so every load is in
  // flight before the first use (a loop pays one memory round trip per iteration); fixed order: deterministic
  float v = 0.f;
#pragma unroll
  for (int z = 0; z < ORCA_GEMM_SPLITS; ++z) {
    const float x = g.part[(int64_t)(z < g.gz ? z : 0) * MN + i];
    v += z < g.gz ? x : 0.f;
  }
  gemm_store(g, (int)(i / g.N), (int)(i % g.N), v);
}

// out[n] (+)= alpha * sum_q X[q*ld + n], n < ncols, in two branch-free passes (a loop over rows pays a full memory
// round trip per iteration): block (x, y) of k_colsum sums the rows q = y + 64 * (r + 16 u) of 64 columns, 16 row
// lanes r with COLSUM_U loads each in flight, into part[y][n]; k_colsum_finish adds the 64 slices in a fixed order.
constexpr int COLSUM_SLICES = 64, COLSUM_U = 8;     // one pass covers 64 * 16 * 8 = 8192 rows
constexpr int COLSUM_JOBS = 4, COLSUM_MAXC = 320;   // up to four sums of at most 320 columns in one launch pair
struct ColsumJobs {
  const float* X[COLSUM_JOBS];
  int64_t ld[COLSUM_JOBS];
  int ncols[COLSUM_JOBS];
  float* out[COLSUM_JOBS];
};
__global__ __launch_bounds__(1024) void k_colsum(ColsumJobs jobs, int64_t Q, float* __restrict__ part_all) {
  __shared__ float red[16][64];
  const int job = blockIdx.z;
  const float* __restrict__ X = jobs.X[job];
  const int64_t ld = jobs.ld[job];
  const int ncols = jobs.ncols[job];
  float* __restrict__ part = part_all + (int64_t)job * COLSUM_SLICES * COLSUM_MAXC;
  if ((int)blockIdx.x * 64 >= ncols) return;
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, r = threadIdx.x >> 6;
  const int cc = c < ncols ? c : 0;
  float v = 0.f;
  for (int64_t base = 0; base < Q; base += (int64_t)COLSUM_SLICES * 16 * COLSUM_U) {
    float x[COLSUM_U];
#pragma unroll
    for (int u = 0; u < COLSUM_U; ++u) {
      const int64_t q = base + blockIdx.y + (int64_t)COLSUM_SLICES * (r + 16 * u);
      x[u] = X[(q < Q ? q : 0) * ld + cc];
      if (q >= Q) x[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < COLSUM_U; ++u) v += x[u];
  }
  red[r][cl] = v;
  __syncthreads();
  if (r == 0 && c < ncols) {
    v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][cl];
    part[(int64_t)blockIdx.y * ncols + c] = v;
  }
}
// one wave per column: lane z holds slice z, summed by a fixed butterfly (deterministic)
__global__ __launch_bounds__(256) void k_colsum_finish(ColsumJobs jobs, const float* __restrict__ part_all, float alpha,
                                                       int accumulate) {
  static_assert(COLSUM_SLICES == 64, "one slice per lane");
  const int job = blockIdx.y;
  const int ncols = jobs.ncols[job];
  float* out = jobs.out[job];
  const float* __restrict__ part = part_all + (int64_t)job * COLSUM_SLICES * COLSUM_MAXC;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), z = threadIdx.x & 63;
  if (c >= ncols) return;
  const float v = wave_sum(part[(int64_t)z * ncols + c]);
  if (z == 0) out[c] = accumulate ? out[c] + alpha * v : alpha * v;
}

int group_splits(const OrcaGemm& p) {
  if (p.splits <= 1 || p.K <= 64) return 1;
  const int sp = p.splits > ORCA_GEMM_SPLITS ? ORCA_GEMM_SPLITS : p.splits;
  int kc = (p.K + sp - 1) / sp;
  kc = (kc + 15) / 16 * 16;
  return (p.K + kc - 1) / kc;
}

}  // namespace

int64_t orca_sgemm_group_part_floats(const OrcaGemm* g, int n) {
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    const int gz = group_splits(g[i]);
    if (gz > 1) total += ((int64_t)gz * g[i].M * (g[i].N + (g[i].ones_out ? 1 : 0)) + 63) / 64 * 64;
  }
  return total;
}

int orca_sgemm_group(hipStream_t s, const OrcaGemm* in, int n, float* part) {
  if (n <= 0) return ORCA_OK;
  if (n > ORCA_GEMM_GROUP_MAX) {
    orca_set_error("orca_sgemm_group: %d problems in one group (at most %d)", n, ORCA_GEMM_GROUP_MAX);
    return ORCA_EINVAL;
  }
  GemmGroup grp{};
  int64_t tiles64 = 0;
  int np = 0;
  bool any_split = false;
  int64_t po = 0;
  for (int i = 0; i < n; ++i) {
    const OrcaGemm& p = in[i];
    if (p.M <= 0 || (p.N <= 0 && !p.ones_out)) continue;
    GemmArgs& g = grp.g[np++];
    g = GemmArgs{p.M, p.N + (p.ones_out ? 1 : 0), p.K, p.A, p.sam, p.sak, p.B, p.sbk, p.sbn, p.C, p.scm, p.scn, p.bias, p.alpha,
                 p.accumulate, p.relu, p.K < 16 ? 16 : p.K, nullptr, p.gate, p.rowzero, p.rowzero_n, p.ones_out, 0, 0, 1};
    g.gz = group_splits(p);
    if (g.gz > 1) {
      if (!part || p.bias || p.relu || p.gate || p.rowzero) {
        orca_set_error("orca_sgemm_group: a split-K problem needs a partial-sum buffer and takes no bias / relu / gate / rowzero");
        return ORCA_EINVAL;
      }
      const int sp = p.splits > ORCA_GEMM_SPLITS ? ORCA_GEMM_SPLITS : p.splits;
      int kc = (p.K + sp - 1) / sp;
      g.kchunk = (kc + 15) / 16 * 16;
      g.part = part + po;
      po += ((int64_t)g.gz * g.M * g.N + 63) / 64 * 64;
      any_split = true;
    }
    g.gx = (g.N + 63) / 64;
    tiles64 += (int64_t)g.gx * ((g.M + 63) / 64) * g.gz;
  }
  if (np == 0) return ORCA_OK;
  grp.n = np;
  {   // activations x weights (see k_sgemm_rows): every problem row-major in A, aligned for float4 rows, no split-K
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool rows = !any_split, kfast = grp.g[0].sbk == 1 && grp.g[0].sbn != 1;
    for (int i = 0; i < np && rows; ++i) {
      const GemmArgs& g = grp.g[i];
      const bool kf = g.sbk == 1 && g.sbn != 1;
      rows = g.sak == 1 && g.K % 4 == 0 && g.sam % 4 == 0 && al16(g.A) && !g.ones_out && g.M >= 256 && kf == kfast &&
             (kf ? (g.sbn % 4 == 0 && al16(g.B)) : g.sbn == 1);
    }
    static const bool off = getenv("ORCA_GEMM_ROWS") && atoi(getenv("ORCA_GEMM_ROWS")) == 0;
    if (rows && !off) {
      int first = 0;
      for (int i = 0; i < np; ++i) {
        grp.first[i] = first;
        grp.g[i].gx = (grp.g[i].N + 127) / 128;
        grp.g[i].gy = (grp.g[i].M + 15) / 16;
        first += grp.g[i].gx * grp.g[i].gy;
      }
      grp.first[np] = first;
      if (kfast) k_sgemm_rows<true><<<(unsigned)first, 256, 0, s>>>(grp);
      else k_sgemm_rows<false><<<(unsigned)first, 256, 0, s>>>(grp);
      return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
    }
  }
  {   // weight gradients (see k_wgrad): every problem split, reduction index slowest in both operands, float4-able
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool wg = any_split;
    for (int i = 0; i < np && wg; ++i) {
      const GemmArgs& g = grp.g[i];
      const int Nr = g.ones_out ? g.N - 1 : g.N;
      wg = g.gz > 1 && g.sam == 1 && g.sbn == 1 && g.M % 4 == 0 && Nr % 4 == 0 && g.sak % 4 == 0 && g.sbk % 4 == 0 &&
           al16(g.A) && al16(g.B) && g.scn == 1 && !g.bias && !g.relu && !g.gate && !g.rowzero;
    }
    static const bool off = getenv("ORCA_GEMM_WGRAD") && atoi(getenv("ORCA_GEMM_WGRAD")) == 0;
    if (wg && !off) {
      GemmGroup w = grp;
      int64_t po2 = 0;
      struct It { int v, code; };
      std::vector<It> list;
      bool fits = true;
      // k pieces per tile: enough for about one round of workgroups (four per CU), at least WG_SPLITS and never more than
      // the caller sized `part` for -- the GRU's two gradients are 60 tiles over 2 k rows: with 8 pieces half the chip idles
      int n_tiles = 0, cap = ORCA_GEMM_SPLITS;
      for (int i = 0; i < np; ++i) {
        const GemmArgs& g = grp.g[i];
        n_tiles += ((g.N + 63) / 64) * wgrad_m_items_host(g.M);
        cap = g.gz < cap ? g.gz : cap;
      }
      int splits = n_tiles > 0 ? (1024 + n_tiles - 1) / n_tiles : WG_SPLITS;
      splits = splits < WG_SPLITS ? WG_SPLITS : splits;
      static const int forced = getenv("ORCA_WGRAD_SPLITS") ? atoi(getenv("ORCA_WGRAD_SPLITS")) : 0;
      if (forced > 0) splits = forced;
      splits = splits > cap ? cap : splits;
      for (int i = 0; i < np; ++i) {
        GemmArgs& g = w.g[i];
        const int kc = (g.K + splits - 1) / splits;
        g.kchunk = (kc + 15) / 16 * 16;
        g.gz = (g.K + g.kchunk - 1) / g.kchunk;
        if (g.gz != w.g[0].gz) fits = false;       // (one k split for the whole launch: the problems share K)
        g.part = part + po2;              // (never more than the 16 or 32 pieces the caller sized `part` for)
        po2 += ((int64_t)g.gz * g.M * g.N + 63) / 64 * 64;
        g.gx = (g.N + 63) / 64;
        const int full = g.M >> 6, r = g.M & 63;
        g.gy = full + (r == 0 ? 0 : (r > 32 && r <= 48) ? 2 : 1);
        if (g.gx > 16 || g.gy > 16) fits = false;
        for (int mi = 0; mi < g.gy; ++mi) {
          int v = 4;
          if (mi >= full && r <= 48) v = r > 32 ? (mi == full ? 2 : 1) : (r > 16 ? 2 : 1);
          for (int bx = 0; bx < g.gx; ++bx) list.push_back(It{v, i << 8 | mi << 4 | bx});
        }
      }
      if (fits && (int)list.size() <= WG_ITEMS_MAX) {
        std::stable_sort(list.begin(), list.end(), [](const It& x, const It& y) { return x.v > y.v; });
        WgItems items{};
        items.n = (int)list.size();
        items.gz = w.g[0].gz;
        for (int i = 0; i < items.n; ++i) items.code[i] = (uint16_t)list[i].code;
        k_wgrad<<<(unsigned)(items.n * items.gz), 256, 0, s>>>(w, items);
        int fb = 0;
        for (int i = 0; i < np; ++i) {
          w.first[i] = fb;
          fb += (int)(((int64_t)w.g[i].M * w.g[i].N + 255) / 256);
        }
        w.first[np] = fb;
        k_splitk_reduce<<<(unsigned)fb, 256, 0, s>>>(w);
        return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
      }
    }
  }
  // under two blocks per CU: halve the tile, double the blocks (a block is one wave per SIMD: its chain of slabs is
  // latency-bound, and only more resident blocks hide that); under one: quarter it
  const int TM = tiles64 < 256 ? 16 : tiles64 < 512 ? 32 : 64;
  int first = 0;
  for (int i = 0; i < np; ++i) {
    grp.first[i] = first;
    grp.g[i].gy = (grp.g[i].M + TM - 1) / TM;
    first += grp.g[i].gx * grp.g[i].gy * grp.g[i].gz;
  }
  grp.first[np] = first;
  if (TM == 16) k_sgemm<16><<<(unsigned)first, 256, 0, s>>>(grp);
  else if (TM == 32) k_sgemm<32><<<(unsigned)first, 256, 0, s>>>(grp);
  else k_sgemm<64><<<(unsigned)first, 256, 0, s>>>(grp);
  if (any_split) {
    GemmGroup red{};
    int nr = 0, fb = 0;
    for (int i = 0; i < np; ++i) {
      if (grp.g[i].gz <= 1) continue;
      red.g[nr] = grp.g[i];
      red.first[nr++] = fb;
      fb += (int)(((int64_t)grp.g[i].M * grp.g[i].N + 255) / 256);
    }
    red.n = nr;
    red.first[nr] = fb;
    k_splitk_reduce<<<(unsigned)fb, 256, 0, s>>>(red);
  }
  return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
}

int orca_sgemm(hipStream_t s, int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
               int64_t sbn, float* C, int64_t scm, int64_t scn, const float* bias, float alpha, bool accumulate, bool relu,
               float* part, int splits) {
  OrcaGemm g{M, N, K, A, sam, sak, B, sbk, sbn, C, scm, scn, bias, alpha, accumulate ? 1 : 0, relu ? 1 : 0, nullptr, nullptr, 0,
             nullptr, part ? splits : 1};
  return orca_sgemm_group(s, &g, 1, part);
}

// up to COLSUM_JOBS column sums over the same Q rows in one launch pair; part: COLSUM_JOBS * 64 * COLSUM_MAXC floats
int orca_colsum_group(hipStream_t s, int n, const float* const* X, const int64_t* ld, const int* ncols, float* const* out,
                      int64_t Q, float alpha, bool accumulate, float* part) {
  if (n <= 0) return ORCA_OK;
  if (n > COLSUM_JOBS) {
    orca_set_error("orca_colsum_group: %d sums in one group (at most %d)", n, COLSUM_JOBS);
    return ORCA_EINVAL;
  }
  ColsumJobs jobs{};
  int maxc = 0;
  for (int i = 0; i < n; ++i) {
    if (ncols[i] <= 0 || ncols[i] > COLSUM_MAXC) {
      orca_set_error("orca_colsum_group: %d columns (1 .. %d supported)", ncols[i], COLSUM_MAXC);
      return ORCA_EINVAL;
    }
    jobs.X[i] = X[i]; jobs.ld[i] = ld[i]; jobs.ncols[i] = ncols[i]; jobs.out[i] = out[i];
    maxc = ncols[i] > maxc ? ncols[i] : maxc;
  }
  k_colsum<<<dim3((unsigned)((maxc + 63) / 64), COLSUM_SLICES, (unsigned)n), 1024, 0, s>>>(jobs, Q, part);
  k_colsum_finish<<<dim3((unsigned)((maxc + 3) / 4), (unsigned)n), 256, 0, s>>>(jobs, part, alpha, accumulate ? 1 : 0);
  return hipGetLastError() == hipSuccess ? ORCA_OK : ORCA_EHIP;
}

int orca_colsum(hipStream_t s, const float* X, int64_t Q, int64_t ld, int ncols, float* out, float alpha, bool accumulate,
                float* part) {
  if (ncols <= 0) return ORCA_OK;
  const int nc = ncols;
  return orca_colsum_group(s, 1, &X, &ld, &nc, &out, Q, alpha, accumulate, part);
}

namespace {
#define sgemm orca_sgemm
#define colsum orca_colsum

// ---- per-query pieces -------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// the slots of one query as the forward kernel reads them (lane j = slot j)
// lane j's value as a wave-uniform scalar (j uniform): v_readlane into an SGPR.  __shfl is an LDS round trip per call
// (ds_bpermute) and leaves the result -- here row numbers that become load addresses -- in vector registers: the
// attention kernels below made 219 of them per query and kept a 64-bit address pair per row load (62 spilled VGPRs).
__device__ __forceinline__ int lane_value(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ float lane_value(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
// sum over the wave on the DPP path (an inclusive scan inside each row of 16 lanes, then the row totals passed on):
// six dependent VALU adds instead of six LDS round trips; the total is read from lane 63 as a uniform scalar.
// (Another order of additions than wave_sum's butterfly: used by the attention backward only.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_scalar(float v) {
  v = dpp_add<0x111, 0xf>(v);      // row_shr:1
  v = dpp_add<0x112, 0xf>(v);      // row_shr:2
  v = dpp_add<0x114, 0xf>(v);      // row_shr:4
  v = dpp_add<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds the row's sum
  v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return lane_value(v, 63);
}
struct Slots {
  int nrow, erow, ovl;
  float dt;
  bool masked;
};
// (Two levels of loads, each issued as a whole and unconditionally -- an optional array that is absent is replaced by a
// valid address and its value discarded, lanes past K load slot 0: written with an `if` per optional input every load
// was waited for where it was issued, five dependent round trips instead of two.)
__device__ __forceinline__ Slots load_slots(const orca_tgat_inputs& in, int64_t q, int K, int lane) {
  const int64_t p = q * K + (lane < K ? lane : 0);
  const int32_t* any32 = reinterpret_cast<const int32_t*>(in.edge_table);
  const uint8_t* any8 = reinterpret_cast<const uint8_t*>(in.edge_table);
  const int nr = (in.nbr_idx ? in.nbr_idx : any32)[in.nbr_idx ? p : 0];
  const int er = (in.edge_idx ? in.edge_idx : any32)[in.edge_idx ? p : 0];
  const int id = (in.nbr_ids ? in.nbr_ids : any32)[in.nbr_ids ? p : 0];
  const uint8_t mk = (in.nbr_ids ? any8 : in.mask)[in.nbr_ids ? 0 : p];
  const float dt = (in.dt ? in.dt : in.edge_table)[in.dt ? p : 0];
  Slots S{0, 0, 0, 0.f, true};
  const int nrow = in.nbr_idx ? nr : (int)p;
  const uint8_t fl = (in.nbr_flags ? in.nbr_flags : any8)[in.nbr_flags ? nrow : 0];
  if (lane < K) {
    S.nrow = nrow;
    S.erow = in.edge_idx ? er : (int)p;
    S.masked = in.nbr_ids ? (id == 0) : (mk != 0);
    if (in.dt) S.dt = dt;
    if (in.nbr_flags) S.ovl = fl ? 1 : 0;
  }
  return S;
}
// A key row [nbr (100) | edge (d_e) | time (100)] as NS register slots per lane, one kind of column per slot (no
// per-column branches: every load is unconditional and clamped): slots 0-1 the neighbour's row (columns lane,
// lane + 64), 2 .. 1+MAXE the edge row, the last two the time encoding.
constexpr int MAXE = (ORCA_MAX_DE + 63) / 64, NS = MAXE + 4;
__device__ __forceinline__ int slot_col(int s, int lane, int d_e) {
  return s < 2 ? lane + 64 * s : s < 2 + MAXE ? D + lane + 64 * (s - 2) : D + d_e + lane + 64 * (s - 2 - MAXE);
}
__device__ __forceinline__ bool slot_ok(int s, int lane, int d_e) {
  return s < 2 ? lane + 64 * s < D : s < 2 + MAXE ? lane + 64 * (s - 2) < d_e : lane + 64 * (s - 2 - MAXE) < D;
}
struct TimeCols { float w0, w1, b0, b1; };      // this lane's columns of the time encoder (gather mode)
__device__ __forceinline__ TimeCols load_time_cols(const orca_tgat_inputs& in, int lane) {
  TimeCols t{0.f, 0.f, 0.f, 0.f};
  if (in.dt) {
    const int c1 = lane + 64 < D ? lane + 64 : 0;
    t.w0 = in.time_w[lane]; t.b0 = in.time_b[lane];
    t.w1 = in.time_w[c1]; t.b1 = in.time_b[c1];
  }
  return t;
}
__device__ __forceinline__ void key_row(float (&x)[NS], const orca_tgat_inputs& in, const Slots& S, const TimeCols& tc,
                                        int64_t q, int j, int K, int d_e, int lane) {
  const int nr = lane_value(S.nrow, j), er = lane_value(S.erow, j), ov = lane_value(S.ovl, j);      // (j is wave-uniform)
  const float dt = lane_value(S.dt, j);
  const float* np = (ov ? in.nbr_overlay : in.nbr_table) + (int64_t)nr * D;      // scalar bases, per-lane 32-bit offsets
  const float* ep = in.edge_table + (int64_t)er * d_e;
  const bool hi = lane + 64 < D;
  // (a uniform base plus an UNSIGNED 32-bit byte offset is what selects the scalar-base form of the load)
  auto at = [](const float* base, unsigned col) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + col * 4u); };
  x[0] = at(np, (unsigned)lane);
  x[1] = at(np, hi ? (unsigned)lane + 64u : 0u);
  if (!hi) x[1] = 0.f;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int c = lane + 64 * i;
    x[2 + i] = at(ep, c < d_e ? (unsigned)c : 0u);
    if (c >= d_e) x[2 + i] = 0.f;
  }
  if (in.dt) {
    x[2 + MAXE] = orca_cosf(orca_time_arg(dt, tc.w0, tc.b0));
    x[3 + MAXE] = hi ? orca_cosf(orca_time_arg(dt, tc.w1, tc.b1)) : 0.f;
  } else {
    const float* tp = in.nbr_time + (q * K + j) * D;
    x[2 + MAXE] = tp[lane];
    x[3 + MAXE] = tp[hi ? lane + 64 : 0];
    if (!hi) x[3 + MAXE] = 0.f;
  }
}
// a [2 x kdim] row of the workspace (q~, dxbar, ...) in the same slots
__device__ __forceinline__ void load_slots2(float (&v0)[NS], float (&v1)[NS], const float* __restrict__ p, int kdim, int d_e, int lane) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const bool ok = slot_ok(s, lane, d_e);
    const int c = ok ? slot_col(s, lane, d_e) : 0;
    v0[s] = p[c];
    v1[s] = p[kdim + c];
    if (!ok) v0[s] = v1[s] = 0.f;
  }
}

// xq[q] = [src | src_time] (the q projection's input), x1[q][200:300] = src (the merger's second input)
__global__ __launch_bounds__(256) void k_bwd_prep(orca_tgat_inputs in, int64_t Q, float* __restrict__ xq,
                                                  float* __restrict__ x1) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int64_t row = in.src_idx ? (int64_t)in.src_idx[q] : q;
  const float* tab = (in.src_flags && in.src_flags[row]) ? in.src_overlay : in.src_table;
  for (int c = lane; c < D; c += 64) {
    const float s = tab[row * D + c];
    xq[q * QD + c] = s;
    x1[q * 3 * D + QD + c] = s;
    xq[q * QD + D + c] = in.dt ? orca_cosf(orca_time_arg(0.0f, in.time_w[c], in.time_b[c])) : in.src_time[q * D + c];
  }
}

// recompute of the attention core: a[q][h][j] = softmax_j(q~_h . x_j) over the unmasked slots,
// xbar[q][h] = sum_j a_hj x_j;  invalid[q] = 1 for an all-padded query (temporal_attention.py:46-47,52).
// One wave per query; a wave is alone with its chain of row gathers (the launch is a single round of 16 waves per CU),
// so the rows are fetched FIVE at a time and visited once: scores of the five, then an online softmax (running maximum,
// accumulators rescaled when it moves) adds them to the weighted sums -- two dependent memory round trips per query
// instead of twenty (one row per trip, two passes: 35 us per launch).  KT = compile-time bound of K.
constexpr int AF_ROWS = 5;
template <int KT>
__global__ __launch_bounds__(256, 4) void k_bwd_attn_fwd(orca_tgat_inputs in, int64_t Q, int K, int d_e,
                                                         const float* __restrict__ qt, float* __restrict__ a_out,
                                                         float* __restrict__ xbar, float* __restrict__ invalid,
                                                         float* __restrict__ wsum) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int kdim = QD + d_e;
  const Slots S = load_slots(in, q, K, lane);
  const TimeCols tc = load_time_cols(in, lane);
  unsigned long long live = __ballot(lane < K && !S.masked);
  const bool inv = live == 0ull;
  if (inv) live = 1ull;
  float q0[NS], q1[NS], b0[NS], b1[NS];
  load_slots2(q0, q1, qt + q * 2 * kdim, kdim, d_e, lane);
#pragma unroll
  for (int s = 0; s < NS; ++s) b0[s] = b1[s] = 0.f;
  const bool drop = in.dropout_p > 0.f;
  const float dsc = drop ? 1.0f / (1.0f - in.dropout_p) : 1.f;
  float s0 = 0.f, s1 = 0.f;                   // lane j keeps the scores of slot j
  float m0 = -INFINITY, m1 = -INFINITY;       // running maxima, sums of exponentials, sums of kept weights
  float l0 = 0.f, l1 = 0.f, w0s = 0.f, w1s = 0.f;
#pragma unroll
  for (int jb = 0; jb < KT; jb += AF_ROWS) {
    if (jb >= K) break;                       // wave-uniform
    float x[AF_ROWS][NS], p0[AF_ROWS], p1[AF_ROWS];
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) key_row(x[u], in, S, tc, q, jb + u < K ? jb + u : 0, K, d_e, lane);
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) {
      p0[u] = p1[u] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) { p0[u] += x[u][s] * q0[s]; p1[u] += x[u][s] * q1[s]; }
    }
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) { p0[u] = wave_sum(p0[u]); p1[u] = wave_sum(p1[u]); }
    float n0 = m0, n1 = m1;
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) {
      const bool ok = jb + u < K && ((live >> (jb + u)) & 1ull);
      if (ok) { n0 = fmaxf(n0, p0[u]); n1 = fmaxf(n1, p1[u]); }
      if (lane == jb + u) { s0 = p0[u]; s1 = p1[u]; }
    }
    const float r0 = m0 == -INFINITY ? 0.f : expf(m0 - n0), r1 = m1 == -INFINITY ? 0.f : expf(m1 - n1);
    l0 *= r0; l1 *= r1; w0s *= r0; w1s *= r1;
#pragma unroll
    for (int s = 0; s < NS; ++s) { b0[s] *= r0; b1[s] *= r1; }
    m0 = n0; m1 = n1;
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) {
      const bool ok = jb + u < K && ((live >> (jb + u)) & 1ull);      // wave-uniform
      const float e0 = ok ? expf(p0[u] - m0) : 0.f, e1 = ok ? expf(p1[u] - m1) : 0.f;
      l0 += e0; l1 += e1;
      float w0 = e0, w1 = e1;     // dropout acts on what is summed, not on the softmax output itself
      if (drop) {
        w0 *= orca_dropout_mult(in.dropout_seed, in.dropout_q0 + q, 0, jb + u, in.dropout_p, dsc);
        w1 *= orca_dropout_mult(in.dropout_seed, in.dropout_q0 + q, 1, jb + u, in.dropout_p, dsc);
      }
      w0s += w0; w1s += w1;
#pragma unroll
      for (int s = 0; s < NS; ++s) { b0[s] += w0 * x[u][s]; b1[s] += w1 * x[u][s]; }
    }
  }
  const float i0 = 1.0f / l0, i1 = 1.0f / l1;
  if (lane < K) {     // the softmax output itself (its backward needs it)
    const bool ok = (live >> lane) & 1ull;
    a_out[(q * 2 + 0) * K + lane] = ok ? expf(s0 - m0) * i0 : 0.f;
    a_out[(q * 2 + 1) * K + lane] = ok ? expf(s1 - m1) * i1 : 0.f;
  }
  if (drop && lane == 0) { wsum[q * 2] = w0s * i0; wsum[q * 2 + 1] = w1s * i1; }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (slot_ok(s, lane, d_e)) {
      const int c = slot_col(s, lane, d_e);
      xbar[q * 2 * kdim + c] = b0[s] * i0;
      xbar[q * 2 * kdim + kdim + c] = b1[s] * i1;
    }
  }
  if (lane == 0) invalid[q] = inv ? 1.f : 0.f;
}

// dropout: ocat[q][h*100 + c] += wsum[q][h] * bv[h*100 + c]  (the value bias reaches the output times the weight sum)
__global__ void k_add_scaled_bias(float* __restrict__ ocat, int64_t Q, const float* __restrict__ wsum,
                                  const float* __restrict__ bv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q * QD) return;
  const int64_t q = i / QD;
  const int c = (int)(i % QD);
  ocat[i] += wsum[q * 2 + c / D] * bv[c];
}
// dropout: scaled[q][c] = docat[q][c] * wsum[q][h] (for the bv gradient); dsum[q][h] = docat[q][h] . bv_h
__global__ __launch_bounds__(256) void k_bias_scale_bwd(const float* __restrict__ docat, int64_t Q, const float* __restrict__ wsum,
                                                        const float* __restrict__ bv, float* __restrict__ scaled,
                                                        float* __restrict__ dsum) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  for (int h = 0; h < 2; ++h) {
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float g = docat[q * QD + h * D + c];
      scaled[q * QD + h * D + c] = g * wsum[q * 2 + h];
      acc += g * bv[h * D + c];
    }
    acc = wave_sum(acc);
    if (lane == 0) dsum[q * 2 + h] = acc;
  }
}
// LayerNorm backward (eps 1e-5, biased variance): dy from g = dL/dout, gx = g * xhat for the affine gradients
__global__ __launch_bounds__(256) void k_bwd_ln(const float* __restrict__ y, const float* __restrict__ g,
                                                const float* __restrict__ gamma, int64_t Q, float* __restrict__ dy,
                                                float* __restrict__ gx) {
  const int lane = threadIdx.x & 63, c1 = lane + 64;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const float ya = y[q * D + lane], yb = c1 < D ? y[q * D + c1] : 0.f;
  const float mean = wave_sum(ya + yb) * (1.0f / D);
  const float da = ya - mean, db = c1 < D ? yb - mean : 0.f;
  const float var = wave_sum(da * da + db * db) * (1.0f / D);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const float xa = da * rstd, xb = db * rstd;
  const float ga = g[q * D + lane], gb = c1 < D ? g[q * D + c1] : 0.f;
  const float ha = ga * gamma[lane], hb = c1 < D ? gb * gamma[c1] : 0.f;
  const float m1 = wave_sum(ha + hb) * (1.0f / D);
  const float m2 = wave_sum(ha * xa + hb * xb) * (1.0f / D);
  dy[q * D + lane] = (ha - m1 - xa * m2) * rstd;
  gx[q * D + lane] = ga * xa;
  if (c1 < D) {
    dy[q * D + c1] = (hb - m1 - xb * m2) * rstd;
    gx[q * D + c1] = gb * xb;
  }
}

__device__ __forceinline__ float mul_rounded(float a, float b) {     // a product that is never fused into an fma
#pragma clang fp contract(off)
  return a * b;
}
// backward of the attention core.  In: dxbar[q][h] = g_h (gradient of the weighted key sums), a, q~.  Out:
// dqt[q][h] = sum_j ds_hj x_j with ds = softmax backward of da_hj = g_h . x_j;  the key-row gradient
// dx_j = sum_h (a_hj g_h + ds_hj q~_h) goes to d_nbr (columns 0..99), is dropped for the edge columns (the edge
// features are data, not parameters) and for the time columns goes to d_nbr_time (dense mode) or, through
// d cos(dt w + b), into this query's row of tgrad = [d time_w | d time_b] (gather mode).
// The key rows are visited ONCE, five at a time (as in k_bwd_attn_fwd): with c_h = sum_i a_hi da_hi,
// sum_j ds_hj x_j = sum_j a_hj da_hj x_j - c_h sum_j a_hj x_j, two sums that need no second look at the rows, and dx_j
// needs no x_j at all.  (Keeping all K rows in registers for two passes took 197 VGPRs: two rounds of waves, 49 us.)
template <int KT>
__global__ __launch_bounds__(256, 4) void k_bwd_attn_bwd(orca_tgat_inputs in, int64_t Q, int K, int d_e,
                                                         const float* __restrict__ qt, const float* __restrict__ a_in,
                                                         const float* __restrict__ dxbar, float* __restrict__ dqt,
                                                         float* __restrict__ d_nbr, float* __restrict__ d_nbr_time,
                                                         float* __restrict__ tgrad, const float* __restrict__ dsum) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int kdim = QD + d_e;
  const Slots S = load_slots(in, q, K, lane);
  const TimeCols tc = load_time_cols(in, lane);
  float g0[NS], g1[NS], t0[NS], t1[NS], u0[NS], u1[NS];
  load_slots2(g0, g1, dxbar + q * 2 * kdim, kdim, d_e, lane);
#pragma unroll
  for (int s = 0; s < NS; ++s) t0[s] = t1[s] = u0[s] = u1[s] = 0.f;
  const float a0 = lane < K ? a_in[(q * 2 + 0) * K + lane] : 0.f;
  const float a1 = lane < K ? a_in[(q * 2 + 1) * K + lane] : 0.f;
  // dropout: the key sums (and each head's weight sum, whose gradient is dsum) were taken with a' = a * mult
  float m0 = 1.f, m1 = 1.f, ds0 = 0.f, ds1 = 0.f;
  if (in.dropout_p > 0.f) {
    const float sc = 1.0f / (1.0f - in.dropout_p);
    const int j = lane < K ? lane : 0;
    m0 = orca_dropout_mult(in.dropout_seed, in.dropout_q0 + q, 0, j, in.dropout_p, sc);
    m1 = orca_dropout_mult(in.dropout_seed, in.dropout_q0 + q, 1, j, in.dropout_p, sc);
    ds0 = dsum[q * 2];
    ds1 = dsum[q * 2 + 1];
  }
  float da0 = 0.f, da1 = 0.f;      // lane j: da_hj - z_h
  // (softmax backward does not see a constant z_h added to da_h.: slot 0's value is taken off, which keeps the two sums
  // below -- they cancel against each other -- small when the da share a large common part)
  float z0 = 0.f, z1 = 0.f;
#pragma unroll
  for (int jb = 0; jb < KT; jb += AF_ROWS) {
    if (jb >= K) break;            // wave-uniform
    float x[AF_ROWS][NS], p0[AF_ROWS], p1[AF_ROWS];
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) key_row(x[u], in, S, tc, q, jb + u < K ? jb + u : 0, K, d_e, lane);
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) {
      p0[u] = p1[u] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) { p0[u] += x[u][s] * g0[s]; p1[u] += x[u][s] * g1[s]; }
    }
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) { p0[u] = wave_sum_scalar(p0[u]); p1[u] = wave_sum_scalar(p1[u]); }
    if (jb == 0) {
      z0 = (p0[0] + ds0) * lane_value(m0, 0);
      z1 = (p1[0] + ds1) * lane_value(m1, 0);
    }
#pragma unroll
    for (int u = 0; u < AF_ROWS; ++u) {
      const bool ok = jb + u < K;
      const int j = ok ? jb + u : 0;
      const float w0 = ok ? lane_value(a0, j) : 0.f, w1 = ok ? lane_value(a1, j) : 0.f;
      const float d0 = (p0[u] + ds0) * lane_value(m0, j) - z0, d1 = (p1[u] + ds1) * lane_value(m1, j) - z1;
      if (lane == jb + u) { da0 = d0; da1 = d1; }
      const float e0 = w0 * d0, e1 = w1 * d1;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        t0[s] += e0 * x[u][s]; t1[s] += e1 * x[u][s];
        u0[s] += w0 * x[u][s]; u1[s] += w1 * x[u][s];
      }
    }
  }
  const float c0 = wave_sum_scalar(a0 * da0), c1 = wave_sum_scalar(a1 * da1);
  const float s0 = a0 * (da0 - c0), s1 = a1 * (da1 - c1);      // lane j: ds_hj
  const float aw0 = a0 * m0, aw1 = a1 * m1;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (slot_ok(s, lane, d_e)) {
      const int c = slot_col(s, lane, d_e);
      // (products rounded before the subtraction, not fused into it: with a single live slot both terms are the same
      // rounded number and the gradient is exactly zero, as autograd's is)
      dqt[q * 2 * kdim + c] = t0[s] - mul_rounded(c0, u0[s]);
      dqt[q * 2 * kdim + kdim + c] = t1[s] - mul_rounded(c1, u1[s]);
    }
  }
  // the key-row gradients: dx_j = a'_0j g_0 + a'_1j g_1 + ds_0j q~_0 + ds_1j q~_1 on the neighbour and time columns
  float q0[NS], q1[NS];
  load_slots2(q0, q1, qt + q * 2 * kdim, kdim, d_e, lane);
  float tw[2] = {0.f, 0.f}, tb[2] = {0.f, 0.f};
  const bool hi = lane + 64 < D;
  // (nobody reads the key-row gradients when the neighbour rows are detached history rows and the time encoder is
  // frozen: d_nbr == NULL skips 400 B of stores per slot)
  if (!d_nbr && !d_nbr_time && !tgrad) return;
  for (int j = 0; j < K; ++j) {
    const float w0 = lane_value(aw0, j), w1 = lane_value(aw1, j), e0 = lane_value(s0, j), e1 = lane_value(s1, j);
    const float dt = lane_value(S.dt, j);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int sn = h2, st = 2 + MAXE + h2;
      const float dxn = w0 * g0[sn] + w1 * g1[sn] + e0 * q0[sn] + e1 * q1[sn];
      const float dxt = w0 * g0[st] + w1 * g1[st] + e0 * q0[st] + e1 * q1[st];
      if (h2 == 0 || hi) {
        const int c = lane + 64 * h2;
        if (d_nbr) d_nbr[(q * K + j) * D + c] = dxn;
        if (in.dt) {
          if (tgrad) {
            const float dth = -orca_sinf(orca_time_arg(dt, h2 ? tc.w1 : tc.w0, h2 ? tc.b1 : tc.b0)) * dxt;
            tw[h2] += dt * dth;
            tb[h2] += dth;
          }
        } else if (d_nbr_time) {
          d_nbr_time[(q * K + j) * D + c] = dxt;
        }
      }
    }
  }
  if (tgrad) {
    tgrad[q * QD + lane] = tw[0];
    tgrad[q * QD + D + lane] = tb[0];
    if (hi) {
      tgrad[q * QD + lane + 64] = tw[1];
      tgrad[q * QD + D + lane + 64] = tb[1];
    }
  }
}

// d_src[q] = dx1[q][200:300] + dxq[q][0:100];  the source's time columns dxq[q][100:200] go to d_src_time (dense
// mode) or, through d cos(0*w + b) = -sin(b), are added to the time_b half of tgrad (gather mode)
__global__ void k_bwd_finish(orca_tgat_inputs in, int64_t Q, const float* __restrict__ dx1,
                             const float* __restrict__ dxq, float* __restrict__ d_src, float* __restrict__ d_src_time,
                             float* __restrict__ tgrad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q * D) return;
  const int64_t q = i / D;
  const int c = (int)(i % D);
  if (d_src) d_src[i] = dx1[q * 3 * D + QD + c] + dxq[q * QD + c];
  const float dst = dxq[q * QD + D + c];
  if (in.dt) {
    if (tgrad) tgrad[q * QD + D + c] += -orca_sinf(orca_time_arg(0.0f, in.time_w[c], in.time_b[c])) * dst;
  } else if (d_src_time) {
    d_src_time[i] = dst;
  }
}

// ---- fused row chains (d_e = 172) ----------------------------------------------------------------------------
// Between the two per-query attention kernels the backward is a chain of activations x weights products, each row of
// every product depending on the same row of the one before.  As launches of their own they are bound by the floor
// under a launch of this size (8-22 us each, 12 per layer); here a 1024-thread workgroup takes 16 rows through a whole
// chain, the activations staying in LDS (row stride / 4 odd: float4 reads of 16 rows hit distinct banks) and each
// weight fragment read once per workgroup, straight from L2 into MFMA operands as in k_sgemm_rows.  What the
// weight-gradient GEMMs and the attention kernels need is written to the same workspace rows as before.
//   chain A: prep -> qs = 0.1 (xq Wq^T + bq) -> q~_h = qs_h Wk_h
//   chain B: ocat -> out-proj -> fc1 -> fc2 -> LayerNorm backward -> fc2^T -> fc1^T -> out-proj^T -> value map^T
//   chain C: dqs_h = 0.1 dq~_h Wk_h^T -> dxq = dq Wq -> d_src / time gradients
constexpr int CH_WAVES = 16, CH_T = 64 * CH_WAVES;

// One stage: C_h [16 x N] = A[:, h*ash : h*ash + K] B_h for h < NH.  A in LDS (row stride lda).  B_h is a weight
// matrix, used transposed (KFAST: the matrix is k-contiguous, a wave's tile is 16 columns) or as stored (a 32-column
// tile, two columns per lane).  The weights are read from a copy in MFMA-fragment order (k_chain_pack, one launch per
// backward call): every wave load is 1 KB contiguous -- straight from the row-major tensors a load touched 16 half
// lines and the stages ran at 34 GB/s per CU, half of what packed weights stream at.  The NH*NT tiles x KS reduction
// pieces ("items") are dealt to the 16 waves, round by round; an item's weights arrive in chunks of 8 fragments
// (Frag), and the NEXT chunk -- of this item, of the wave's next item, or the first of the NEXT STAGE -- is requested
// before the current one is multiplied.  Rounds and chunks are compile-time and fully unrolled (no loop-carried
// loads: s_waitcnt placement stays exact).  KS > 1: the pieces meet in P and are added in a fixed order.
// epi(h, m, n, value) stores an element; run() ends with a barrier.
struct Frag { float v[32]; };

template <int N, int K, int NH, bool KFAST, int KS>
struct ChainStage {
  static_assert(K % 4 == 0 && N % 2 == 0, "float4 rows of A, two columns per lane");
  static constexpr int TW = KFAST ? 16 : 32;               // columns of a wave's tile
  static constexpr int NT = (N + TW - 1) / TW, NP = NT * TW;
  static constexpr int KB = (K + 15) / 16;                 // 16-deep blocks of the reduction
  static constexpr int BPI = (KB + KS - 1) / KS;           // blocks per item
  static constexpr int ITEMS = NH * NT * KS;
  static constexpr int PFLOATS = KS > 1 ? KS * NH * 16 * NP : 0;
  static constexpr int FPB = KFAST ? 1 : 2;                // fragments ([64 lanes][4 floats]) per block
  static constexpr int CB = 8 / FPB;                       // blocks per chunk
  static constexpr int CPI = (BPI + CB - 1) / CB;          // chunks per item
  static constexpr int R = (ITEMS + CH_WAVES - 1) / CH_WAVES;
  static constexpr int FRAGS = NH * NT * KB * FPB;         // of the packed copy: [tile = h*NT + nt][block][FPB]
  const float* Wp;

  // requests chunk ch of an item's weights (item and block clamped: never skipped; what lies past K is multiplied by
  // zeros of A)
  __device__ __forceinline__ void load(Frag& f, int item, int ch) const {
    const int lane = threadIdx.x & 63;
    item = item < ITEMS ? item : ITEMS - 1;
    const int ks = item / (NH * NT), tile = item % (NH * NT);
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      if (ch * CB + i >= BPI) continue;
      int kb = ks * BPI + ch * CB + i;
      if (KS * BPI > KB) kb = kb < KB ? kb : KB - 1;
      const float* bp = Wp + ((int64_t)(tile * KB + kb) * FPB * 64 + lane) * 4;
#pragma unroll
      for (int u = 0; u < FPB; ++u) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(bp + u * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) f.v[4 * (i * FPB + u) + e] = t[e];
      }
    }
  }

  // f: chunk 0 of item `wave`, requested earlier; on return f holds whatever next(f) requested (the next stage's first)
  template <class Epi, class Next>
  __device__ __forceinline__ void run(Frag& f, const float* As, int lda, int ash, float* P, Epi epi, Next next) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int item = wave + CH_WAVES * rr;
      const bool active = item < ITEMS;                    // wave-uniform
      const int itc = active ? item : ITEMS - 1;
      const int ks = itc / (NH * NT), tile = itc % (NH * NT), h = tile / NT, nt = tile % NT;
      const float* ap = As + r * lda + h * ash + 4 * kq;
      const int n = KFAST ? nt * 16 + r : nt * 32 + 2 * r;
      const bool nok = n < N;
      f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
      for (int ch = 0; ch < CPI; ++ch) {
        Frag g;
        if (ch + 1 < CPI) load(g, itc, ch + 1);
        else if (rr + 1 < R) load(g, item + CH_WAVES, 0);
        else next(g);
        __builtin_amdgcn_sched_barrier(0);     // the requests stay ahead of the MFMAs (the scheduler sinks them to their use)
        if (active) {
#pragma unroll
          for (int i = 0; i < CB; ++i) {
            if (ch * CB + i >= BPI) continue;
            const int k0 = 16 * (ks * BPI + ch * CB + i);
            const bool ok = k0 + 4 * kq < K;
            f32x4 a = *reinterpret_cast<const f32x4*>(ap + (ok ? k0 : 0));
            if (!ok) a = zero4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {       // (the packed weights hold zeros past N and K)
              if constexpr (KFAST) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], f.v[4 * i + e], acc0, 0, 0, 0);
              } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], f.v[8 * i + 2 * e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], f.v[8 * i + 2 * e + 1], acc1, 0, 0, 0);
              }
            }
          }
        }
        f = g;
      }
      if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = 4 * kq + i;
          if constexpr (KS == 1) {
            if (nok) {
              epi(h, m, n, acc0[i]);
              if constexpr (!KFAST) epi(h, m, n + 1, acc1[i]);
            }
          } else if constexpr (KFAST) {
            P[((ks * NH + h) * 16 + m) * NP + n] = acc0[i];
          } else {
            *reinterpret_cast<float2*>(&P[((ks * NH + h) * 16 + m) * NP + n]) = make_float2(acc0[i], acc1[i]);
          }
        }
      }
    }
    if constexpr (KS > 1) {
      __syncthreads();
      for (int idx = threadIdx.x; idx < NH * 16 * N; idx += CH_T) {
        const int h = idx / (16 * N), m = (idx % (16 * N)) / N, n = idx % N;
        float v = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) v += P[((ks * NH + h) * 16 + m) * NP + n];
        epi(h, m, n, v);
      }
    }
    __syncthreads();
  }
};

// the stages of the three chains and where their packed weights start (in fragments of 256 floats)
template <int KD> struct ChainW {
  using A1 = ChainStage<QD, QD, 1, true, 2>;      // qs = xq Wq^T
  using A2 = ChainStage<KD, D, 2, false, 1>;      // q~_h = qs_h Wk_h
  using B1 = ChainStage<D, KD, 2, true, 3>;       // ocat_h = xbar_h Wv_h^T
  using B2 = ChainStage<QD, QD, 1, true, 2>;      // o = ocat Wo^T
  using B3 = ChainStage<D, 3 * D, 1, true, 2>;    // h1 = x1 W1^T
  using B4 = ChainStage<D, D, 1, true, 2>;        // y = h1 W2^T
  using B5 = ChainStage<D, D, 1, false, 4>;       // dz1 = dy W2
  using B6 = ChainStage<3 * D, D, 1, false, 1>;   // dx1 = dz1 W1
  using B7 = ChainStage<QD, QD, 1, false, 2>;     // docat = do Wo
  using B8 = ChainStage<KD, D, 2, false, 1>;      // dxbar_h = docat_h Wv_h
  using C1 = ChainStage<D, KD, 2, true, 3>;       // dqs_h = dq~_h Wk_h^T
  using C2 = ChainStage<QD, QD, 1, false, 2>;     // dxq = dq Wq
  static constexpr int a1 = 0, a2 = a1 + A1::FRAGS, b1 = a2 + A2::FRAGS, b2 = b1 + B1::FRAGS, b3 = b2 + B2::FRAGS,
                       b4 = b3 + B3::FRAGS, b5 = b4 + B4::FRAGS, b6 = b5 + B5::FRAGS, b7 = b6 + B6::FRAGS,
                       b8 = b7 + B7::FRAGS, c1 = b8 + B8::FRAGS, c2 = c1 + C1::FRAGS, total = c2 + C2::FRAGS;
};
// fragment-order copies of the weights: fragment = [64 lanes][4 floats]; lane (r = lane & 15, kq = lane >> 4).
// KFAST (tile of 16 columns): fragment (tile, block) element j = B(16 block + 4 kq + j, 16 tile + r);
// otherwise (tile of 32 columns, two fragments per block): fragment (tile, block, u) element j =
// B(16 block + 4 kq + 2 u + (j >> 1), 32 tile + 2 r + (j & 1)).  Zeros past N and K.  B_h(k, n) = src[h*sh + k*sbk + n*sbn].
struct ChainPackJob {
  const float* src;
  int sbk, sbn, sh, N, K, NT, KB, kfast, first;
};
constexpr int CHAIN_PACK_JOBS = 12;
struct ChainPack { ChainPackJob j[CHAIN_PACK_JOBS]; int total; };
__global__ __launch_bounds__(256) void k_chain_pack(ChainPack jobs, float* __restrict__ dst) {
  const int gid = blockIdx.x * 256 + threadIdx.x, frag = gid >> 6, lane = gid & 63;
  if (frag >= jobs.total) return;
  int p = 0;
#pragma unroll
  for (int i = 1; i < CHAIN_PACK_JOBS; ++i) p += frag >= jobs.j[i].first ? 1 : 0;
  const ChainPackJob& J = jobs.j[p];
  const int local = frag - J.first, r = lane & 15, kq = lane >> 4;
  const int fpb = J.kfast ? 1 : 2;
  const int tile = local / (J.KB * fpb), kb = (local / fpb) % J.KB, u = local % fpb;
  const int h = tile / J.NT, nt = tile % J.NT;
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 16 * kb + 4 * kq + (J.kfast ? j : 2 * u + (j >> 1));
    const int n = J.kfast ? 16 * nt + r : 32 * nt + 2 * r + (j & 1);
    v[j] = (k < J.K && n < J.N) ? J.src[(int64_t)h * J.sh + (int64_t)k * J.sbk + (int64_t)n * J.sbn] : 0.f;
  }
  *reinterpret_cast<f32x4*>(dst + (int64_t)gid * 4) = v;
}
template <class S> ChainPackJob chain_pack_job(const float* src, int sbk, int sbn, int sh, int N, int K, bool kfast, int first) {
  return ChainPackJob{src, sbk, sbn, sh, N, K, S::NT, S::KB, kfast ? 1 : 0, first};
}

constexpr int CH_S2 = QD + 4, CH_S3 = 3 * D + 8, CH_S1 = D + 8;     // LDS row strides of 200-, 300- and 100-wide rows
template <int KD> struct ChainLds {
  static constexpr int SB = 2 * KD + 4;                                // stride of a [2 x kdim] row
  static_assert((SB / 4) % 2 == 1 && (CH_S2 / 4) % 2 == 1 && (CH_S3 / 4) % 2 == 1 && (CH_S1 / 4) % 2 == 1, "bank spread");
  // chain A: xq, qs, P;  chain C: dq~ rows, dq, dxq, P;  chain B: xbar rows (later h1, y, dy, dh1), P, ocat/docat, x1/dx1
  static constexpr int A_P = ChainW<KD>::A1::PFLOATS;
  static constexpr int A_FLOATS = 2 * 16 * CH_S2 + A_P;
  static constexpr int C_P1 = ChainW<KD>::C1::PFLOATS, C_P2 = ChainW<KD>::C2::PFLOATS;
  static constexpr int C_P = C_P1 > C_P2 ? C_P1 : C_P2;
  static constexpr int C_FLOATS = 16 * SB + 2 * 16 * CH_S2 + C_P;
  static constexpr int B_P = ChainW<KD>::B1::PFLOATS > ChainW<KD>::B5::PFLOATS ? ChainW<KD>::B1::PFLOATS : ChainW<KD>::B5::PFLOATS;
  static_assert(ChainW<KD>::B2::PFLOATS <= B_P && ChainW<KD>::B3::PFLOATS <= B_P && ChainW<KD>::B7::PFLOATS <= B_P, "partial sums fit");
  static constexpr int B_FLOATS = 16 * SB + B_P + 16 * CH_S2 + 16 * CH_S3;
  static_assert(4 * 16 * CH_S1 <= 16 * SB, "the narrow rows alias the xbar rows");
};

template <int KD>
__global__ __launch_bounds__(CH_T) void k_chain_a(orca_tgat_inputs in, int64_t Q, const float* __restrict__ wp,
                                                  const float* __restrict__ bq, float* __restrict__ xq, float* __restrict__ x1, float* __restrict__ qs,
                                                  float* __restrict__ qt) {
  extern __shared__ __align__(16) float lds[];
  float* XQ = lds;
  float* QS = XQ + 16 * CH_S2;
  float* P = QS + 16 * CH_S2;
  __shared__ float sbq[QD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * 16;
  using W = ChainW<KD>;
  const typename W::A1 g1{wp + W::a1 * 256};
  const typename W::A2 g2{wp + W::a2 * 256};
  Frag f;
  g1.load(f, wave, 0);
  if (threadIdx.x < QD) sbq[threadIdx.x] = bq[threadIdx.x];
  {   // wave = row: xq = [src | src_time], x1[:, 200:300] = src   (rows past Q repeat the last one and store nothing)
    const int64_t q = m0 + wave, qc = q < Q ? q : Q - 1;
    const int64_t row = in.src_idx ? (int64_t)in.src_idx[qc] : qc;
    const float* tab = (in.src_flags && in.src_flags[row]) ? in.src_overlay : in.src_table;
    for (int c = lane; c < D; c += 64) {
      const float s = tab[row * D + c];
      const float tv = in.dt ? orca_cosf(orca_time_arg(0.0f, in.time_w[c], in.time_b[c])) : in.src_time[qc * D + c];
      XQ[wave * CH_S2 + c] = s;
      XQ[wave * CH_S2 + D + c] = tv;
      if (q < Q) {
        xq[q * QD + c] = s;
        xq[q * QD + D + c] = tv;
        x1[q * 3 * D + QD + c] = s;
      }
    }
  }
  __syncthreads();
  g1.run(f, XQ, CH_S2, 0, P, [&](int, int m, int n, float v) {
    v = 0.1f * (v + sbq[n]);
    QS[m * CH_S2 + n] = v;
    if (m0 + m < Q) qs[(m0 + m) * QD + n] = v;
  }, [&](Frag& g) { g2.load(g, wave, 0); });
  if (!qt) return;       // (block-uniform: the forward kept q~, orca_tgat_backward_saved)
  g2.run(f, QS, CH_S2, D, P, [&](int h, int m, int n, float v) {
    if (m0 + m < Q) qt[(m0 + m) * 2 * KD + h * KD + n] = v;
  }, [](Frag&) {});
}

struct ChainB {
  const float *wp, *bv, *o_b, *fc1_b, *fc2_b, *ln_w;                            // packed weights, biases
  const float *grad_out, *xbar, *invalid, *wsum;                                // inputs
  float *x1, *ocat, *h1, *dy, *gx, *dh1, *dx1, *docat, *dxbar, *scaled, *dsum;  // rows kept for the weight gradients
  int drop;
};
template <int KD>
__global__ __launch_bounds__(CH_T) void k_chain_b(ChainB c, int64_t Q) {
  using L = ChainLds<KD>;
  extern __shared__ __align__(16) float lds[];
  float* BIG = lds;                        // xbar rows; once they are consumed: h1, y, dy, dh1
  float* P = BIG + 16 * L::SB;
  float* OC = P + L::B_P;                  // ocat, later docat
  float* X1 = OC + 16 * CH_S2;             // x1, later dx1
  float *H1 = BIG, *Y = BIG + 16 * CH_S1, *DY = BIG + 2 * 16 * CH_S1, *DH = BIG + 3 * 16 * CH_S1;
  __shared__ float inval[16], ws[32], sbv[QD], sbo[QD], sb1[D], sb2[D], sg[D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * 16;
  using W = ChainW<KD>;
  const typename W::B1 g1{c.wp + W::b1 * 256};
  const typename W::B2 g2{c.wp + W::b2 * 256};
  const typename W::B3 g3{c.wp + W::b3 * 256};
  const typename W::B4 g4{c.wp + W::b4 * 256};
  const typename W::B5 g5{c.wp + W::b5 * 256};
  const typename W::B6 g6{c.wp + W::b6 * 256};
  const typename W::B7 g7{c.wp + W::b7 * 256};
  const typename W::B8 g8{c.wp + W::b8 * 256};
  Frag f;
  g1.load(f, wave, 0);
  if (threadIdx.x < QD) {
    sbv[threadIdx.x] = c.bv[threadIdx.x];
    sbo[threadIdx.x] = c.o_b[threadIdx.x];
  } else if (threadIdx.x < QD + D) {
    const int i = threadIdx.x - QD;
    sb1[i] = c.fc1_b[i];
    sb2[i] = c.fc2_b[i];
    sg[i] = c.ln_w[i];
  }
  for (int i = threadIdx.x; i < 16 * (2 * KD / 4); i += CH_T) {       // xbar rows, float4 at a time
    const int m = i / (2 * KD / 4), c4 = i % (2 * KD / 4);
    const int64_t q = m0 + m < Q ? m0 + m : Q - 1;
    *reinterpret_cast<f32x4*>(&BIG[m * L::SB + 4 * c4]) = *reinterpret_cast<const f32x4*>(&c.xbar[q * 2 * KD + 4 * c4]);
  }
  for (int i = threadIdx.x; i < 16 * D; i += CH_T) {                  // the merger's second input (written by chain A)
    const int m = i / D, n = i % D;
    const int64_t q = m0 + m < Q ? m0 + m : Q - 1;
    X1[m * CH_S3 + QD + n] = c.x1[q * 3 * D + QD + n];
  }
  if (threadIdx.x < 16) {
    const int64_t q = m0 + threadIdx.x < Q ? m0 + threadIdx.x : Q - 1;
    inval[threadIdx.x] = c.invalid[q];
    ws[2 * threadIdx.x] = c.drop ? c.wsum[q * 2] : 1.f;
    ws[2 * threadIdx.x + 1] = c.drop ? c.wsum[q * 2 + 1] : 1.f;
  }
  __syncthreads();
  // ocat_h = xbar_h Wv_h^T + bv_h (dropout: times the sum of the head's kept weights)
  g1.run(f, BIG, L::SB, KD, P, [&](int h, int m, int n, float v) {
    v += ws[2 * m + h] * sbv[h * D + n];
    OC[m * CH_S2 + h * D + n] = v;
    if (m0 + m < Q) c.ocat[(m0 + m) * QD + h * D + n] = v;
  }, [&](Frag& g) { g2.load(g, wave, 0); });
  // o = ocat Wo^T + bo, zero for a query without neighbours                          (temporal_attention.py:52)
  g2.run(f, OC, CH_S2, 0, P, [&](int, int m, int n, float v) {
    v = inval[m] != 0.f ? 0.f : v + sbo[n];
    X1[m * CH_S3 + n] = v;
    if (m0 + m < Q) c.x1[(m0 + m) * 3 * D + n] = v;
  }, [&](Frag& g) { g3.load(g, wave, 0); });
  // h1 = relu(x1 W1^T + b1), y = h1 W2^T + b2                                         (utils/util.py:22-24)
  g3.run(f, X1, CH_S3, 0, P, [&](int, int m, int n, float v) {
    v = fmaxf(v + sb1[n], 0.f);
    H1[m * CH_S1 + n] = v;
    if (m0 + m < Q) c.h1[(m0 + m) * D + n] = v;
  }, [&](Frag& g) { g4.load(g, wave, 0); });
  g4.run(f, H1, CH_S1, 0, P, [&](int, int m, int n, float v) { Y[m * CH_S1 + n] = v + sb2[n]; },
         [&](Frag& g) { g5.load(g, wave, 0); });
  {   // LayerNorm backward, wave = row (k_bwd_ln)
    const int m = wave, c1 = lane + 64;
    const int64_t q = m0 + m < Q ? m0 + m : Q - 1;
    const float ya = Y[m * CH_S1 + lane], yb = c1 < D ? Y[m * CH_S1 + c1] : 0.f;
    const float mean = wave_sum(ya + yb) * (1.0f / D);
    const float da = ya - mean, db = c1 < D ? yb - mean : 0.f;
    const float var = wave_sum(da * da + db * db) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float xa = da * rstd, xb = db * rstd;
    const float ga = c.grad_out[q * D + lane], gb = c1 < D ? c.grad_out[q * D + c1] : 0.f;
    const float ha = ga * sg[lane], hb = c1 < D ? gb * sg[c1] : 0.f;
    const float s1 = wave_sum(ha + hb) * (1.0f / D);
    const float s2 = wave_sum(ha * xa + hb * xb) * (1.0f / D);
    const float va = (ha - s1 - xa * s2) * rstd, vb = (hb - s1 - xb * s2) * rstd;
    DY[m * CH_S1 + lane] = va;
    if (c1 < D) DY[m * CH_S1 + c1] = vb;
    if (m0 + m < Q) {
      c.dy[q * D + lane] = va;
      c.gx[q * D + lane] = ga * xa;
      if (c1 < D) {
        c.dy[q * D + c1] = vb;
        c.gx[q * D + c1] = gb * xb;
      }
    }
  }
  __syncthreads();
  // fc2: dz1 = (dy W2) where h1 > 0;  fc1: dx1 = dz1 W1, its first 200 columns zero where the output was zeroed
  g5.run(f, DY, CH_S1, 0, P, [&](int, int m, int n, float v) {
    if (!(H1[m * CH_S1 + n] > 0.f)) v = 0.f;
    DH[m * CH_S1 + n] = v;
    if (m0 + m < Q) c.dh1[(m0 + m) * D + n] = v;
  }, [&](Frag& g) { g6.load(g, wave, 0); });
  g6.run(f, DH, CH_S1, 0, P, [&](int, int m, int n, float v) {
    if (n < QD && inval[m] != 0.f) v = 0.f;
    X1[m * CH_S3 + n] = v;
    if (m0 + m < Q) c.dx1[(m0 + m) * 3 * D + n] = v;
  }, [&](Frag& g) { g7.load(g, wave, 0); });
  // out-proj: docat = do Wo
  g7.run(f, X1, CH_S3, 0, P, [&](int, int m, int n, float v) {
    OC[m * CH_S2 + n] = v;
    if (m0 + m < Q) {
      c.docat[(m0 + m) * QD + n] = v;
      if (c.drop) c.scaled[(m0 + m) * QD + n] = v * ws[2 * m + n / D];      // the bv gradient's summand
    }
  }, [&](Frag& g) { g8.load(g, wave, 0); });
  if (c.drop) {   // dsum_h = docat_h . bv_h, wave = row (k_bias_scale_bwd)
    const int m = wave;
    for (int h = 0; h < 2; ++h) {
      float acc = 0.f;
      for (int n = lane; n < D; n += 64) acc += OC[m * CH_S2 + h * D + n] * sbv[h * D + n];
      acc = wave_sum(acc);
      if (lane == 0 && m0 + m < Q) c.dsum[(m0 + m) * 2 + h] = acc;
    }
  }
  // value map: dxbar_h = docat_h Wv_h
  g8.run(f, OC, CH_S2, D, P, [&](int h, int m, int n, float v) {
    if (m0 + m < Q) c.dxbar[(m0 + m) * 2 * KD + h * KD + n] = v;
  }, [](Frag&) {});
}

template <int KD>
__global__ __launch_bounds__(CH_T) void k_chain_c(orca_tgat_inputs in, int64_t Q, const float* __restrict__ wp,
                                                  const float* __restrict__ dqt,
                                                  const float* __restrict__ dx1, float* __restrict__ dq,
                                                  float* __restrict__ d_src, float* __restrict__ d_src_time,
                                                  float* __restrict__ tgrad) {
  using L = ChainLds<KD>;
  extern __shared__ __align__(16) float lds[];
  float* BIG = lds;
  float* DQ = BIG + 16 * L::SB;
  float* DXQ = DQ + 16 * CH_S2;
  float* P = DXQ + 16 * CH_S2;
  const int64_t m0 = (int64_t)blockIdx.x * 16;
  const int wave = threadIdx.x >> 6;
  using W = ChainW<KD>;
  const typename W::C1 g1{wp + W::c1 * 256};
  const typename W::C2 g2{wp + W::c2 * 256};
  Frag f;
  g1.load(f, wave, 0);
  for (int i = threadIdx.x; i < 16 * (2 * KD / 4); i += CH_T) {
    const int m = i / (2 * KD / 4), c4 = i % (2 * KD / 4);
    const int64_t q = m0 + m < Q ? m0 + m : Q - 1;
    *reinterpret_cast<f32x4*>(&BIG[m * L::SB + 4 * c4]) = *reinterpret_cast<const f32x4*>(&dqt[q * 2 * KD + 4 * c4]);
  }
  __syncthreads();
  // key fold: dqs_h = dq~_h Wk_h^T, dq = 0.1 dqs;  q projection: dxq = dq Wq
  g1.run(f, BIG, L::SB, KD, P, [&](int h, int m, int n, float v) {
    v *= 0.1f;
    DQ[m * CH_S2 + h * D + n] = v;
    if (m0 + m < Q) dq[(m0 + m) * QD + h * D + n] = v;
  }, [&](Frag& g) { g2.load(g, wave, 0); });
  g2.run(f, DQ, CH_S2, 0, P, [&](int, int m, int n, float v) { DXQ[m * CH_S2 + n] = v; }, [](Frag&) {});
  for (int i = threadIdx.x; i < 16 * D; i += CH_T) {     // k_bwd_finish
    const int m = i / D, n = i % D;
    const int64_t q = m0 + m;
    if (q >= Q) continue;
    if (d_src) d_src[q * D + n] = dx1[q * 3 * D + QD + n] + DXQ[m * CH_S2 + n];
    const float dst = DXQ[m * CH_S2 + D + n];
    if (in.dt) {
      if (tgrad) tgrad[q * QD + D + n] += -orca_sinf(orca_time_arg(0.0f, in.time_w[n], in.time_b[n])) * dst;
    } else if (d_src_time) {
      d_src_time[q * D + n] = dst;
    }
  }
}

struct Ws {   // float offsets of the per-query rows inside the workspace
  int64_t xq, qs, qt, a, xbar, ocat, x1, h1, y, dy, gx, dh1, dx1, docat, dxbar, dqt, dq, dxq, tgrad, invalid, wsum, dsum, scaled,
      wpack, part, total;
};
constexpr int SPLITS = 16;      // split-K pieces of the weight-gradient GEMMs (all of a layer's in one launch)
Ws ws_layout(int d_e, int64_t Q, int K) {
  const int64_t kdim = QD + d_e;
  Ws w{};
  int64_t o = 0;
  auto take = [&](int64_t per) { int64_t r = o; o += (Q * per + 63) / 64 * 64; return r; };
  w.xq = take(QD); w.qs = take(QD); w.qt = take(2 * kdim); w.a = take(2 * K); w.xbar = take(2 * kdim);
  w.ocat = take(QD); w.x1 = take(3 * D); w.h1 = take(D); w.y = take(D); w.dy = take(D); w.gx = take(D);
  w.dh1 = take(D); w.dx1 = take(3 * D); w.docat = take(QD); w.dxbar = take(2 * kdim); w.dqt = take(2 * kdim);
  w.dq = take(QD); w.dxq = take(QD); w.tgrad = take(QD); w.invalid = take(1); w.wsum = take(2); w.dsum = take(2);
  w.scaled = take(QD);
  w.wpack = o;      // fragment-order copies of the weights for the fused chains
  if (d_e == 172) o += (int64_t)ChainW<QD + 172>::total * 256;
  w.part = o;
  // the eight weight-gradient problems with their bias columns and the three that are column sums only (each rounded up
  // to 64 floats), or a column sum's slices
  o += (int64_t)SPLITS * ((int64_t)D * (D + 1) + (int64_t)D * (3 * D + 1) + 2 * (int64_t)QD * (QD + 1) + 4 * (int64_t)D * (kdim + 1) + 2 * D + QD)
       + 64 * ORCA_GEMM_GROUP_MAX + 4 * 64 * 320;
  w.total = o;
  return w;
}

// the fused chains are built for d_e = 172 and read float4 rows of the workspace
bool chain_usable(int d_e, const float* workspace) {
  static const bool off = getenv("ORCA_BWD_CHAIN") && atoi(getenv("ORCA_BWD_CHAIN")) == 0;
  return !off && d_e == 172 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
}

}  // namespace

ORCA_API int64_t orca_tgat_backward_workspace_floats(int d_e, int64_t Q, int K) {
  if (d_e < 1 || d_e > ORCA_MAX_DE || Q < 0 || K < 1 || K > ORCA_MAX_K) return -1;
  return ws_layout(d_e, Q, K).total;
}

namespace {
int tgat_backward_impl(const orca_tgat_params* P, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                       const float* grad_out, const orca_tgat_params* G, float* d_src, float* d_nbr,
                       float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b,
                       float* workspace, const float* saved, int phases, void* stream_);
}
ORCA_API int orca_tgat_backward(const orca_tgat_params* P, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                                const float* grad_out, const orca_tgat_params* G, float* d_src, float* d_nbr,
                                float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b,
                                float* workspace, void* stream_) {
  return tgat_backward_impl(P, d_e, in, Q, K, grad_out, G, d_src, d_nbr, d_src_time, d_nbr_time, d_time_w, d_time_b, workspace,
                            nullptr, 3, stream_);
}
ORCA_API int orca_tgat_backward_saved(const orca_tgat_params* P, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                                      const float* grad_out, const orca_tgat_params* G, float* d_src, float* d_nbr,
                                      float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b,
                                      float* workspace, const float* saved, void* stream_) {
  ORCA_REQUIRE(saved, "orca_tgat_backward_saved: null buffer");
  return tgat_backward_impl(P, d_e, in, Q, K, grad_out, G, d_src, d_nbr, d_src_time, d_nbr_time, d_time_w, d_time_b, workspace,
                            saved, 3, stream_);
}
ORCA_API int orca_tgat_backward_phase(const orca_tgat_params* P, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                                      const float* grad_out, const orca_tgat_params* G, float* d_src, float* d_nbr,
                                      float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b,
                                      float* workspace, const float* saved, int phases, void* stream_) {
  ORCA_REQUIRE(phases == 1 || phases == 2 || phases == 3, "orca_tgat_backward_phase: phases=%d (1 data gradients, 2 parameter "
               "gradients, 3 both)", phases);
  return tgat_backward_impl(P, d_e, in, Q, K, grad_out, G, d_src, d_nbr, d_src_time, d_nbr_time, d_time_w, d_time_b, workspace,
                            saved, phases, stream_);
}
namespace {
int tgat_backward_impl(const orca_tgat_params* P, int d_e, const orca_tgat_inputs* in, int64_t Q, int K,
                       const float* grad_out, const orca_tgat_params* G, float* d_src, float* d_nbr,
                       float* d_src_time, float* d_nbr_time, float* d_time_w, float* d_time_b,
                       float* workspace, const float* saved, int phases, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(Q >= 0, "orca_tgat_backward: negative Q");
  if (Q == 0) return ORCA_OK;
  ORCA_REQUIRE(P && G && in && grad_out && workspace, "orca_tgat_backward: null pointer");
  ORCA_REQUIRE(d_e >= 1 && d_e <= ORCA_MAX_DE, "orca_tgat_backward: d_e=%d outside [1,%d]", d_e, ORCA_MAX_DE);
  ORCA_REQUIRE(K >= 1 && K <= ORCA_MAX_K, "orca_tgat_backward: K=%d outside [1,%d]", K, ORCA_MAX_K);
  ORCA_REQUIRE(Q < (1LL << 31) / 1024, "orca_tgat_backward: Q too large for one call");
  (void)hipGetLastError();      // (a status left behind by an earlier runtime call is not this call's)
  ORCA_REQUIRE(P->q_w && P->k_w && P->v_w && P->in_b && P->o_w && P->o_b && P->fc1_w && P->fc1_b && P->fc2_w &&
               P->fc2_b && P->ln_w && P->ln_b, "orca_tgat_backward: null parameter");
  ORCA_REQUIRE(G->q_w && G->k_w && G->v_w && G->in_b && G->o_w && G->o_b && G->fc1_w && G->fc1_b && G->fc2_w &&
               G->fc2_b && G->ln_w && G->ln_b, "orca_tgat_backward: null gradient buffer");
  ORCA_REQUIRE(in->src_table && in->nbr_table && in->edge_table, "orca_tgat_backward: null feature table");
  ORCA_REQUIRE(in->nbr_ids || in->mask, "orca_tgat_backward: need nbr_ids or mask");
  ORCA_REQUIRE(in->dt ? (in->time_w && in->time_b) : (in->src_time && in->nbr_time),
               "orca_tgat_backward: need (dt, time_w, time_b) or (src_time, nbr_time)");
  ORCA_REQUIRE(!in->src_flags || in->src_overlay, "orca_tgat_backward: src_flags without src_overlay");
  ORCA_REQUIRE(!in->nbr_flags || in->nbr_overlay, "orca_tgat_backward: nbr_flags without nbr_overlay");
  ORCA_REQUIRE(!in->dt || ((d_time_w != nullptr) == (d_time_b != nullptr)),
               "orca_tgat_backward: d_time_w and d_time_b go together");
  const int kdim = QD + d_e;
  const int M = (int)Q;
  const Ws w = ws_layout(d_e, Q, K);
  float* W = workspace;
  float *xq = W + w.xq, *qs = W + w.qs, *qt = W + w.qt, *a = W + w.a, *xbar = W + w.xbar, *ocat = W + w.ocat,
        *x1 = W + w.x1, *h1 = W + w.h1, *y = W + w.y, *dy = W + w.dy, *gx = W + w.gx, *dh1 = W + w.dh1,
        *dx1 = W + w.dx1, *docat = W + w.docat, *dxbar = W + w.dxbar, *dqt = W + w.dqt, *dq = W + w.dq,
        *dxq = W + w.dxq, *tgrad = W + w.tgrad, *invalid = W + w.invalid, *wsum = W + w.wsum, *dsum = W + w.dsum, *part = W + w.part, *wpack = W + w.wpack;
  const bool drop = in->dropout_p > 0.f;
  ORCA_REQUIRE(in->dropout_p >= 0.f && in->dropout_p < 1.f, "orca_tgat_backward: dropout_p=%g outside [0,1)", (double)in->dropout_p);
  if (saved) {     // what the forward kept (orca_tgat_forward_saving): neither q~ nor the attention core is recomputed
    const orca_saved_layout SL = orca_saved(d_e, Q, K);
    float* sv = const_cast<float*>(saved);      // (read only below)
    qt = sv + SL.qt; a = sv + SL.a; xbar = sv + SL.xbar; wsum = sv + SL.wsum; invalid = sv + SL.invalid;
  }
  const unsigned qb = (unsigned)ceil_div64(Q, 4);
  int rc = ORCA_OK;
  auto ew = [&](int64_t n) { return (unsigned)ceil_div64(n, 256); };

  float* scaled = W + w.scaled;
  auto G1 = [&](int M_, int N_, int K_, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                float* C, int64_t scm, int64_t scn, const float* bias, float alpha) {
    return OrcaGemm{M_, N_, K_, A, sam, sak, B, sbk, sbn, C, scm, scn, bias, alpha, 0, 0, nullptr, nullptr, 0, nullptr, 1};
  };
  const bool want_t = in->dt && d_time_w;
  if (phases & 1) {
    // ---- forward recompute, then the data gradients: a chain (fused per 16-row tile when the shapes are the built ones) --
    const bool chain = chain_usable(d_e, workspace);
    const unsigned tiles = (unsigned)ceil_div64(Q, 16);
    using CL = ChainLds<QD + 172>;
    if (chain) {
      static bool attr_set[64] = {};
      if (orca_first_use_on_device(attr_set)) {
        // (the kernels also hold a few hundred static floats: ask for what each launch uses, not for the whole LDS)
        ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_chain_a<QD + 172>, hipFuncAttributeMaxDynamicSharedMemorySize, CL::A_FLOATS * (int)sizeof(float)));
        ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_chain_b<QD + 172>, hipFuncAttributeMaxDynamicSharedMemorySize, CL::B_FLOATS * (int)sizeof(float)));
        ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_chain_c<QD + 172>, hipFuncAttributeMaxDynamicSharedMemorySize, CL::C_FLOATS * (int)sizeof(float)));
      }
      {
        using CW = ChainW<QD + 172>;
        const int KDc = QD + 172;
        ChainPack pk{};
        int i = 0;
        pk.j[i++] = chain_pack_job<CW::A1>(P->q_w, 1, QD, 0, QD, QD, true, CW::a1);
        pk.j[i++] = chain_pack_job<CW::A2>(P->k_w, KDc, 1, D * KDc, KDc, D, false, CW::a2);
        pk.j[i++] = chain_pack_job<CW::B1>(P->v_w, 1, KDc, D * KDc, D, KDc, true, CW::b1);
        pk.j[i++] = chain_pack_job<CW::B2>(P->o_w, 1, QD, 0, QD, QD, true, CW::b2);
        pk.j[i++] = chain_pack_job<CW::B3>(P->fc1_w, 1, 3 * D, 0, D, 3 * D, true, CW::b3);
        pk.j[i++] = chain_pack_job<CW::B4>(P->fc2_w, 1, D, 0, D, D, true, CW::b4);
        pk.j[i++] = chain_pack_job<CW::B5>(P->fc2_w, D, 1, 0, D, D, false, CW::b5);
        pk.j[i++] = chain_pack_job<CW::B6>(P->fc1_w, 3 * D, 1, 0, 3 * D, D, false, CW::b6);
        pk.j[i++] = chain_pack_job<CW::B7>(P->o_w, QD, 1, 0, QD, QD, false, CW::b7);
        pk.j[i++] = chain_pack_job<CW::B8>(P->v_w, KDc, 1, D * KDc, KDc, D, false, CW::b8);
        pk.j[i++] = chain_pack_job<CW::C1>(P->k_w, 1, KDc, D * KDc, D, KDc, true, CW::c1);
        pk.j[i++] = chain_pack_job<CW::C2>(P->q_w, QD, 1, 0, QD, QD, false, CW::c2);
        static_assert(CHAIN_PACK_JOBS == 12, "one job per stage");
        pk.total = CW::total;
        k_chain_pack<<<(unsigned)ceil_div64((int64_t)CW::total * 64, 256), 256, 0, s>>>(pk, wpack);
      }
      k_chain_a<QD + 172><<<tiles, CH_T, CL::A_FLOATS * sizeof(float), s>>>(*in, Q, wpack, P->in_b, xq, x1, qs, saved ? nullptr : qt);
    } else {
      k_bwd_prep<<<qb, 256, 0, s>>>(*in, Q, xq, x1);
      // qs = 0.1 (xq Wq^T + bq)                                                    (scaling: head_dim^-0.5)
      rc |= sgemm(s, M, QD, QD, xq, QD, 1, P->q_w, 1, QD, qs, QD, 1, P->in_b, 0.1f, false, false);
      if (!saved) {   // q~_h = qs_h Wk_h   [Q,100] x [100,kdim], both heads in one launch
        OrcaGemm g[2];
        for (int h = 0; h < 2; ++h)
          g[h] = G1(M, kdim, D, qs + h * D, QD, 1, P->k_w + (int64_t)h * D * kdim, kdim, 1, qt + h * kdim, 2 * kdim, 1, nullptr, 1.f);
        rc |= orca_sgemm_group(s, g, 2, nullptr);
      }
    }
    if (saved) {
    } else if (K <= 10) k_bwd_attn_fwd<10><<<qb, 256, 0, s>>>(*in, Q, K, d_e, qt, a, xbar, invalid, wsum);
    else k_bwd_attn_fwd<ORCA_MAX_K><<<qb, 256, 0, s>>>(*in, Q, K, d_e, qt, a, xbar, invalid, wsum);
    if (chain) {
      const ChainB c{wpack, P->in_b + 2 * QD, P->o_b, P->fc1_b, P->fc2_b, P->ln_w,
                     grad_out, xbar, invalid, wsum,
                     x1, ocat, h1, dy, gx, dh1, dx1, docat, dxbar, scaled, dsum, drop ? 1 : 0};
      k_chain_b<QD + 172><<<tiles, CH_T, CL::B_FLOATS * sizeof(float), s>>>(c, Q);
    } else {
      {   // ocat_h = xbar_h Wv_h^T + bv_h  (dropout: + (sum of the head's kept weights) bv_h)
        OrcaGemm g[2];
        for (int h = 0; h < 2; ++h)
          g[h] = G1(M, D, kdim, xbar + h * kdim, 2 * kdim, 1, P->v_w + (int64_t)h * D * kdim, 1, kdim, ocat + h * D, QD, 1,
                    drop ? nullptr : P->in_b + 2 * QD + h * D, 1.f);
        rc |= orca_sgemm_group(s, g, 2, nullptr);
      }
      if (drop) k_add_scaled_bias<<<ew(Q * QD), 256, 0, s>>>(ocat, Q, wsum, P->in_b + 2 * QD);
      {   // o = ocat Wo^T + bo -> x1[:, 0:200], zero for invalid queries                 (temporal_attention.py:52)
        OrcaGemm g = G1(M, QD, QD, ocat, QD, 1, P->o_w, 1, QD, x1, 3 * D, 1, P->o_b, 1.f);
        g.rowzero = invalid;
        g.rowzero_n = QD;
        rc |= orca_sgemm_group(s, &g, 1, nullptr);
      }
      // h1 = relu(x1 W1^T + b1), y = h1 W2^T + b2                                     (utils/util.py:22-24)
      rc |= sgemm(s, M, D, 3 * D, x1, 3 * D, 1, P->fc1_w, 1, 3 * D, h1, D, 1, P->fc1_b, 1.f, false, true);
      rc |= sgemm(s, M, D, D, h1, D, 1, P->fc2_w, 1, D, y, D, 1, P->fc2_b, 1.f, false, false);
      k_bwd_ln<<<qb, 256, 0, s>>>(y, grad_out, P->ln_w, Q, dy, gx);
      {   // fc2: dz1 = (dy W2) where h1 > 0
        OrcaGemm g = G1(M, D, D, dy, D, 1, P->fc2_w, D, 1, dh1, D, 1, nullptr, 1.f);
        g.gate = h1;
        rc |= orca_sgemm_group(s, &g, 1, nullptr);
      }
      {   // fc1: dx1 = dz1 W1; do = dx1[:, 0:200] = 0 where the output was zeroed
        OrcaGemm g = G1(M, 3 * D, D, dh1, D, 1, P->fc1_w, 3 * D, 1, dx1, 3 * D, 1, nullptr, 1.f);
        g.rowzero = invalid;
        g.rowzero_n = QD;
        rc |= orca_sgemm_group(s, &g, 1, nullptr);
      }
      // out-proj: docat = do Wo
      rc |= sgemm(s, M, QD, QD, dx1, 3 * D, 1, P->o_w, QD, 1, docat, QD, 1, nullptr, 1.f, false, false);
      if (drop)       // docat scaled by the weight sums (the bv gradient's summand), dsum = docat_h . bv_h
        k_bias_scale_bwd<<<qb, 256, 0, s>>>(docat, Q, wsum, P->in_b + 2 * QD, scaled, dsum);
      {   // value map: dxbar_h = docat_h Wv_h
        OrcaGemm g[2];
        for (int h = 0; h < 2; ++h)
          g[h] = G1(M, kdim, D, docat + h * D, QD, 1, P->v_w + (int64_t)h * D * kdim, kdim, 1, dxbar + h * kdim, 2 * kdim, 1, nullptr, 1.f);
        rc |= orca_sgemm_group(s, g, 2, nullptr);
      }
    }
    if (K <= 10)
      k_bwd_attn_bwd<10><<<qb, 256, 0, s>>>(*in, Q, K, d_e, qt, a, dxbar, dqt, d_nbr, d_nbr_time, want_t ? tgrad : nullptr, dsum);
    else
      k_bwd_attn_bwd<ORCA_MAX_K><<<qb, 256, 0, s>>>(*in, Q, K, d_e, qt, a, dxbar, dqt, d_nbr, d_nbr_time, want_t ? tgrad : nullptr, dsum);
    if (chain) {
      k_chain_c<QD + 172><<<tiles, CH_T, CL::C_FLOATS * sizeof(float), s>>>(*in, Q, wpack, dqt, dx1, dq, d_src, d_src_time,
                                                                         want_t ? tgrad : nullptr);
    } else {
      {   // key fold: dqs_h = dq~_h Wk_h^T, dq = 0.1 dqs
        OrcaGemm g[2];
        for (int h = 0; h < 2; ++h)
          g[h] = G1(M, D, kdim, dqt + h * kdim, 2 * kdim, 1, P->k_w + (int64_t)h * D * kdim, 1, kdim, dq + h * D, QD, 1, nullptr, 0.1f);
        rc |= orca_sgemm_group(s, g, 2, nullptr);
      }
      // q projection: dxq = dq Wq
      rc |= sgemm(s, M, QD, QD, dq, QD, 1, P->q_w, QD, 1, dxq, QD, 1, nullptr, 1.f, false, false);
      k_bwd_finish<<<ew(Q * D), 256, 0, s>>>(*in, Q, dx1, dxq, d_src, d_src_time, want_t ? tgrad : nullptr);
    }
  }
  if (!(phases & 2)) {
    if (rc != ORCA_OK) return ORCA_EHIP;      // (orca_last_error keeps the failing callee's own message)
    if (hipGetLastError() != hipSuccess) {
      orca_set_error("orca_tgat_backward: kernel launch failed");
      return ORCA_EHIP;
    }
    return ORCA_OK;
  }
  // ---- backward: every weight gradient (a reduction over the queries) in ONE split-K launch, the bias gradients as a
  // column of ones:  dW2 += dy^T h1, dW1 += dz1^T x1, dWo += do^T ocat, dWv_h += docat_h^T xbar_h, dWk_h += qs_h^T dq~_h
  // (the bk gradient is zero: softmax is shift-invariant), dWq += dq^T xq;  LayerNorm's affine gradients -- column
  // sums of g * xhat and of g -- ride along as two problems that have only the ones column
  {
    OrcaGemm g[ORCA_GEMM_GROUP_MAX];
    auto GW = [&](int M_, int N_, const float* A, int64_t sak, const float* B, int64_t sbk, float* C, float* bias_grad) {
      OrcaGemm r = G1(M_, N_, M, A, 1, sak, B, sbk, 1, C, N_, 1, nullptr, 1.f);
      r.accumulate = 1;
      r.ones_out = bias_grad;
      r.splits = SPLITS;
      return r;
    };
    g[0] = GW(D, D, dy, D, h1, D, G->fc2_w, G->fc2_b);
    g[1] = GW(D, 3 * D, dh1, D, x1, 3 * D, G->fc1_w, G->fc1_b);
    g[2] = GW(QD, QD, dx1, 3 * D, ocat, QD, G->o_w, G->o_b);
    for (int h = 0; h < 2; ++h) {
      g[3 + h] = GW(D, kdim, docat + h * D, QD, xbar + h * kdim, 2 * kdim, G->v_w + (int64_t)h * D * kdim,
                    drop ? nullptr : G->in_b + 2 * QD + h * D);
      g[5 + h] = GW(D, kdim, qs + h * D, QD, dqt + h * kdim, 2 * kdim, G->k_w + (int64_t)h * D * kdim, nullptr);
    }
    g[7] = GW(QD, QD, dq, QD, xq, QD, G->q_w, G->in_b);
    g[8] = GW(D, 0, gx, D, gx, 0, nullptr, G->ln_w);
    g[9] = GW(D, 0, grad_out, D, grad_out, 0, nullptr, G->ln_b);
    int ng = 10;
    if (drop) g[ng++] = GW(QD, 0, scaled, QD, scaled, 0, nullptr, G->in_b + 2 * QD);   // the value bias under dropout: column sums too
    rc |= orca_sgemm_group(s, g, ng, part);
  }
  if (want_t) {   // what is left of the bias-like gradients: the time encoder's w and b
    const float* X[3];
    int64_t ld[3];
    int nc[3], nj = 0;
    float* out[3];
    {
      X[nj] = tgrad; ld[nj] = QD; nc[nj] = D; out[nj++] = d_time_w;
      X[nj] = tgrad + D; ld[nj] = QD; nc[nj] = D; out[nj++] = d_time_b;
    }
    rc |= orca_colsum_group(s, nj, X, ld, nc, out, Q, 1.f, true, part);
  }
  if (rc != ORCA_OK) return ORCA_EHIP;      // (orca_last_error keeps the failing callee's own message)
  if (hipGetLastError() != hipSuccess) {
    orca_set_error("orca_tgat_backward: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}
}  // namespace
