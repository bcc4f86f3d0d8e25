// Split-reduction GRU workgroup (device code shared by gru.hip and the step driver's first kernel in engine.hip).
#pragma once
#include <type_traits>
#include <utility>
#include "common.h"

constexpr int GRU_NWG = 7;        // hidden-unit tiles = ceil(100 / 16)
__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int GRU_KW = 4;             // waves per workgroup of the split kernel
constexpr int GRU_SPLIT_TILES = 384;  // row tiles the grid covers in one pass (workgroups loop beyond that)

// One wave's share of the reduction for MT row tiles: NSTEP k-blocks of rows of WIDTH columns starting at column 0
// of arow[m] (already offset by the lane's 4*(l>>4)), against the packed gate matrices at bgate[0..2] (already
// offset by the lane); the MT tiles share every weight fragment.  Fully unrolled and branch-free; only the LAST
// step can reach past the row (WIDTH_LEFT = columns of the row that remain at the first k-block): its load is
// clamped and its fragment zeroed when consumed.
// ODD: the row width is not a multiple of 4 floats (the zero-feature datasets' 301-wide message, train.py:139-142): rows
// then start at any 4-byte boundary -- the 16-byte fragment loads say so (f32x4u) -- and the row's last fragment is loaded
// element by element, each column clamped, so that nothing is read past the row's end.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NSTEP, int GS, int MT, bool ODD, typename WidthLeft>
__device__ __forceinline__ void gru_chain(const float* const (&arow)[MT], const f32x4* const (&bgate)[3], int kq,
                                          WidthLeft width_left, f32x4 (&acc)[MT][3]) {
  static_assert(GS <= NSTEP, "ring deeper than the chain");
  f32x4 bq[GS][3], aq[GS][MT];
  auto last_ok = [&]() { return (NSTEP - 1) * 16 + kq < width_left; };
  auto load_a = [&](const float* p) -> f32x4 {
    if constexpr (ODD) return *reinterpret_cast<const f32x4u*>(p);
    else return *reinterpret_cast<const f32x4*>(p);
  };
  auto fill = [&](auto J) {
    constexpr int j = decltype(J)::value;
    if constexpr (j < NSTEP) {
#pragma unroll
      for (int k = 0; k < 3; ++k) bq[j % GS][k] = bgate[k][j * 64];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if constexpr (j < NSTEP - 1) {
          aq[j % GS][m] = load_a(arow[m] + j * 16);
        } else if constexpr (ODD) {
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = j * 16 + kq + k;          // column counted from this wave's first one
            v[k] = arow[m][c < width_left ? j * 16 + k : 0];
          }
          aq[j % GS][m] = v;
        } else {
          aq[j % GS][m] = load_a(arow[m] + (last_ok() ? j * 16 : 0));
        }
      }
    }
  };
  [&]<int... I>(std::integer_sequence<int, I...>) { (fill(std::integral_constant<int, I>{}), ...); }
  (std::make_integer_sequence<int, GS>{});
  __builtin_amdgcn_sched_barrier(0);
  auto step = [&](auto J) {
    constexpr int j = decltype(J)::value;
    f32x4 a[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      a[m] = aq[j % GS][m];
      if constexpr (j == NSTEP - 1) {
        if constexpr (ODD) {
          const int c = j * 16 + kq;
          a[m] = f32x4{c < width_left ? a[m][0] : 0.f, c + 1 < width_left ? a[m][1] : 0.f, c + 2 < width_left ? a[m][2] : 0.f,
                       c + 3 < width_left ? a[m][3] : 0.f};
        } else {
          const bool ok = last_ok();
          a[m] = f32x4{ok ? a[m][0] : 0.f, ok ? a[m][1] : 0.f, ok ? a[m][2] : 0.f, ok ? a[m][3] : 0.f};
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 3; ++g)
          acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][k], bq[j % GS][g][k], acc[m][g], 0, 0, 0);
    // the slot is refilled only once its MFMAs are issued (fenced on both sides), so the allocator can reuse the
    // slot's registers: GS steps of operands are live, not GS + 2
    __builtin_amdgcn_sched_barrier(0);
    fill(std::integral_constant<int, j + GS>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, I>{}), ...); }
  (std::make_integer_sequence<int, NSTEP>{});
}

// Out-of-place GRU, reduction split over the 4 waves of a workgroup.  Workgroup = (16*MT rows) x (one 16-wide
// tile of hidden units); waves 0..2 each take a third of the message columns (10 k-blocks at MSG = 472), wave 3
// the 7 k-blocks of the memory columns, so the longest dependent MFMA chain is 10 steps instead of 37 and four
// times as many waves are in flight at the few-thousand-row counts of a batch.  Waves 1..3 hand their
// accumulators to wave 0 through LDS (waves 0..2 hold r, z, i_n partials, wave 3 holds r, z, h_n).  MT = 2 halves
// the weight traffic from L2 (the dominant stream: 111 KB per workgroup) and the number of workgroups, so that a
// batch's ~2000 rows are resident in one round.  The grid is sized for GRU_SPLIT_TILES row tiles and loops, so a
// launch bound far above the device-side row count does not pay for thousands of empty workgroups.
// `wg` of `n_wg` is this workgroup's place among those doing GRU work (a kernel of its own, or a role inside the
// step driver's first kernel); `part` is GRU_KW-1 x MT x 3 x 64 float4 of LDS.
// SUBS > 1: the block holds SUBS such 4-wave groups (wg = the FIRST group's index, groups wg .. wg+SUBS-1, each with
// its own `part`): the groups run their tile loops in lockstep -- the barriers are the block's -- so the trip count
// is that of the first group (the lowest tile index) and a group whose tile lies past the rows skips the work of an
// iteration but not its barriers.
template <int MSG, int GS, int MT, int SUBS = 1>
__device__ __forceinline__ void gru_split_workgroup(const float* __restrict__ blob, const int32_t* __restrict__ ids,
                                                    int64_t n_max, const int32_t* __restrict__ n_dev,
                                                    const float* __restrict__ x, const float* __restrict__ h,
                                                    float* __restrict__ out, float* last_update,
                                                    const float* __restrict__ msg_ts, int wg, int n_wg,
                                                    f32x4 (*part)[MT][3][64], long long* __restrict__ phase_clock,
                                                    float* __restrict__ save_gates = nullptr) {
  const int wg_first = wg;
  if constexpr (SUBS > 1) {
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    wg += sub;
    part += sub * (GRU_KW - 1);
  }
  const long long clk0 = phase_clock ? wall_clock64() : 0;
  static_assert(ORCA_D % 4 == 0, "a float4 fragment of a memory row must not straddle its end");
  // the message k-blocks are dealt to the message waves SPW at a time; the last of them takes what is left (SPW_L)
  constexpr int KBx = (MSG + 15) / 16, KBh = (ORCA_D + 15) / 16, SPW = (KBx + GRU_KW - 2) / (GRU_KW - 1), TR = 16 * MT;
  constexpr int SPW_L = KBx - (GRU_KW - 2) * SPW;
  constexpr bool ODD = MSG % 4 != 0;
  static_assert(SPW_L >= 1, "every message wave has at least one k-block");
  constexpr int D = ORCA_D;
  constexpr int64_t PW_X = 0, PW_H = 3 * (int64_t)GRU_NWG * KBx * 256, BIAS = PW_H + 3 * (int64_t)GRU_NWG * KBh * 256;
  const int t = wg % GRU_NWG;                           // hidden-unit tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & (GRU_KW - 1));   // everything derived from it stays in scalar registers
  const int kq = 4 * (lane >> 4);
  const bool hw = wave == GRU_KW - 1;                       // this wave walks the memory columns
  const int kb0 = hw ? 0 : wave * SPW;
  const int KBm = hw ? KBh : KBx;
  const f32x4* bgate0[3];
#pragma unroll
  for (int g = 0; g < 3; ++g)
    bgate0[g] = reinterpret_cast<const f32x4*>(blob) + (hw ? PW_H : PW_X) / 4 + ((int64_t)(g * GRU_NWG + t) * KBm + kb0) * 64;
  const float* bias = blob + BIAS;
  const int u = t * 16 + (lane & 15);
  const float br = bias[u], bz = bias[112 + u], bi = bias[224 + u], bh = bias[336 + u];   // padded to 112: in range
  int64_t n = n_max;
  if (n_dev) {
    const int64_t nd = *n_dev;
    n = nd < n ? nd : n;
  }
  long long clk1 = 0, clk2 = 0;
  const int64_t tstride = n_wg / GRU_NWG, last = n_max - 1;   // ids[] is valid memory up to n_max
  for (int64_t tile = wg / GRU_NWG, lead = wg_first / GRU_NWG; (SUBS > 1 ? lead : tile) * TR < n; tile += tstride, lead += tstride) {
    const bool act = SUBS == 1 || (tile * TR < n && wg < n_wg);
    const int64_t r0 = act ? tile * TR : 0;
    // the weights do not depend on the tile: the lane offset is made opaque per iteration, or the compiler hoists
    // every fragment out of the loop into registers and halves the waves a SIMD can hold
    int lane_off = lane;
    asm volatile("" : "+v"(lane_off));
    const f32x4* bgate[3] = {bgate0[0] + lane_off, bgate0[1] + lane_off, bgate0[2] + lane_off};
    int64_t id_a[MT];
    const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int64_t ra = r0 + m * 16 + (lane & 15);
      id_a[m] = ids[ra < last ? ra : last];             // rows beyond n read a valid row and are never stored
      arow[m] = hw ? h + id_a[m] * D + kq : x + id_a[m] * MSG + kb0 * 16 + kq;
    }
    // the old memory values the blend needs (rows of the epilogue's lane mapping): requested now by every wave --
    // branch-free -- so that wave 0 does not pay two dependent round trips after the barrier
    int64_t id_o[MT][4];
    float hp[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t ro = r0 + m * 16 + kq + r;
        id_o[m][r] = ids[ro < last ? ro : last];
      }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) hp[m][r] = h[id_o[m][r] * D + (u < D ? u : 0)];
    // ... and the message time stamps that become last_update (hidden tile 0 stores them): read where they are stored
    // they were a dependent round trip at the very end of one workgroup in seven -- the ones the launch then waits for
    float mts[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) mts[m] = (last_update ? msg_ts : blob)[last_update ? id_a[m] : 0];
    if (phase_clock) clk1 = wall_clock64();
    f32x4 acc[MT][3];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[m][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (act) {
      if (hw) gru_chain<KBh, (GS < KBh ? GS : KBh), MT, false>(arow, bgate, kq, std::integral_constant<int, D>{}, acc);
      else if (SPW_L != SPW && wave == GRU_KW - 2) gru_chain<SPW_L, (GS < SPW_L ? GS : SPW_L), MT, ODD>(arow, bgate, kq, MSG - kb0 * 16, acc);
      else gru_chain<SPW, GS, MT, ODD>(arow, bgate, kq, MSG - kb0 * 16, acc);
    }
    if (phase_clock) clk2 = wall_clock64();

    if (act && wave != 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 3; ++g) part[wave - 1][m][g][lane] = acc[m][g];
    }
    __syncthreads();
    if (act && wave == 0) {
      // fused gate epilogue (torch.nn.GRUCell, gate order r, z, n)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4 ar = acc[m][0], az = acc[m][1], a3 = acc[m][2];
        const f32x4 ah = part[GRU_KW - 2][m][2][lane];
#pragma unroll
        for (int w = 0; w < GRU_KW - 1; ++w) {
          ar += part[w][m][0][lane];
          az += part[w][m][1][lane];
          if (w < GRU_KW - 2) a3 += part[w][m][2][lane];
        }
        if (u < D) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t ro = r0 + m * 16 + kq + r;
            if (ro < n) {
              const float rr = gru_sigmoid(ar[r] + br);
              const float zz = gru_sigmoid(az[r] + bz);
              const float nn = tanhf(a3[r] + bi + rr * (ah[r] + bh));
              out[id_o[m][r] * D + u] = (1.0f - zz) * nn + zz * hp[m][r];
              if (save_gates) {     // what the backward would otherwise recompute (orca_gru_update_saving): [row][r, z, n, h_n][D]
                float* sg = save_gates + ro * 4 * D + u;
                sg[0] = rr; sg[D] = zz; sg[2 * D] = nn; sg[3 * D] = ah[r] + bh;
              }
            }
          }
        }
        if (t == 0 && last_update && lane < 16 && r0 + m * 16 + lane < n) last_update[id_a[m]] = mts[m];
      }
    }
    __syncthreads();   // `part` is reused by the next tile
  }
  if (phase_clock && lane == 0) {
    long long* pc = phase_clock + ((int64_t)wg * GRU_KW + wave) * 4;
    pc[0] = clk0; pc[1] = clk1; pc[2] = clk2; pc[3] = wall_clock64();
  }
}

