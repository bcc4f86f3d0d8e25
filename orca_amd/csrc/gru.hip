// GRU memory updater (modules/memory_updater.py:25-86 of the reference: nn.GRUCell(msg_dim -> 100)
// applied to the nodes that hold a pending raw message) on v_mfma_f32_16x16x4_f32 with a fused gate epilogue:
// the four pre-activations r, z, i_n, h_n of 16 rows x 16 hidden units accumulate in registers, so the gate
// math (sigmoid, tanh, blend with the old memory) never leaves them and the new memory row is written
// straight to its table: in place (persist) or into the overlay table that replaces the reference's
// per-batch full-table clone (memory_updater.py:72,76).  Three kernels, chosen by orca_gru_update:
//   k_gru_split  : out of place, device-counted or small row counts -- 4 waves split the reduction of one
//                  (16 rows) x (16 hidden units) tile, looping grid;
//   k_gru_stream : out of place, many rows -- one wave per tile, 37-step register ring;
//   k_gru_rows   : in-place-safe (a row is read and written by one workgroup), any message width.
#include <type_traits>
#include <utility>
#include "common.h"
#include "gru_split.h"

namespace {

constexpr int D = ORCA_D;
constexpr int NWG = GRU_NWG;      // waves per workgroup = ceil(100 / 16) hidden-unit tiles

struct GBlob {
  int XP, KBx, KBh, AS;
  int64_t pw_x, pw_h, bias, total;   // pw_x: 3 gates x PW(100, msg_dim); pw_h: 3 x PW(100,100); bias: 4 x 112
};

__host__ __device__ inline GBlob gru_layout(int msg_dim) {
  GBlob L;
  L.KBx = pw_tiles(msg_dim);
  L.XP = L.KBx * 16;
  L.KBh = pw_tiles(D);
  L.AS = L.XP + L.KBh * 16 + 4;
  int64_t o = 0;
  L.pw_x = o; o += 3 * (int64_t)NWG * L.KBx * 256;
  L.pw_h = o; o += 3 * (int64_t)NWG * L.KBh * 256;
  L.bias = o; o += 4 * 112;     // b_r = b_ir + b_hr, b_z = b_iz + b_hz, b_in, b_hn
  L.total = o;
  return L;
}

__global__ void k_gru_bias(float* dst, const float* b_ih, const float* b_hh) {
  int i = threadIdx.x;
  if (i >= 112) return;
  bool ok = i < D;
  dst[i] = ok ? b_ih[i] + b_hh[i] : 0.f;
  dst[112 + i] = ok ? b_ih[D + i] + b_hh[D + i] : 0.f;
  dst[224 + i] = ok ? b_ih[2 * D + i] : 0.f;
  dst[336 + i] = ok ? b_hh[2 * D + i] : 0.f;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int GP = 2;             // k-blocks of weights fetched ahead (double buffered in registers)
constexpr int KS2 = 2;            // the reduction (message | memory columns) is split over KS wave groups

// In-place-safe variant (out may alias h: a row is read and written by ONE workgroup).
// Workgroup = 16*MT rows x (2*7) waves.  Wave (ks, t) accumulates, for hidden units 16t..16t+15 and its
// share of the k-blocks, the four pre-activations r, z, i_n, h_n; group 1 hands its partial sums to
// group 0 through LDS, which applies the gates.  Small MT = many short dependent MFMA chains in flight
// (the row counts here are a few thousand: latency, not throughput, is what has to be engineered).
template <int MT>
__global__ __launch_bounds__(KS2 * NWG * 64) void k_gru_rows(const float* __restrict__ blob, int msg_dim,
                                                              const int32_t* __restrict__ ids, int64_t n_max,
                                                              const int32_t* __restrict__ n_dev,
                                                              const float* __restrict__ x, const float* h, float* out,
                                                              float* last_update, const float* __restrict__ msg_ts) {
  constexpr int TRows = 16 * MT;
  extern __shared__ __align__(16) float lds[];
  __shared__ int64_t s_id[TRows];
  const GBlob L = gru_layout(msg_dim);
  int64_t n = n_max;
  if (n_dev) {
    int64_t nd = *n_dev;
    n = nd < n ? nd : n;
  }
  const int64_t r0 = (int64_t)blockIdx.x * TRows;
  if (r0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ks = wave / NWG, t = wave - ks * NWG;
  if (tid < TRows) {
    int64_t r = r0 + tid;
    s_id[tid] = r < n ? (ids ? (int64_t)ids[r] : r) : -1;
  }
  __syncthreads();
  // gather [x(msg_dim) 0.. | h(100) 0..] rows into LDS
  for (int m = wave; m < TRows; m += KS2 * NWG) {
    const int64_t id = s_id[m];
    float* row = lds + m * L.AS;
    for (int c = lane; c < L.XP; c += 64) row[c] = (id >= 0 && c < msg_dim) ? x[id * msg_dim + c] : 0.0f;
    for (int c = lane; c < L.KBh * 16 + 4; c += 64) row[L.XP + c] = (id >= 0 && c < D) ? h[id * D + c] : 0.0f;
  }
  __syncthreads();

  const f32x4* blob4 = reinterpret_cast<const f32x4*>(blob);
  const f32x4* px = blob4 + L.pw_x / 4;
  const f32x4* ph = blob4 + L.pw_h / 4;
  f32x4 ar[MT], az[MT], ai[MT], ah[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) ar[m] = az[m] = ai[m] = ah[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* a_row = lds + (lane & 15) * L.AS + 4 * (lane >> 4);
  // steps 0..KBx-1 walk the message columns (gates r, z, i_n), steps KBx.. the memory columns (r, z, h_n);
  // the A row is laid out [x | h] so step s reads A columns 16s..16s+15 either way
  const int n_steps = L.KBx + L.KBh;
  const int s_lo = ks * ((n_steps + KS2 - 1) / KS2);
  const int s_hi = min(n_steps, s_lo + (n_steps + KS2 - 1) / KS2);
  auto bload = [&](int s, f32x4 (&b)[3]) {
    const bool xs = s < L.KBx;
    const f32x4* base = xs ? px : ph;
    const int kb = xs ? s : s - L.KBx, KBm = xs ? L.KBx : L.KBh;
#pragma unroll
    for (int g = 0; g < 3; ++g) b[g] = base[((int64_t)(g * NWG + t) * KBm + kb) * 64 + lane];
  };
  f32x4 cur[GP][3], nxt[GP][3];
#pragma unroll
  for (int g = 0; g < GP; ++g)
    if (s_lo + g < s_hi) bload(s_lo + g, cur[g]);
  for (int s0 = s_lo; s0 < s_hi; s0 += GP) {
#pragma unroll
    for (int g = 0; g < GP; ++g)
      if (s0 + GP + g < s_hi) bload(s0 + GP + g, nxt[g]);
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      const int s = s0 + g;
      if (s < s_hi) {
        f32x4 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(a_row + m * 16 * L.AS + s * 16);
        const bool xs = s < L.KBx;   // wave-uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            ar[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][k], cur[g][0][k], ar[m], 0, 0, 0);
            az[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][k], cur[g][1][k], az[m], 0, 0, 0);
            if (xs) ai[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][k], cur[g][2][k], ai[m], 0, 0, 0);
            else    ah[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][k], cur[g][2][k], ah[m], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < GP; ++g)
#pragma unroll
      for (int k = 0; k < 3; ++k) cur[g][k] = nxt[g][k];
  }
  // group 1 -> group 0 hand-over of the partial pre-activations
  f32x4* part = reinterpret_cast<f32x4*>(lds + TRows * L.AS);   // [NWG][4 gates][MT][64 lanes]
  if (ks == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      part[((t * 4 + 0) * MT + m) * 64 + lane] = ar[m];
      part[((t * 4 + 1) * MT + m) * 64 + lane] = az[m];
      part[((t * 4 + 2) * MT + m) * 64 + lane] = ai[m];
      part[((t * 4 + 3) * MT + m) * 64 + lane] = ah[m];
    }
  }
  __syncthreads();
  if (ks == 0) {
    // fused gate epilogue (torch.nn.GRUCell, gate order r, z, n)
    const float* bias = blob + L.bias;
    const int u = t * 16 + (lane & 15);
    if (u < D) {
      const float br = bias[u], bz = bias[112 + u], bi = bias[224 + u], bh = bias[336 + u];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 pr = part[((t * 4 + 0) * MT + m) * 64 + lane], pz = part[((t * 4 + 1) * MT + m) * 64 + lane];
        const f32x4 pi = part[((t * 4 + 2) * MT + m) * 64 + lane], pn = part[((t * 4 + 3) * MT + m) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m * 16 + 4 * (lane >> 4) + r;
          const int64_t id = s_id[row];
          if (id < 0) continue;
          float rr = sigmoidf_(ar[m][r] + pr[r] + br);
          float zz = sigmoidf_(az[m][r] + pz[r] + bz);
          float nn = tanhf(ai[m][r] + pi[r] + bi + rr * (ah[m][r] + pn[r] + bh));
          float hp = lds[row * L.AS + L.XP + u];
          out[id * D + u] = (1.0f - zz) * nn + zz * hp;
        }
      }
    }
    if (last_update && tid < TRows && s_id[tid] >= 0) last_update[s_id[tid]] = msg_ts[s_id[tid]];
  }
}


constexpr int RW = 1;             // 16-row sub-tiles (waves) per workgroup: single-wave workgroups spread evenly over the CUs
constexpr int GS = 5;             // steps of operands fetched ahead (ring of registers)

// Out-of-place GRU: workgroup = (64 rows) x (one 16-wide tile of hidden units); wave w owns rows 16w..16w+15
// and walks the whole reduction for them.  No LDS, no barrier: the A fragments (16 B per lane straight from
// the [message | memory] rows in HBM/L2) and the B fragments (packed weights, shared by the 4 waves through
// L1) stream GS steps ahead of the MFMAs in a register ring.  MSG (the message width) is a template
// parameter so that the 37-step chain is fully unrolled and BRANCH-FREE -- loads are clamped and their
// results fixed up when consumed, never skipped -- because the compiler's s_waitcnt placement collapses to
// vmcnt(0) at any control flow (and at any use right after a load), which serialises every step on a full
// memory round trip.  7 workgroups per row tile give every CU work at the few-thousand-row counts of a
// batch; 64-row tiles keep the weight traffic from L2 at 111 KB per workgroup.  A row is read by 7
// workgroups and written by each of them: `out` must not alias `h` (orca_gru_update routes the in-place
// case to k_gru_rows).
template <int MSG>
__global__ __launch_bounds__(RW * 64) void k_gru_stream(const float* __restrict__ blob,
                                                        const int32_t* __restrict__ ids, int64_t n_max,
                                                        const int32_t* __restrict__ n_dev,
                                                        const float* __restrict__ x, const float* __restrict__ h,
                                                        float* __restrict__ out, float* last_update,
                                                        const float* __restrict__ msg_ts,
                                                        long long* __restrict__ phase_clock) {
  const long long clk0 = phase_clock ? wall_clock64() : 0;
  static_assert(MSG % 4 == 0, "a float4 A fragment must not straddle the end of a row");
  constexpr int KBx = (MSG + 15) / 16, KBh = (D + 15) / 16, NS = KBx + KBh;
  constexpr int64_t PW_X = 0, PW_H = 3 * (int64_t)NWG * KBx * 256, BIAS = PW_H + 3 * (int64_t)NWG * KBh * 256;
  const int t = blockIdx.x % NWG;                       // hidden-unit tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = ((int64_t)(blockIdx.x / NWG) * RW + wave) * 16;
  const f32x4* blob4 = reinterpret_cast<const f32x4*>(blob);
  const f32x4* px = blob4 + PW_X / 4 + (int64_t)t * KBx * 64 + lane;      // gate g at + g*NWG*KBx*64
  const f32x4* ph = blob4 + PW_H / 4 + (int64_t)t * KBh * 64 + lane;
  const int kq = 4 * (lane >> 4);
  // ---- everything that depends on nothing: first GS steps of weights, biases, ids (valid memory up to n_max) --
  f32x4 bq[GS][3], aq[GS];
#pragma unroll
  for (int g = 0; g < GS; ++g) {
#pragma unroll
    for (int k = 0; k < 3; ++k) bq[g][k] = px[((int64_t)k * NWG * KBx + g) * 64];   // GS <= KBx: message steps
  }
  const float* bias = blob + BIAS;
  const int u = t * 16 + (lane & 15);
  const float br = bias[u], bz = bias[112 + u], bi = bias[224 + u], bh = bias[336 + u];   // padded to 112: in range
  const int64_t last = n_max - 1;
  const int64_t ra = r0 + (lane & 15);
  const int64_t id_a_raw = ids[ra < last ? ra : last];
  int64_t id_o_raw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t ro = r0 + kq + r;
    id_o_raw[r] = ids[ro < last ? ro : last];
  }
  int64_t n = n_max;
  if (n_dev) {
    const int64_t nd = *n_dev;
    n = nd < n ? nd : n;
  }
  if (r0 >= n) return;                                   // wave-uniform; there is no barrier in this kernel
  // lane l feeds MFMA row l&15 (node id_a); in the epilogue it owns rows 4*(l>>4)+r (nodes id_o[r])
  const bool live_a = ra < n;
  const float* xrow = x + (live_a ? id_a_raw : 0) * MSG;
  const float* hrow = h + (live_a ? id_a_raw : 0) * D;
  // A fragment of step s (compile-time): 4 consecutive floats of this lane's row.  The load is clamped into
  // the row and NOT predicated; columns beyond the row are zeroed when the fragment is consumed, and rows
  // beyond n need no zeroing at all: an MFMA output row depends on its own A row only and those rows are
  // never stored.
  auto aload = [&](auto S) -> f32x4 {
    constexpr int s = decltype(S)::value;
    constexpr bool xs = s < KBx;
    constexpr int kb = xs ? s : s - KBx, width = xs ? MSG : D;
    constexpr bool full = kb * 16 + 16 <= width;       // the whole k-block lies inside the row
    const int c = kb * 16 + kq;
    const float* row = xs ? xrow : hrow;
    return *reinterpret_cast<const f32x4*>(row + (full ? c : (c < width ? c : 0)));
  };
  auto afix = [&](auto S, f32x4 v) -> f32x4 {
    constexpr int s = decltype(S)::value;
    constexpr bool xs = s < KBx;
    constexpr int kb = xs ? s : s - KBx, width = xs ? MSG : D;
    if constexpr (kb * 16 + 16 <= width) return v;
    const bool ok = kb * 16 + kq < width;
    return f32x4{ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f};
  };
  auto bload = [&](auto S, f32x4 (&b)[3]) {
    constexpr int s = decltype(S)::value;
    constexpr bool xs = s < KBx;
    constexpr int kb = xs ? s : s - KBx, KBm = xs ? KBx : KBh;
    const f32x4* base = xs ? px : ph;
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = base[((int64_t)k * NWG * KBm + kb) * 64];
  };
  auto ring_fill = [&](auto S) {      // operands of step S into its ring slot
    constexpr int s = decltype(S)::value;
    if constexpr (s < NS) {
      aq[s % GS] = aload(S);
      if constexpr (s >= GS) bload(S, bq[s % GS]);     // the first GS steps' weights are already on their way
    }
  };
  float hp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = r0 + kq + r < n && u < D;
    hp[r] = h[(ok ? id_o_raw[r] : 0) * D + (u < D ? u : 0)];
  }
  // prologue: A of the first GS steps
  [&]<int... I>(std::integer_sequence<int, I...>) { (ring_fill(std::integral_constant<int, I>{}), ...); }
  (std::make_integer_sequence<int, GS>{});
  __builtin_amdgcn_sched_barrier(0);
  const long long clk1 = phase_clock ? wall_clock64() : 0;

  f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, ai = ar, ah = ar;
  auto step = [&](auto S) {
    constexpr int s = decltype(S)::value;
    const f32x4 a = afix(S, aq[s % GS]);
    const f32x4 b0 = bq[s % GS][0], b1 = bq[s % GS][1], b2 = bq[s % GS][2];
    ring_fill(std::integral_constant<int, s + GS>{});
    __builtin_amdgcn_sched_barrier(0);   // keep the refill ahead of this step's MFMAs
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b0[k], ar, 0, 0, 0);
      az = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b1[k], az, 0, 0, 0);
      if constexpr (s < KBx) ai = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b2[k], ai, 0, 0, 0);
      else                   ah = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b2[k], ah, 0, 0, 0);
    }
  };
  [&]<int... I>(std::integer_sequence<int, I...>) { (step(std::integral_constant<int, I>{}), ...); }
  (std::make_integer_sequence<int, NS>{});
  const long long clk2 = phase_clock ? wall_clock64() : 0;

  // fused gate epilogue (torch.nn.GRUCell, gate order r, z, n)
  if (u < D) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r0 + kq + r < n) {
        const float rr = sigmoidf_(ar[r] + br);
        const float zz = sigmoidf_(az[r] + bz);
        const float nn = tanhf(ai[r] + bi + rr * (ah[r] + bh));
        out[id_o_raw[r] * D + u] = (1.0f - zz) * nn + zz * hp[r];
      }
    }
  }
  if (t == 0 && last_update && lane < 16 && live_a) last_update[id_a_raw] = msg_ts[id_a_raw];
  if (phase_clock && lane == 0) {
    long long* pc = phase_clock + ((int64_t)blockIdx.x * RW + wave) * 4;
    pc[0] = clk0; pc[1] = clk1; pc[2] = clk2; pc[3] = wall_clock64();
  }
}


// The split-reduction workgroup as a kernel of its own (gru_split.h)
template <int MSG, int GS, int MT>
__global__ __launch_bounds__(GRU_KW * 64) void k_gru_split(const float* __restrict__ blob,
                                                        const int32_t* __restrict__ ids, int64_t n_max,
                                                        const int32_t* __restrict__ n_dev,
                                                        const float* __restrict__ x, const float* __restrict__ h,
                                                        float* __restrict__ out, float* last_update,
                                                        const float* __restrict__ msg_ts,
                                                        long long* __restrict__ phase_clock, float* __restrict__ save_gates) {
  __shared__ f32x4 part[GRU_KW - 1][MT][3][64];
  gru_split_workgroup<MSG, GS, MT>(blob, ids, n_max, n_dev, x, h, out, last_update, msg_ts, (int)blockIdx.x, (int)gridDim.x,
                                   part, phase_clock, save_gates);
}

}  // namespace

ORCA_API int64_t orca_gru_blob_floats(int msg_dim) {
  if (msg_dim < 1 || msg_dim > 3 * D + ORCA_MAX_DE) return -1;
  return gru_layout(msg_dim).total;
}

static int gru_pack_impl(int msg_dim, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                         float* blob, bool packed_before, void* stream_);
ORCA_API int orca_gru_pack(int msg_dim, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                           float* blob, void* stream_) {
  return gru_pack_impl(msg_dim, w_ih, w_hh, b_ih, b_hh, blob, false, stream_);
}
ORCA_API int orca_gru_repack(int msg_dim, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                             float* blob, void* stream_) {
  return gru_pack_impl(msg_dim, w_ih, w_hh, b_ih, b_hh, blob, true, stream_);
}
static int gru_pack_impl(int msg_dim, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                         float* blob, bool packed_before, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(msg_dim >= 1 && msg_dim <= 3 * D + ORCA_MAX_DE, "orca_gru_pack: msg_dim=%d unsupported", msg_dim);
  ORCA_REQUIRE(w_ih && w_hh && b_ih && b_hh && blob, "orca_gru_pack: null pointer");
  GBlob L = gru_layout(msg_dim);
  if (!packed_before) {       // the padding between the packed tiles is written once; a re-pack rewrites every tile
    ORCA_HIP_TRY(hipMemsetAsync(blob, 0, (size_t)L.total * sizeof(float), s));
  }
  int rc = ORCA_OK;
  OrcaPackJob jobs[6];
  for (int g = 0; g < 3; ++g) {
    jobs[2 * g] = orca_pack_job(blob + L.pw_x + (int64_t)g * NWG * L.KBx * 256, L.KBx, 0, D, msg_dim,
                                w_ih + (int64_t)g * D * msg_dim, msg_dim, 1);
    jobs[2 * g + 1] = orca_pack_job(blob + L.pw_h + (int64_t)g * NWG * L.KBh * 256, L.KBh, 0, D, D,
                                    w_hh + (int64_t)g * D * D, D, 1);
  }
  rc |= orca_pack_group(jobs, 6, s);
  k_gru_bias<<<1, 128, 0, s>>>(blob + L.bias, b_ih, b_hh);
  if (rc != ORCA_OK || hipGetLastError() != hipSuccess) {
    orca_set_error("orca_gru_pack: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}

static long long* g_gru_clock = nullptr;
// Profiling aid: buf = device int64 [n_workgroups * 4 waves * 4] (or NULL); the streaming GRU kernel stamps the
// 100 MHz counter at entry, after issuing its first loads, after the MFMA chain and at exit.
ORCA_API int orca_debug_gru_phase_clock(void* buf) {
  g_gru_clock = (long long*)buf;
  return ORCA_OK;
}

static int gru_launch_stream(const float* blob, const int32_t* ids, int64_t n, const int32_t* n_dev, const float* x,
                             const float* h, float* out, float* last_update, const float* msg_ts, hipStream_t stream) {
  ORCA_LAUNCH_TIMED((k_gru_stream<3 * D + 172>), (unsigned)(ceil_div64(n, 16 * RW) * NWG), RW * 64, 0, stream, blob, ids, n,
                    n_dev, x, h, out, last_update, msg_ts, g_gru_clock);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

template <int MSG = 3 * D + 172>
static int gru_launch_split(const float* blob, const int32_t* ids, int64_t n, const int32_t* n_dev, const float* x,
                            const float* h, float* out, float* last_update, const float* msg_ts, hipStream_t stream,
                            float* save_gates = nullptr) {
  // 16-row tiles (MT = 1).  32-row tiles (MT = 2: half the weight traffic, half the workgroups) measured 18.5 vs
  // 19.1 us on an isolated 2084-row work-list but 1 % slower inside the step, and slower at 2730 rows (24.0 vs 22.3)
  const int64_t tiles = ceil_div64(n, 16);
  const unsigned grid = (unsigned)((tiles < GRU_SPLIT_TILES ? tiles : GRU_SPLIT_TILES) * NWG);
  ORCA_LAUNCH_TIMED((k_gru_split<MSG, 3, 1>), grid, GRU_KW * 64, 0, stream, blob, ids, n, n_dev, x, h, out, last_update,
                    msg_ts, g_gru_clock, save_gates);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

static int gru_launch_rows(const float* blob, int msg_dim, const int32_t* ids, int64_t n, const int32_t* n_dev,
                           const float* x, const float* h, float* out, float* last_update, const float* msg_ts,
                           hipStream_t stream) {
  GBlob L = gru_layout(msg_dim);
  size_t lds_bytes = (size_t)16 * L.AS * sizeof(float) + (size_t)NWG * 4 * 64 * sizeof(f32x4);
  static bool attr_set[64] = {};
  if (orca_first_use_on_device(attr_set)) {
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_gru_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  }
  ORCA_LAUNCH_TIMED((k_gru_rows<1>), (unsigned)ceil_div64(n, 16), KS2 * NWG * 64, lds_bytes, stream, blob, msg_dim, ids, n,
                    n_dev, x, h, out, last_update, msg_ts);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}

ORCA_API int orca_gru_update(const float* blob, int msg_dim, const int32_t* ids, int64_t n, const int32_t* n_dev,
                             const float* x, const float* h, float* out, float* last_update, const float* msg_ts,
                             void* stream) {
  ORCA_REQUIRE(n >= 0, "orca_gru_update: n < 0");
  if (n == 0) return ORCA_OK;
  ORCA_REQUIRE(blob && x && h && out, "orca_gru_update: null pointer");
  ORCA_REQUIRE(msg_dim >= 1 && msg_dim <= 3 * D + ORCA_MAX_DE, "orca_gru_update: msg_dim=%d unsupported", msg_dim);
  ORCA_REQUIRE(!last_update || msg_ts, "orca_gru_update: last_update without msg_ts");
  // in place (out aliases h): a row must be read and written by one workgroup
  if (out == h) return gru_launch_rows(blob, msg_dim, ids, n, n_dev, x, h, out, last_update, msg_ts, (hipStream_t)stream);
  // out of place with an id list at the common message width (d_e = 172): (16-row tile) x (hidden tile)
  // workgroups streaming their operands; anything else (other widths, ids == NULL) takes the row kernel
  // (the zero-feature datasets' width, train.py:139-142 d_e = 1: the same split kernel; its grid loops over the row tiles)
  if (msg_dim == 3 * D + 1 && ids)
    return gru_launch_split<3 * D + 1>(blob, ids, n, n_dev, x, h, out, last_update, msg_ts, (hipStream_t)stream);
  if (msg_dim != 3 * D + 172 || !ids)
    return gru_launch_rows(blob, msg_dim, ids, n, n_dev, x, h, out, last_update, msg_ts, (hipStream_t)stream);
  // a device-counted work-list (launch bound far above the real row count) or a few thousand rows: latency is what
  // counts, the reduction is split over 4 waves; many rows: one wave per (row tile, hidden tile) streams the lot
  if (n_dev || n <= 4096) return gru_launch_split(blob, ids, n, n_dev, x, h, out, last_update, msg_ts, (hipStream_t)stream);
  return gru_launch_stream(blob, ids, n, n_dev, x, h, out, last_update, msg_ts, (hipStream_t)stream);
}

ORCA_API int orca_gru_update_saving(const float* blob, int msg_dim, const int32_t* ids, int64_t n, const int32_t* n_dev,
                                    const float* x, const float* h, float* out, float* gates, void* stream) {
  ORCA_REQUIRE(n >= 0, "orca_gru_update_saving: n < 0");
  if (n == 0) return ORCA_OK;
  ORCA_REQUIRE(blob && x && h && out && gates && ids, "orca_gru_update_saving: null pointer");
  ORCA_REQUIRE(msg_dim == 3 * D + 172 && out != h, "orca_gru_update_saving: built for msg_dim=%d, out of place (got %d%s)",
               3 * D + 172, msg_dim, out == h ? ", in place" : "");
  return gru_launch_split(blob, ids, n, n_dev, x, h, out, nullptr, nullptr, (hipStream_t)stream, gates);
}

// out_table[id] = src_table[id] (+ last_update[id] = msg_ts[id]) for the ids of a device-counted work-list:
// commits rows the GRU wrote to the overlay into the persistent memory (memory_updater.py:36-39)
namespace {
__global__ __launch_bounds__(256) void k_commit_rows(const int32_t* __restrict__ ids, int64_t n_max,
                                                     const int32_t* __restrict__ n_dev, const float* __restrict__ src,
                                                     float* __restrict__ dst, float* last_update,
                                                     const float* __restrict__ msg_ts, int64_t n_nodes) {
  int64_t n = n_max;
  if (n_dev) { int64_t nd = *n_dev; n = nd < n ? nd : n; }
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= n) return;
  const int64_t id = ids[r];
  if (id < 0 || id >= n_nodes) return;
  for (int c = lane; c < D; c += 64) dst[id * D + c] = src[id * D + c];
  if (last_update && lane == 0) last_update[id] = msg_ts[id];
}
}  // namespace

ORCA_API int orca_commit_rows(const int32_t* ids, int64_t n, const int32_t* n_dev, const float* src_table,
                              float* dst_table, float* last_update, const float* msg_ts, int64_t n_nodes, void* stream) {
  ORCA_REQUIRE(n >= 0 && n_nodes >= 0, "orca_commit_rows: n < 0");
  if (n == 0) return ORCA_OK;
  ORCA_REQUIRE(ids && src_table && dst_table && (!last_update || msg_ts), "orca_commit_rows: null pointer");
  k_commit_rows<<<(unsigned)ceil_div64(n, 4), 256, 0, (hipStream_t)stream>>>(ids, n, n_dev, src_table, dst_table,
                                                                            last_update, msg_ts, n_nodes);
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}
