// GRU memory updater (modules/memory_updater.py:25-86 of the reference: nn.GRUCell(msg_dim -> 100)
// applied to the nodes that hold a pending raw message) as one MFMA kernel with a fused gate epilogue.
//
// A workgroup owns TR = 16*MT rows (nodes).  Their [message | memory] rows are gathered once into
// LDS; wave t (7 waves) owns hidden units 16t..16t+15 and accumulates, for each of its MT row tiles,
// the four pre-activations r, z, i_n, h_n of those units on v_mfma_f32_16x16x4_f32 -- so the gate
// math (sigmoid, tanh, blend with the old memory) happens in the accumulator registers and the new
// memory row is written straight to its table: in place (persist) or into the overlay table that
// replaces the reference's per-batch full-table clone (memory_updater.py:72,76).
#include "common.h"

namespace {

constexpr int D = ORCA_D;
constexpr int NWG = 7;            // waves per workgroup = ceil(100 / 16) hidden-unit tiles

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

constexpr int KS = 2;             // the reduction (message | memory columns) is split over KS wave groups
constexpr int GP = 2;             // k-blocks of weights fetched ahead (double buffered in registers)

// Workgroup = 16*MT rows x (KS*7) waves.  Wave (ks, t) accumulates, for hidden units 16t..16t+15 and its
// share of the k-blocks, the four pre-activations r, z, i_n, h_n; group 1 hands its partial sums to
// group 0 through LDS, which applies the gates.  Small MT = many short dependent MFMA chains in flight
// (the row counts here are a few thousand: latency, not throughput, is what has to be engineered).
template <int MT>
__global__ __launch_bounds__(KS * NWG * 64) void k_gru_update(const float* __restrict__ blob, int msg_dim,
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
  for (int m = wave; m < TRows; m += KS * NWG) {
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
  const int s_lo = ks * ((n_steps + KS - 1) / KS);
  const int s_hi = min(n_steps, s_lo + (n_steps + KS - 1) / KS);
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

}  // namespace

ORCA_API int64_t orca_gru_blob_floats(int msg_dim) {
  if (msg_dim < 1 || msg_dim > 3 * D + ORCA_MAX_DE) return -1;
  return gru_layout(msg_dim).total;
}

ORCA_API int orca_gru_pack(int msg_dim, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                           float* blob, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  ORCA_REQUIRE(msg_dim >= 1 && msg_dim <= 3 * D + ORCA_MAX_DE, "orca_gru_pack: msg_dim=%d unsupported", msg_dim);
  ORCA_REQUIRE(w_ih && w_hh && b_ih && b_hh && blob, "orca_gru_pack: null pointer");
  GBlob L = gru_layout(msg_dim);
  ORCA_HIP_TRY(hipMemsetAsync(blob, 0, (size_t)L.total * sizeof(float), s));
  int rc = ORCA_OK;
  for (int g = 0; g < 3; ++g) {
    rc |= orca_pack_matrix(blob + L.pw_x + (int64_t)g * NWG * L.KBx * 256, NWG, L.KBx, 0, D, msg_dim,
                           w_ih + (int64_t)g * D * msg_dim, msg_dim, 1, s);
    rc |= orca_pack_matrix(blob + L.pw_h + (int64_t)g * NWG * L.KBh * 256, NWG, L.KBh, 0, D, D,
                           w_hh + (int64_t)g * D * D, D, 1, s);
  }
  k_gru_bias<<<1, 128, 0, s>>>(blob + L.bias, b_ih, b_hh);
  if (rc != ORCA_OK || hipGetLastError() != hipSuccess) {
    orca_set_error("orca_gru_pack: kernel launch failed");
    return ORCA_EHIP;
  }
  return ORCA_OK;
}

template <int MT>
static int gru_launch(const float* blob, int msg_dim, const int32_t* ids, int64_t n, const int32_t* n_dev, const float* x,
                      const float* h, float* out, float* last_update, const float* msg_ts, hipStream_t stream) {
  GBlob L = gru_layout(msg_dim);
  const int rows = 16 * MT;
  size_t lds_bytes = (size_t)rows * L.AS * sizeof(float) + (size_t)NWG * 4 * MT * 64 * sizeof(f32x4);
  static bool attr_set = false;
  if (!attr_set) {
    ORCA_HIP_TRY(hipFuncSetAttribute((const void*)k_gru_update<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    attr_set = true;
  }
  ORCA_REQUIRE(lds_bytes <= 160 * 1024 - 256, "orca_gru_update: LDS budget exceeded (%zu B)", lds_bytes);
  k_gru_update<MT><<<(unsigned)ceil_div64(n, rows), KS * NWG * 64, lds_bytes, stream>>>(blob, msg_dim, ids, n, n_dev, x, h,
                                                                                    out, last_update, msg_ts);
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
  // 16-row tiles keep every CU busy with short chains while the row count is small; above ~1.5 waves of
  // workgroups the 32-row tile halves the weight traffic from L2
  if (n_dev || n <= 6144) return gru_launch<1>(blob, msg_dim, ids, n, n_dev, x, h, out, last_update, msg_ts, (hipStream_t)stream);
  return gru_launch<2>(blob, msg_dim, ids, n, n_dev, x, h, out, last_update, msg_ts, (hipStream_t)stream);
}
