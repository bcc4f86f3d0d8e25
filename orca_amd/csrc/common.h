// Internal helpers shared by the HIP translation units of liborca_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/orca_hip.h"

#define ORCA_API extern "C" __attribute__((visibility("default")))

void orca_set_error(const char* fmt, ...);

#define ORCA_HIP_TRY(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      orca_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ORCA_EHIP;                                                                 \
    }                                                                                   \
  } while (0)

#define ORCA_REQUIRE(cond, ...)                                                         \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      orca_set_error(__VA_ARGS__);                                                      \
      return ORCA_EINVAL;                                                               \
    }                                                                                   \
  } while (0)

#define ORCA_LAUNCH_CHECK() ORCA_HIP_TRY(hipGetLastError())

// fence index of a CSR's timestamps (sampler.h, csr.hip): lvl[k] == NULL from the first unused level on
constexpr int ORCA_FENCE_B0 = 32, ORCA_FENCE_LEVELS = 6;
struct orca_fences { const double* lvl[ORCA_FENCE_LEVELS]; };
int orca_csr_fences(const orca_csr_t* c, orca_fences* out);

// ---- library-internal entry points (hidden visibility; used by the native step driver) -------------
// First pass of orca_worklist_build fused into the sampler: every sampled neighbour / root whose flag is set
// stores slot[id] = epoch + position, positions numbered [neighbours (Q*K) ; roots (Q)]; *count is zeroed.
struct orca_claim {
  const uint8_t* flags;
  int64_t* slot;
  int64_t epoch, n_nodes;
  int32_t* count;
};
int orca_sample_recent_claim(const orca_csr_t* c, const int64_t* nodes, const double* ts, int64_t Q, int K,
                             int32_t* out_nbr, int32_t* out_eidx, float* out_et, float* out_dt, int* err_flag,
                             const orca_claim* claim, void* stream);
// second pass alone (the claim pass ran elsewhere with the same epoch and position numbering)
int orca_worklist_append(const int32_t* ids32, int64_t n32, const int64_t* ids64, int64_t n64, const uint8_t* flags,
                         const int64_t* slot, int64_t epoch, int64_t n_nodes, int32_t* worklist, int32_t* count,
                         void* stream);
// memory[ids[p]] = rows[p], last_update[ids[p]] = msg_ts[ids[p]] for the positions whose id is pending (the
// row commit of update_memory, memory_updater.py:36-39, from rows gathered by position)
int orca_persist_rows(const int64_t* ids, int64_t n, const uint8_t* pending, const float* rows, float* memory,
                      float* last_update, const float* msg_ts, int64_t n_nodes, void* stream);
// orca_history_push / orca_message_store with the last-occurrence winners precomputed (winner[p] != 0)
int orca_history_push_winners(float* emb, const float* x, const int64_t* ids, int64_t n, const uint8_t* winner,
                              int64_t n_nodes, void* stream);
int orca_message_store_winners(const int64_t* src, const int64_t* dst, const int64_t* eidx, const double* t, int64_t bs,
                               const float* memory, const float* last_update, const float* edge_feats, int d_e,
                               const float* time_w, const float* time_b, const uint8_t* winner, int64_t n_nodes,
                               float* messages, float* msg_ts, uint8_t* pending, void* stream);

// orca_find_uncached without the order: out[0 .. *count) holds the same ids in no particular order (state.hip)
int orca_find_uncached_unordered(const int32_t* nbr, int64_t n_nbr, const int64_t* batch_nodes, int64_t n_batch,
                                 const uint8_t* cache_flag, int64_t* slot, int64_t epoch, int64_t n_nodes, uint32_t* bitmap,
                                 int64_t* out, int64_t cap, int32_t* count, void* stream);
// its two halves for a caller whose own kernel stamped the batch ids (slot[id] = epoch) and zeroed count[0..1]: the
// marking pass, and the one-workgroup finish (drop the minimum, clear the bitmap) that also lays the evaluated set behind
// the Q roots of nodes / nodes32 / qts: the listed ids, then id -1 / node 0 padding up to cap, at time *tmax
int orca_uncached_mark(const int32_t* nbr, int64_t n_nbr, const uint8_t* cache_flag, const int64_t* slot, int64_t epoch,
                       int64_t n_nodes, uint32_t* bitmap, int64_t* out, int64_t cap, int32_t* count, void* stream);
int orca_uncached_finish_nodes(int64_t* out, int64_t cap, int32_t* count, uint32_t* bitmap, int64_t n_nodes, int64_t Q,
                               const double* tmax, int64_t* nodes, int32_t* nodes32, double* qts, int32_t* err, void* stream);

// orca_tgat_forward with the step driver's extra row writes fused in (either half may be left null):
//  persist: a query q < persist_n whose source row was read from the overlay (src_flags[row] set) also commits it:
//           persist_memory[row] = that row, persist_last_update[row] = persist_msg_ts[row]  (memory_updater.py:25-39;
//           safe inside the kernel because a flagged row of the memory table is read by nobody: flagged reads go
//           to the overlay);
//  push   : the output row of query q also goes to push_table[push_ids[q]] when push_winner[q] (history.py:26-30).
//  store  : wave q < 2*store_bs also builds and stores the raw message of position q of [src;dst] when
//           store_winner[q] (tgn_model.py:172-191 + memory.py:25-28).  Only for a kernel that reads none of the
//           memory / message / pending tables (an inner-history layer), after the persist of the same batch.
struct orca_tgat_side {
  float* persist_memory;
  float* persist_last_update;
  const float* persist_msg_ts;
  int64_t persist_n;
  float* push_table;
  const int64_t* push_ids;
  const uint8_t* push_winner;
  int64_t store_bs;                 // 0 = no store
  const int64_t* store_nodes;       // [src;dst;..] ids of the batch
  const double* store_ts;           // [t;t;..]
  const int64_t* store_eidx;        // [bs]
  const uint8_t* store_winner;      // [2*bs]
  const float* store_memory;
  const float* store_last_update;
  float* store_messages;
  float* store_msg_ts;
  uint8_t* store_pending;
  int32_t* zero_word;               // or null: a device counter this launch resets to 0
  // or null (all five together): what the backward pass would otherwise recompute (orca_tgat_forward_saving) --
  // q~ [Q, 2, kdim], the softmax weights before dropout [Q, 2, K], the weighted key sums x-bar [Q, 2, kdim], each
  // head's sum of kept weights [Q, 2] and the all-padded marker [Q]
  float* save_qt;
  float* save_a;
  float* save_xbar;
  float* save_wsum;
  float* save_invalid;
  const int32_t* q_extra;           // or null: only the first q_base + *q_extra (<= Q) queries are evaluated -- a launch
  int64_t q_base;                   // sized for a bound whose actual length is a device-side count (Path C's uncached set)
  float* src_rows_out;              // [Q, ORCA_D] or null: the (overlay-resolved) source row of every query, i.e. the
                                    // root's memory after its pending message is applied -- what a multi-rank step
                                    // all-gathers so that no rank has to redo the GRU of another rank's positives
};
// float offsets of the pieces of a `saved` buffer (orca_tgat_forward_saving -> orca_tgat_backward_saved)
struct orca_saved_layout { int64_t qt, a, xbar, wsum, invalid, total; };
static inline orca_saved_layout orca_saved(int d_e, int64_t Q, int K) {
  const int64_t kdim = 2 * ORCA_D + d_e;
  orca_saved_layout L{};
  int64_t o = 0;
  auto take = [&](int64_t per) { const int64_t r = o; o += (Q * per + 63) / 64 * 64; return r; };
  L.qt = take(2 * kdim); L.a = take(2 * K); L.xbar = take(2 * kdim); L.wsum = take(2); L.invalid = take(1);
  L.total = o;
  return L;
}
int orca_tgat_forward_side(const float* blob, int d_e, const orca_tgat_inputs* in, int64_t Q, int K, float* out,
                           float* out_weights, const orca_tgat_side* side, void* stream);

// Measurement hook of the step driver: when a (start, stop) event pair is armed on this thread, the next
// ORCA_LAUNCH_TIMED launch attaches it to its own dispatch (hipExtLaunchKernelGGL: the events take the kernel's
// begin / end timestamps, no marker packets are added to the stream) and disarms it.
struct orca_launch_events { hipEvent_t start, stop; };
extern thread_local orca_launch_events orca_armed_events;
#define ORCA_LAUNCH_TIMED(kernel, grid, block, lds, stream, ...)                                      \
  do {                                                                                                \
    orca_launch_events _ev = orca_armed_events;                                                       \
    orca_armed_events = orca_launch_events{nullptr, nullptr};                                         \
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), (uint32_t)(lds), stream, _ev.start, _ev.stop, 0, __VA_ARGS__); \
  } while (0)

// true the first time it is called for the current device (function attributes are per device: a process that
// drives several GPUs has to set them on each)
static inline bool orca_first_use_on_device(bool (&seen)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  if (seen[dev]) return false;
  seen[dev] = true;
  return true;
}

__host__ __device__ static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- packed weight format -------------------------------------------------------------------
// A matrix W with N output rows and Kd reduction columns is stored as NT = ceil(N/16) n-tiles x
// KB = ceil(Kd/16) k-blocks; each (nt, kb) block is 64 lanes x float4 = 1 KiB, lane l holding
// W[nt*16 + (l&15)][kb*16 + 4*(l>>4) + s], s = 0..3 (zero outside the matrix).  One wave-wide
// 16-byte load therefore fetches the B operands of four v_mfma_f32_16x16x4_f32 steps, fully
// coalesced.  The A operand uses the same k permutation (lane l: A[l&15][kb*16 + 4*(l>>4) + s]).
__host__ __device__ constexpr int pw_tiles(int n) { return (n + 15) / 16; }
__host__ __device__ constexpr int64_t pw_floats(int N, int Kd) { return (int64_t)pw_tiles(N) * pw_tiles(Kd) * 256; }

// ---- lane exchanges without the LDS --------------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32: an LDS round trip (a hundred-odd cycles) per call, and a reduction is a chain of
// them.  The xor distances a reduction needs all exist as vector instructions on gfx950: quad permutes (1, 2), two masked
// row shifts (4), a row rotation (8), and the two row / half swaps new in this chip (16, 32).  The swaps exchange between TWO
// registers -- v_permlane16_swap: the odd rows of a with the even rows of b; v_permlane32_swap: the upper half of a with the
// lower half of b -- which is exactly a halving step: with a = (what the lanes whose bit is clear keep) and b = (what the
// others keep), a + b afterwards is `kept + partner's copy` in every lane, without a select.  (Inline asm: the builtins'
// second result is lost by this compiler -- both elements of the returned pair come out as the first.  The s_nop's stand
// in for the hazard checks the compiler cannot do inside an asm statement.)
__device__ __forceinline__ void lanes_swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lanes_swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <int CTRL>
__device__ __forceinline__ float lanes_dpp(float v) {     // a full permutation: every lane receives
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lanes_xor1(float v) { return lanes_dpp<0xB1>(v); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ float lanes_xor2(float v) { return lanes_dpp<0x4E>(v); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ float lanes_xor8(float v) { return lanes_dpp<0x128>(v); }   // row_ror:8
__device__ __forceinline__ float lanes_xor4(float v) {
  const int iv = __builtin_bit_cast(int, v);
  int x = __builtin_amdgcn_update_dpp(0, iv, 0x104, 0xF, 0x5, false);     // row_shl:4 into lanes 0-3, 8-11 of a row: lane i <- i + 4
  x = __builtin_amdgcn_update_dpp(x, iv, 0x114, 0xF, 0xA, false);         // row_shr:4 into lanes 4-7, 12-15:       lane i <- i - 4
  return __builtin_bit_cast(float, x);
}
// v + (the value of lane ^ 16) resp. (lane ^ 32), as the butterfly of a reduction takes them
__device__ __forceinline__ float lanes_add_xor16(float v) { float a = v, b = v; lanes_swap16(a, b); return a + b; }
__device__ __forceinline__ float lanes_add_xor32(float v) { float a = v, b = v; lanes_swap32(a, b); return a + b; }
__device__ __forceinline__ float lanes_max_xor16(float v) { float a = v, b = v; lanes_swap16(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float lanes_max_xor32(float v) { float a = v, b = v; lanes_swap32(a, b); return fmaxf(a, b); }

// the sum over the wave in every lane: the same butterfly, distance 32 first, as with __shfl_xor (same bits)
__device__ __forceinline__ float wave_sum(float v) {
  v = lanes_add_xor32(v);
  v = lanes_add_xor16(v);
  v += lanes_xor8(v);
  v += lanes_xor4(v);
  v += lanes_xor2(v);
  v += lanes_xor1(v);
  return v;
}
// cos(x) for the time encoder (model/time_encoding.py:27).  Arguments reach 2.6e6 (Δt in seconds times
// ω_0 = 1), where the libm cosf takes its slow Payne-Hanek path; here the range reduction x = n·π + r is done
// in fp64 (two-constant π, error < 1e-9 for |x| < 2^31·π) and cos(r), |r| <= π/2, is ONE even polynomial of
// degree 12 in fp32 (Chebyshev fit; with the fp32 Horner rounding its absolute error is < 9.1e-8), the sign
// coming from the parity of n.  17 instructions, 7 of them fp64: the attention kernel evaluates 22 of these per
// lane and query, which made its gather stage VALU-bound with the two-polynomial quadrant form.
__device__ __forceinline__ float orca_cosf(float x) {
  const double xd = (double)x;
  const double n = __builtin_rint(xd * 0.31830988618379069122);
  double r = __builtin_fma(-n, 3.14159265358979311600e+00, xd);
  r = __builtin_fma(-n, 1.22464679914735320717e-16, r);
  const unsigned sign = (unsigned)(int)n << 31;
  const float rf = (float)r;
  const float u = rf * rf;
  float p = __builtin_fmaf(u, 1.9906489790e-09f, -2.7524603752e-07f);
  p = __builtin_fmaf(u, p, 2.4801038307e-05f);
  p = __builtin_fmaf(u, p, -1.3888884569e-03f);
  p = __builtin_fmaf(u, p, 4.1666667908e-02f);
  p = __builtin_fmaf(u, p, -0.5f);
  p = __builtin_fmaf(u, p, 1.0f);
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p) ^ sign);
}
// sin(x) with the same fp64 reduction x = n·π + r (the backward pass of the time encoder: d cos = -sin); sin(r),
// |r| <= π/2, is the odd Taylor polynomial through r^13 in fp32 (truncation < 7e-10, rounding ~1e-7 absolute).
__device__ __forceinline__ float orca_sinf(float x) {
  const double xd = (double)x;
  const double n = __builtin_rint(xd * 0.31830988618379069122);
  double r = __builtin_fma(-n, 3.14159265358979311600e+00, xd);
  r = __builtin_fma(-n, 1.22464679914735320717e-16, r);
  const unsigned sign = (unsigned)(int)n << 31;
  const float rf = (float)r;
  const float u = rf * rf;
  float p = __builtin_fmaf(u, 1.6059043837e-10f, -2.5052108385e-08f);
  p = __builtin_fmaf(u, p, 2.7557319224e-06f);
  p = __builtin_fmaf(u, p, -1.9841269841e-04f);
  p = __builtin_fmaf(u, p, 8.3333333333e-03f);
  p = __builtin_fmaf(u, p, -1.6666666667e-01f);
  p = __builtin_fmaf(u * rf, p, rf);
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p) ^ sign);
}
// Attention-weight dropout multiplier of (query, head, slot): counter RNG (splitmix64 finaliser), 24 uniform bits
__host__ __device__ static inline uint64_t orca_mix64_(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ static inline float orca_dropout_mult(uint64_t seed, int64_t q, int h, int j, float p, float scale) {
  const uint64_t r = orca_mix64_(seed ^ orca_mix64_((uint64_t)q * 64u + (uint64_t)(h * 32 + j)));
  return (float)(r >> 40) * (1.0f / 16777216.0f) >= p ? scale : 0.0f;
}
// t·ω + φ of nn.Linear(1, D) (time_encoding.py:26) with TWO roundings: the product, then the bias add.  With
// φ = 0 (the reference's initial value and every fixture it holds) the rounding order cannot matter; with a
// trained φ != 0 torch's own CPU linear rounds once or twice depending on the MKL code path of the host (measured:
// twice on the GPU box's host, once in the build container), which moves single cos values by O(0.1) at
// t ~ 1e6.  Two roundings is also what a GEMM with a bias epilogue does; contraction is switched off so that the
// choice does not hang on the compiler's default.
__device__ __forceinline__ float orca_time_arg(float t, float w, float b) {
#pragma clang fp contract(off)
  const float p = t * w;
  return p + b;
}

// Raw message of one interaction side, built and stored by one wavefront (model/tgn_model.py:172-191 +
// modules/memory.py:25-28): messages[id] = [memory[id] | memory[partner] | edge_feats[eidx] | cos(w (t - last_update[id]) + b)],
// msg_ts[id] = t, pending[id] = 1.
__device__ __forceinline__ void message_store_wave(int64_t id, int64_t partner, int64_t eidx, double t,
                                                   const float* __restrict__ memory,
                                                   const float* __restrict__ last_update,
                                                   const float* __restrict__ edge_feats, int d_e,
                                                   const float* __restrict__ tw, const float* __restrict__ tb,
                                                   float* __restrict__ messages, float* __restrict__ msg_ts,
                                                   uint8_t* __restrict__ pending, int lane) {
  const int msg_dim = 3 * ORCA_D + d_e;
  float* row = messages + id * msg_dim;
  const float tf = (float)t;                                      // tgn_model.py:181
  const float delta = tf - last_update[id];                       // :187
  const float* ms = memory + id * ORCA_D;
  const float* md = memory + partner * ORCA_D;
  const float* ef = edge_feats + eidx * d_e;
  for (int c = lane; c < ORCA_D; c += 64) {
    row[c] = ms[c];
    row[ORCA_D + c] = md[c];
    row[2 * ORCA_D + d_e + c] = orca_cosf(orca_time_arg(delta, tw[c], tb[c]));   // :188
  }
  for (int c = lane; c < d_e; c += 64) row[2 * ORCA_D + c] = ef[c];
  if (lane == 0) {
    msg_ts[id] = tf;
    pending[id] = 1;
  }
}

// Generic strided fp32 MFMA GEMM and column sums of the backward passes (tgat_bwd.hip):
// C(m,n) = alpha * (sum_k A(m,k) B(k,n) + bias[n]) [+ C(m,n)] [relu], X(i,j) = X[i*s0 + j*s1]; with part != NULL and
// splits > 1 the k range is cut into pieces whose partial sums are added in a fixed order (deterministic).
constexpr int ORCA_GEMM_SPLITS = 32;
int orca_sgemm(hipStream_t s, int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
               int64_t sbn, float* C, int64_t scm, int64_t scn, const float* bias, float alpha, bool accumulate, bool relu,
               float* part = nullptr, int splits = 1);
// Several independent problems in ONE launch (and one more for the split-K sums of those that ask for splits): the
// backward's GEMMs are short (10-15 us each), so what is launched together overlaps instead of queueing.
struct OrcaGemm {
  int M, N, K;
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t scm, scn;
  const float* bias;        // [N] or NULL
  float alpha;
  int accumulate, relu;
  const float* gate;        // strides of C, or NULL: the result is zeroed where gate(m,n) <= 0 (ReLU backward)
  const float* rowzero;     // [M] or NULL: rows with rowzero[m] != 0 are zeroed in the columns n < rowzero_n
  int rowzero_n;
  float* ones_out;          // [M] or NULL: B gets a virtual column N of ones, whose result alpha * sum_k A(m,k)
                            // (+ old value when accumulate) goes here -- the bias gradient of a weight-gradient GEMM
  int splits;               // > 1: split-K through `part` (no bias / relu / gate / rowzero then)
};
constexpr int ORCA_GEMM_GROUP_MAX = 12;
// floats of `part` a group needs
int64_t orca_sgemm_group_part_floats(const OrcaGemm* g, int n);
int orca_sgemm_group(hipStream_t s, const OrcaGemm* g, int n, float* part);
// out[n] (+)= alpha * sum_q X[q*ld + n], n < ncols
// (ncols <= 320; part: scratch of 64 * 320 floats)
int orca_colsum(hipStream_t s, const float* X, int64_t Q, int64_t ld, int ncols, float* out, float alpha, bool accumulate,
                float* part);
// up to four such sums (each of at most 320 columns) over the same Q rows in one launch pair; part: 4 * 64 * 320 floats
int orca_colsum_group(hipStream_t s, int n, const float* const* X, const int64_t* ld, const int* ncols, float* const* out,
                      int64_t Q, float alpha, bool accumulate, float* part);

// generic strided pack kernel launcher (tgat.hip)
int orca_pack_matrix(float* dst, int NT_total, int KB, int n0, int n_cnt, int Kd, const float* src,
                     int64_t sn, int64_t sk, hipStream_t stream, bool tail1 = false);
// Several pack jobs in ONE launch: a re-pack after every optimizer step is two dozen tiny kernels otherwise.
struct OrcaPackJob {
  int kind;                 // 0: matrix into MFMA-fragment order (orca_pack_matrix); 1: dst[i < n_pad] = i < n ? src[i] : 0;
                            // 2: *dst = 1
  float* dst;
  const float* src;
  int KB, n0, n_cnt, Kd, tail1;     // kind 0
  int64_t sn, sk;                   // kind 0
  int n, n_pad;                     // kind 1
};
constexpr int ORCA_PACK_GROUP_MAX = 32;
inline OrcaPackJob orca_pack_job(float* dst, int KB, int n0, int n_cnt, int Kd, const float* src, int64_t sn, int64_t sk,
                                 bool tail1 = false) {
  return OrcaPackJob{0, dst, src, KB, n0, n_cnt, Kd, tail1 ? 1 : 0, sn, sk, 0, 0};
}
inline OrcaPackJob orca_copy_pad_job(float* dst, const float* src, int n, int n_pad) {
  return OrcaPackJob{1, dst, src, 0, 0, 0, 0, 0, 0, 0, n, n_pad};
}
inline OrcaPackJob orca_flag_job(float* dst) { return OrcaPackJob{2, dst, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
int orca_pack_group(const OrcaPackJob* jobs, int n, hipStream_t stream);
