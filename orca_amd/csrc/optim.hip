// Adam over a list of tensors in one launch (train.py:180,238: torch.optim.Adam(tgn.parameters(), lr) /
// optimizer.step()).  The model has ~35 parameter tensors of 100 .. 141,600 floats: torch's foreach implementation is
// seven launches and 0.3 ms of host time per step, a third of what is left of the training step.  Arithmetic in the
// order of torch's _multi_tensor_adam (no weight decay, no amsgrad):
//   m += (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <cmath>
#include "common.h"

namespace {
constexpr int ADAM_MAX = 48;       // tensors per launch (the TGN of the bench has 35: one launch)
struct AdamGroup {
  int n;
  int first[ADAM_MAX + 1];         // first block of each tensor (1024 elements per block)
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  int64_t size[ADAM_MAX];
};

__global__ __launch_bounds__(256) void k_adam(AdamGroup grp, float w1, float b2, float w2, float step_size, float bc2_sqrt,
                                              float eps) {
  int t = 0;
#pragma unroll
  for (int i = 1; i < ADAM_MAX; ++i) t += (i < grp.n && (int)blockIdx.x >= grp.first[i]) ? 1 : 0;
  t = __builtin_amdgcn_readfirstlane(t);
  float* __restrict__ p = grp.p[t];
  const float* __restrict__ g = grp.g[t];
  float* __restrict__ m = grp.m[t];
  float* __restrict__ v = grp.v[t];
  const int64_t n = grp.size[t];
  const int64_t base = (int64_t)((int)blockIdx.x - grp.first[t]) * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + 256 * u;
    if (i >= n) continue;
    const float gi = g[i];
    const float mi = m[i] + w1 * (gi - m[i]);
    const float vi = v[i] * b2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}
}  // namespace

ORCA_API int orca_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* sizes, double lr, double beta1, double beta2, double eps,
                            int64_t step, void* stream) {
  ORCA_REQUIRE(n >= 0 && step >= 1 && (n == 0 || (params && grads && exp_avg && exp_avg_sq && sizes)),
               "orca_adam_step: bad arguments");
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)std::sqrt(bc2);
  for (int i0 = 0; i0 < n; i0 += ADAM_MAX) {
    AdamGroup grp{};
    int first = 0, k = 0;
    for (int i = i0; i < n && i < i0 + ADAM_MAX; ++i) {
      if (sizes[i] <= 0) continue;
      ORCA_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], "orca_adam_step: null tensor %d", i);
      ORCA_REQUIRE(sizes[i] < (1LL << 40), "orca_adam_step: tensor %d too large", i);
      grp.first[k] = first;
      grp.p[k] = params[i]; grp.g[k] = grads[i]; grp.m[k] = exp_avg[i]; grp.v[k] = exp_avg_sq[i]; grp.size[k] = sizes[i];
      first += (int)ceil_div64(sizes[i], 1024);
      ++k;
    }
    if (k == 0) continue;
    grp.n = k;
    grp.first[k] = first;
    k_adam<<<(unsigned)first, 256, 0, (hipStream_t)stream>>>(grp, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size,
                                                             bc2_sqrt, (float)eps);
  }
  ORCA_LAUNCH_CHECK();
  return ORCA_OK;
}
