// Adam over a flat parameter arena: one streaming pass per optimizer step.
//
// The reference steps torch.optim.Adam over ~150 parameter tensors per model (Training/Optimizers.py:6-16,50-92) and, under
// bf16 autocast, casts every weight again on first use after the step; zeroing the gradients is a third pass
// (optimizer.zero_grad, Optimizers.py:82).  With parameters, gradients and both moments laid out as flat fp32 arenas
// (branchgan/optim.py) the whole update is ONE launch that reads p, g, m, v once and writes p, m, v, the bf16 copy of p the
// next forward's GEMMs consume, and g = 0:  16 B read + 18 B written per parameter, HBM-bound.
// See include/branchgan.h: bg_adam_step.
#include "common.h"

namespace bg {

struct AdamArgs {
  float* p;
  float* g;
  float* m;
  float* v;
  void* lp;
  int64_t n;
  float step_size;      // lr / (1 - beta1^t)
  float inv_bc2_sqrt;   // 1 / sqrt(1 - beta2^t)
  float one_minus_beta1, beta2, one_minus_beta2, eps, weight_decay, grad_scale;   // 1 - beta rounded from double, as torch passes them
  int zero_grad;
};

// torch.optim.adam._single_tensor_adam, in its order of operations:
//   g' = g * grad_scale (+ weight_decay * p);  m += (g' - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g' * g';
//   p -= step_size * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
__device__ __forceinline__ void adam_one(const AdamArgs& a, float& p, float g, float& m, float& v) {
  g *= a.grad_scale;
  if (a.weight_decay != 0.f) g += a.weight_decay * p;
  m += (g - m) * a.one_minus_beta1;
  v = v * a.beta2 + a.one_minus_beta2 * g * g;
  const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
  p -= a.step_size * (m / denom);
}

#define ADAM_THREADS 256
// ADAM_U: 16-byte vectors per array per thread per iteration; NT: nontemporal loads / stores for the streamed arrays
template <bool HAS_LP, int ADAM_U, bool NT>
__global__ __launch_bounds__(ADAM_THREADS) void adam_step_kernel(AdamArgs a) {
  const int64_t nvec = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * ADAM_THREADS;
  f32x4* __restrict__ P = (f32x4*)a.p;
  f32x4* __restrict__ G = (f32x4*)a.g;
  f32x4* __restrict__ M = (f32x4*)a.m;
  f32x4* __restrict__ V = (f32x4*)a.v;
  bf16x4* __restrict__ L = (bf16x4*)a.lp;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i0 = (int64_t)blockIdx.x * ADAM_THREADS + threadIdx.x; i0 < nvec; i0 += stride * ADAM_U) {
    f32x4 p[ADAM_U], g[ADAM_U], m[ADAM_U], v[ADAM_U];
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < nvec) {
        if (NT) {
          p[u] = __builtin_nontemporal_load(P + i);
          g[u] = __builtin_nontemporal_load(G + i);
          m[u] = __builtin_nontemporal_load(M + i);
          v[u] = __builtin_nontemporal_load(V + i);
        } else {
          p[u] = P[i], g[u] = G[i], m[u] = M[i], v[u] = V[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < nvec) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pj = p[u][j], mj = m[u][j], vj = v[u][j];
          adam_one(a, pj, g[u][j], mj, vj);
          p[u][j] = pj, m[u][j] = mj, v[u][j] = vj;
        }
        if (NT) {
          __builtin_nontemporal_store(p[u], P + i);
          __builtin_nontemporal_store(m[u], M + i);
          __builtin_nontemporal_store(v[u], V + i);
          if (a.zero_grad) __builtin_nontemporal_store(zero, G + i);
        } else {
          P[i] = p[u], M[i] = m[u], V[i] = v[u];
          if (a.zero_grad) G[i] = zero;
        }
        if (HAS_LP) {
          bf16x4 l;
#pragma unroll
          for (int j = 0; j < 4; ++j) l[j] = (bf16_t)p[u][j];
          L[i] = l;     // the next forward reads these: leave them to the caches
        }
      }
    }
  }
  // n % 4 trailing elements
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    float p = a.p[i], m = a.m[i], v = a.v[i];
    adam_one(a, p, a.g[i], m, v);
    a.p[i] = p;
    a.m[i] = m;
    a.v[i] = v;
    if (a.zero_grad) a.g[i] = 0.f;
    if (HAS_LP) ((bf16_t*)a.lp)[i] = (bf16_t)p;
  }
}

}  // namespace bg

extern "C" int bg_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* param_lp, int lp_dtype, int64_t n,
                            double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                            double grad_scale, int zero_grad, void* stream) {
  using namespace bg;
  BG_REQUIRE(n >= 0, "bg_adam_step: n %lld", (long long)n);
  if (n == 0) return BG_OK;
  BG_REQUIRE(param && grad && exp_avg && exp_avg_sq, "bg_adam_step: null argument");
  BG_REQUIRE(step >= 1, "bg_adam_step: step %lld (the first step is 1)", (long long)step);
  BG_REQUIRE(!param_lp || lp_dtype == BG_BF16, "bg_adam_step: param_lp dtype %d (bf16 only)", lp_dtype);
  BG_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 && (uintptr_t)param_lp % 8 == 0,
             "bg_adam_step: arenas must be 16-byte aligned");
  BG_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "bg_adam_step: hyper-parameters out of range");
  AdamArgs a;
  a.p = param, a.g = grad, a.m = exp_avg, a.v = exp_avg_sq, a.lp = param_lp, a.n = n;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.step_size = (float)(lr / bc1);
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.one_minus_beta1 = (float)(1.0 - beta1), a.beta2 = (float)beta2, a.one_minus_beta2 = (float)(1.0 - beta2), a.eps = (float)eps, a.weight_decay = (float)weight_decay;
  a.grad_scale = (float)grad_scale;
  a.zero_grad = zero_grad;
  // one 16-byte vector of each array per thread and no grid-stride loop: measured 5.8 TB/s on a 124M-parameter arena against
  // 4.5 TB/s for 4096 looping workgroups (the streams of a looping workgroup drift apart across HBM channels)
  const int64_t per_block = (int64_t)ADAM_THREADS * 4;
  const unsigned grid = (unsigned)(cdiv(n, per_block) < (1 << 24) ? cdiv(n, per_block) : (1 << 24));
  hipStream_t st = (hipStream_t)stream;
  if (param_lp)
    hipLaunchKernelGGL((adam_step_kernel<true, 1, true>), dim3(grid), dim3(ADAM_THREADS), 0, st, a);
  else
    hipLaunchKernelGGL((adam_step_kernel<false, 1, true>), dim3(grid), dim3(ADAM_THREADS), 0, st, a);
  BG_CHECK_LAUNCH("bg_adam_step");
  return BG_OK;
}
