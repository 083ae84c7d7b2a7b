// Fused softmax statistics over the vocabulary (HBM-bound, one read of the logits).
//
// Replaces Models/Generator.py:47 (softmax over V, full probs tensor), the gathers of
// BranchPredictions/MultiStepPrediction.py:238-240,257-261 and the cross-entropy of
// Models/Losses/GeneratorLosses.py:133-137.  See include/branchgan.h: bg_softmax_stats_fwd / _bwd.
//
// One 256-thread workgroup per row; 16-byte vector loads; online (max, sum-exp) per lane, merged with
// wavefront shuffles then across the 4 waves through LDS.  The backward writes
//   dlogits = softmax * (g_lse - Σ g_logp) + Σ g_logp·onehot(ids)
// in one pass (read logits, write dlogits), never forming the probabilities in memory.
#include "common.h"

namespace bg {

#define BG_MAX_GATHER 80

template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  typedef f32x4 type;
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  typedef bf16x8 type;
};

__device__ __forceinline__ void ms_combine(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { m = mn; s = 0.f; return; }
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_stats_fwd_kernel(const T* logits, int64_t V, int64_t stride, const int64_t* ids,
                                                                int n_ids, float* out_lse, float* out_logp) {
  constexpr int VEC = Vec<T>::N;
  __shared__ float red_m[4], red_s[4];
  __shared__ float lse_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const T* x = logits + row * stride;
  const int mis = (int)((((uintptr_t)x) & 15) / sizeof(T));
  const int64_t vbegin = -mis;
  float m = -INFINITY, s = 0.f;
  auto fold = [&](const float (&v)[VEC]) __attribute__((always_inline)) {
    float vm = v[0];
#pragma unroll
    for (int j = 1; j < VEC; ++j) vm = fmaxf(vm, v[j]);
    if (vm > m) {
      s *= __expf(m - vm);  // m = -inf: s is 0 and exp(-inf) = 0
      m = vm;
    }
    if (m != -INFINITY) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += __expf(v[j] - m);
    }
  };
  auto guarded = [&](int64_t cb) __attribute__((always_inline)) {      // a chunk that may straddle an end of the row
    const int64_t i0 = cb + (int64_t)tid * VEC;
    float v[VEC];
    if (i0 >= 0 && i0 + VEC <= V) {
      typename Vec<T>::type raw = *(const typename Vec<T>::type*)(x + i0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = (float)raw[j];
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = (i0 + j >= 0 && i0 + j < V) ? to_f32(x[i0 + j]) : -INFINITY;
    }
    fold(v);
  };
  // Chunks that lie wholly inside the row go four at a time, their loads requested before the first is folded (round 3: one
  // load per iteration with the exponentials behind it left a single vector in flight per wave); a thread still folds its
  // vectors in increasing order, so the result is the same to the bit.
  constexpr int64_t CH = 256 * VEC;
  constexpr int U = 4;
  int64_t cb = vbegin;
  if (cb < 0) { guarded(cb); cb += CH; }
  for (; cb + U * CH <= V; cb += U * CH) {
    typename Vec<T>::type raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) raw[u] = *(const typename Vec<T>::type*)(x + cb + u * CH + (int64_t)tid * VEC);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = (float)raw[u][j];
      fold(v);
    }
  }
  for (; cb < V; cb += CH) guarded(cb);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ms_combine(m, s, __shfl_xor(m, o), __shfl_xor(s, o));
  if (lane == 0) { red_m[wave] = m; red_s[wave] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = red_m[0], S = red_s[0];
    for (int w = 1; w < 4; ++w) ms_combine(M, S, red_m[w], red_s[w]);
    const float lse = M + logf(S);
    lse_sh = lse;
    out_lse[row] = lse;
  }
  __syncthreads();
  const float lse = lse_sh;
  for (int i = tid; i < n_ids; i += 256) {
    const int64_t id = ids[row * n_ids + i];
    out_logp[row * n_ids + i] = (id >= 0 && id < V) ? to_f32(x[id]) - lse : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_stats_bwd_kernel(const T* logits, int64_t V, int64_t stride, const int64_t* ids,
                                                                int n_ids, const float* lse_p, const float* g_lse,
                                                                const float* g_logp, T* dlogits) {
  constexpr int VEC = Vec<T>::N;
  __shared__ uint32_t sh_id[BG_MAX_GATHER];
  __shared__ float sh_g[BG_MAX_GATHER];
  __shared__ float sh_c;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const T* x = logits + row * stride;
  T* dx = dlogits + row * stride;
  if (tid < n_ids) {
    const int64_t id = ids[row * n_ids + tid];
    const bool ok = id >= 0 && id < V;
    sh_id[tid] = ok ? (uint32_t)id : 0xFFFFFFFFu;
    sh_g[tid] = (ok && g_logp) ? g_logp[row * n_ids + tid] : 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    float c = g_lse ? g_lse[row] : 0.f;
    for (int i = 0; i < n_ids; ++i) c -= sh_g[i];
    sh_c = c;
  }
  __syncthreads();
  const float c = sh_c, lse = lse_p[row];
  const int mis = (int)((((uintptr_t)x) & 15) / sizeof(T));
  const bool out_vec_ok = ((((uintptr_t)dx) & 15) / sizeof(T)) == (unsigned)mis;
  const int64_t vbegin = -mis;
  auto finish = [&](int64_t i0, float (&v)[VEC], bool full) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = c * __expf(v[j] - lse);
    for (int i = 0; i < n_ids; ++i) {
      const uint32_t d = sh_id[i] - (uint32_t)i0;  // i0 may be negative only in the guarded head vector
      const float g = sh_g[i];
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (d == (uint32_t)j) v[j] += g;
    }
    if (full && out_vec_ok) {
      typename Vec<T>::type o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = from_f32<T>(v[j]);
      *(typename Vec<T>::type*)(dx + i0) = o;
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (i0 + j >= 0 && i0 + j < V) dx[i0 + j] = from_f32<T>(v[j]);
    }
  };
  auto guarded = [&](int64_t cb) __attribute__((always_inline)) {
    const int64_t i0 = cb + (int64_t)tid * VEC;
    if (i0 >= V || i0 + VEC <= 0) return;
    const bool full = (i0 >= 0 && i0 + VEC <= V);
    float v[VEC];
    if (full) {
      typename Vec<T>::type raw = *(const typename Vec<T>::type*)(x + i0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = (float)raw[j];
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = (i0 + j >= 0 && i0 + j < V) ? to_f32(x[i0 + j]) : 0.f;
    }
    finish(i0, v, full);
  };
  // One chunk per iteration here, unlike the forward kernel: the four-at-a-time form of this loop (loads first) was
  // bit-identical and 2x faster at 256 rows, but with it the configs[3] step stopped making progress whenever the two backward
  // graphs were replayed side by side (this kernel beside the discriminator's GEMMs; fine alone, fine with the graphs in
  // turn) - not understood then, so not shipped; the likeliest reading since (two stream-K GEMMs of the two streams waiting for
  // each other's workgroups, with this kernel only moving the timing) is in DESIGN §5 item 12, and steps of two large models no
  // longer replay side by side.
  constexpr int64_t CH = 256 * VEC;
  int64_t cb = vbegin;
  for (; cb < V; cb += CH) guarded(cb);
}

}  // namespace bg

using namespace bg;

extern "C" int bg_softmax_stats_fwd(const void* logits, int dtype, int64_t rows, int64_t V, int64_t row_stride,
                                    const int64_t* ids, int n_ids, float* out_lse, float* out_logp, void* stream) {
  BG_REQUIRE(logits && out_lse, "bg_softmax_stats_fwd: null logits/out_lse");
  BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16, "bg_softmax_stats_fwd: dtype %d", dtype);
  BG_REQUIRE(rows > 0 && rows < (1ll << 31) && V > 0 && V < (1ll << 31) && row_stride >= V, "bg_softmax_stats_fwd: rows=%lld V=%lld stride=%lld",
             (long long)rows, (long long)V, (long long)row_stride);
  BG_REQUIRE(n_ids >= 0 && n_ids <= BG_MAX_GATHER && (n_ids == 0 || (ids && out_logp)), "bg_softmax_stats_fwd: n_ids=%d (max %d)", n_ids, BG_MAX_GATHER);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BG_F32)
    hipLaunchKernelGGL(softmax_stats_fwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, V, row_stride, ids, n_ids, out_lse, out_logp);
  else
    hipLaunchKernelGGL(softmax_stats_fwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, V, row_stride, ids, n_ids, out_lse, out_logp);
  BG_CHECK_LAUNCH("bg_softmax_stats_fwd");
  return BG_OK;
}

extern "C" int bg_softmax_stats_bwd(const void* logits, int dtype, int64_t rows, int64_t V, int64_t row_stride,
                                    const int64_t* ids, int n_ids, const float* lse, const float* g_lse,
                                    const float* g_logp, void* dlogits, void* stream) {
  BG_REQUIRE(logits && lse && dlogits, "bg_softmax_stats_bwd: null logits/lse/dlogits");
  BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16, "bg_softmax_stats_bwd: dtype %d", dtype);
  BG_REQUIRE(rows > 0 && rows < (1ll << 31) && V > 0 && V < (1ll << 31) && row_stride >= V, "bg_softmax_stats_bwd: rows=%lld V=%lld stride=%lld",
             (long long)rows, (long long)V, (long long)row_stride);
  BG_REQUIRE(n_ids >= 0 && n_ids <= BG_MAX_GATHER && (n_ids == 0 || ids), "bg_softmax_stats_bwd: n_ids=%d (max %d)", n_ids, BG_MAX_GATHER);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BG_F32)
    hipLaunchKernelGGL(softmax_stats_bwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, V, row_stride, ids, n_ids, lse, g_lse, g_logp, (float*)dlogits);
  else
    hipLaunchKernelGGL(softmax_stats_bwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, st, (const bf16_t*)logits, V, row_stride, ids, n_ids, lse, g_lse, g_logp, (bf16_t*)dlogits);
  BG_CHECK_LAUNCH("bg_softmax_stats_bwd");
  return BG_OK;
}
