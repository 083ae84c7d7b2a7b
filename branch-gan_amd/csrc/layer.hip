// Row-wise glue of the transformer block between the GEMMs, fused so that each activation crosses HBM once.
//
// The reference builds these from separate ATen ops per block and per tree node (Models/GPT2BranchAttention/
// NonResidualDiscriminatorBase.py:46-90 block forward: ln_1 / residual add / ln_2 / MLP; NeoXNonResidualBase.py:73-133),
// and autograd adds one kernel per cast, residual add, bias-gradient reduction and LayerNorm-parameter reduction.
// See include/branchgan.h: bg_add_layernorm_fwd, bg_layernorm_bwd, bg_bias_act_bwd.
//
// All three are HBM-bound streaming kernels: one wave per row (LayerNorm) or one thread per 4 columns (bias/activation),
// 16-byte loads, fp32 arithmetic, residual stream and statistics in fp32, activations in fp32 or bf16.
// Column reductions (dgamma, dbeta, dbias) are deterministic: per-workgroup partials in the caller's workspace, then a
// second small kernel adds them in a fixed order into the fp32 gradient (accumulating: grad += sum).
#include "common.h"
#include "dropout_common.h"

#include <vector>

namespace bg {

#define LN_MAX_NV 16   // 16-byte fp32 vectors per lane: C <= 64 * 4 * 16 = 4096

template <typename T> struct V4;
template <> struct V4<float> {
  typedef f32x4 type;
};
template <> struct V4<bf16_t> {
  typedef bf16x4 type;
};

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]) {
  typename V4<T>::type raw = *(const typename V4<T>::type*)p;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (float)raw[j];
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]) {
  typename V4<T>::type raw;
#pragma unroll
  for (int j = 0; j < 4; ++j) raw[j] = from_f32<T>(v[j]);
  *(typename V4<T>::type*)p = raw;
}

struct LnFwdArgs {
  const float* x;
  const void* delta;
  const void* delta2;   // optional second update of the same dtype (parallel-residual blocks: attention + MLP)
  const float* gamma;
  const float* beta;
  float* x_out;
  void* y;
  float* mean;
  float* rstd;
  int64_t rows;
  int C;
  float eps;
  const float* gamma_b;   // optional second LayerNorm of the same row (parallel-residual blocks normalise x + delta twice)
  const float* beta_b;
  void* y_b;
  RowDropArgs drop;       // DROP kernels: x + dropout(delta) (bg_add_layernorm_dropout_fwd)
};

// TD: dtype of delta (ignored when delta == nullptr), TY: dtype of y.  DROP: delta passes through a dropout on its way into the
// residual stream (mask = function of (step, site, batch, logical row, column), dropout_common.h) - the block's
// `x + dropout(projection)` without a kernel and an activation round trip of its own.
template <typename TD, typename TY, int NV, bool DROP = false>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(LnFwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= a.rows) return;
  const int C = a.C;
  const float* x = a.x + r * C;
  const TD* d = a.delta ? (const TD*)a.delta + r * C : nullptr;
  const TD* d2 = a.delta2 ? (const TD*)a.delta2 + r * C : nullptr;
  float v[NV][4];
  float s = 0.f;
  [[maybe_unused]] DropKeys dk;
  [[maybe_unused]] uint32_t row_key = 0;
  if constexpr (DROP) row_key = row_drop_key(a.drop, r, dk);
  // Every load of the row is REQUESTED before the first one is used (round 3): with the adds inside the `c < C` branch that
  // guards each vector the compiler waited for a vector before it asked for the next - up to six dependent round trips for a
  // 768-wide row with a delta, which is most of the life of the 256-row launches of a level pass.
  typedef typename V4<TD>::type DV;
  DV rd[NV], rd2[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[i][j] = 0.f; rd[i][j] = from_f32<TD>(0.f); rd2[i][j] = from_f32<TD>(0.f); }
    if (c < C) {
      load4<float>(x + c, v[i]);
      if (d) rd[i] = *(const DV*)(d + c);
      if (d2) rd2[i] = *(const DV*)(d2 + c);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    if (c < C) {
      if (d) {
        float dv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dv[j] = (float)rd[i][j];
        if constexpr (DROP) {
#pragma unroll
          for (int j = 0; j < 4; ++j) dv[j] = drop_keep(dk, row_key, (uint32_t)(c + j)) ? dv[j] * dk.inv_keep : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] += dv[j];
        if (d2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i][j] += (float)rd2[i][j];
        }
        store4<float>(a.x_out + r * C + c, v[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) s += v[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    if (c < C) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = v[i][j] - mean;
        q += t * t;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + a.eps);
  if (lane == 0) {
    a.mean[r] = mean;
    a.rstd[r] = rstd;
  }
  TY* y = (TY*)a.y + r * C;
  // gamma / beta of four vectors at a time, all requested before the first is used (same reason as above)
  constexpr int CH = NV < 4 ? NV : 4;
#pragma unroll
  for (int i0 = 0; i0 < NV; i0 += CH) {
    float g[CH][4], b[CH][4], g2[CH][4], b2[CH][4];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = 4 * (lane + 64 * (i0 + k));
#pragma unroll
      for (int j = 0; j < 4; ++j) g[k][j] = b[k][j] = g2[k][j] = b2[k][j] = 0.f;
      if (c < C) {
        load4<float>(a.gamma + c, g[k]);
        load4<float>(a.beta + c, b[k]);
        if (a.y_b) {
          load4<float>(a.gamma_b + c, g2[k]);
          load4<float>(a.beta_b + c, b2[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int i = i0 + k;
      const int c = 4 * (lane + 64 * i);
      if (c < C) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[k][j] + b[k][j];
        store4<TY>(y + c, o);
        if (a.y_b) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g2[k][j] + b2[k][j];
          store4<TY>((TY*)a.y_b + r * C + c, o);
        }
      }
    }
  }
}

struct LnBwdArgs {
  const void* dy;
  const float* x;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* dres;
  float* dx;
  void* dx_lp;
  float* partial;   // [gridDim.x][2 or 3][C]
  int64_t rows;
  int C;
  int wave_lds;     // LDS holds one [2 or 3][C] region per wave (set by the launcher when that fits 64 KB)
  RowDropArgs drop; // DROP kernels: dx_lp and its column sums are the gradient of a delta that entered through a dropout
};

// TY: dtype of dy;  TL: dtype of the low-precision copy of dx (ignored when dx_lp == nullptr);  COLSUM: also the column sums
// of dx as the preceding projection sees it (rounded to TL when dx_lp is given) - that projection's bias gradient
template <typename TY, typename TL, int NV, bool COLSUM, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(LnBwdArgs a) {
  extern __shared__ float lds[];   // [2 or 3][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = a.C;
  const float inv_c = 1.0f / (float)C;
  float gam[NV][4], dg[NV][4], db[NV][4], ds[COLSUM ? NV : 1][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) gam[i][j] = dg[i][j] = db[i][j] = ds[COLSUM ? i : 0][j] = 0.f;
    if (c < C) load4<float>(a.gamma + c, gam[i]);
  }
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < a.rows; r += (int64_t)gridDim.x * 4) {
    const float mean = a.mean[r], rstd = a.rstd[r];
    const float* x = a.x + r * C;
    const TY* dy = (const TY*)a.dy + r * C;
    float xh[NV][4], g[NV][4], rv[NV][4];
    float s1 = 0.f, s2 = 0.f;
    // every global load of the row is issued before the first use: one memory round trip, not two
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) rv[i][j] = 0.f;
      if (c < C && a.dres) load4<float>(a.dres + r * C + c, rv[i]);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) xh[i][j] = g[i][j] = 0.f;
      if (c < C) {
        float xv[4], dv[4];
        load4<float>(x + c, xv);
        load4<TY>(dy + c, dv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (xv[j] - mean) * rstd;
          g[i][j] = dv[j] * gam[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          dg[i][j] += dv[j] * xh[i][j];
          db[i][j] += dv[j];
        }
      }
    }
    s1 = wave_sum(s1) * inv_c;
    s2 = wave_sum(s2) * inv_c;
    [[maybe_unused]] DropKeys dk;
    [[maybe_unused]] uint32_t row_key = 0;
    if constexpr (DROP) row_key = row_drop_key(a.drop, r, dk);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < C) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2) + rv[i][j];
        store4<float>(a.dx + r * C + c, o);
        if constexpr (DROP) {                          // what delta's producer sees: the gradient through the dropout
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = drop_keep(dk, row_key, (uint32_t)(c + j)) ? o[j] * dk.inv_keep : 0.f;
        }
        if (a.dx_lp) store4<TL>((TL*)a.dx_lp + r * C + c, o);
        if constexpr (COLSUM) {
#pragma unroll
          for (int j = 0; j < 4; ++j) ds[i][j] += a.dx_lp ? to_f32(from_f32<TL>(o[j])) : o[j];
        }
      }
    }
  }
  // the four waves' column partials, added in wave order
  constexpr int NP = COLSUM ? 3 : 2;
  float* part = a.partial + (int64_t)blockIdx.x * NP * C;
  if (a.wave_lds) {
    // every wave leaves its partials in its own LDS region (vector stores, one barrier); the sum over the regions below
    // runs in the same order as the wave-by-wave accumulation of the narrow-LDS path: identical bits
    float* mine = lds + wave * NP * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < C) {
        store4<float>(mine + c, dg[i]);
        store4<float>(mine + C + c, db[i]);
        if constexpr (COLSUM) store4<float>(mine + 2 * C + c, ds[i]);
      }
    }
    __syncthreads();
    for (int c = 4 * threadIdx.x; c < NP * C; c += 4 * 256) {
      float acc[4], t[4];
      load4<float>(lds + c, acc);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        load4<float>(lds + w * NP * C + c, t);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += t[j];
      }
      store4<float>(part + c, acc);
    }
    return;
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        if (c < C) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (w == 0) {
              lds[c + j] = dg[i][j];
              lds[C + c + j] = db[i][j];
              if constexpr (COLSUM) lds[2 * C + c + j] = ds[i][j];
            } else {
              lds[c + j] += dg[i][j];
              lds[C + c + j] += db[i][j];
              if constexpr (COLSUM) lds[2 * C + c + j] += ds[i][j];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < NP * C; c += 256) part[c] = lds[c];
}

// out[c] += sum_g partial[g * stride + c] for each of up to REDUCE_BATCH descriptors in one launch.  256 threads = 32 columns
// x 8 slices of g; every thread adds its slice in increasing g, then the 8 slice sums are added in slice order: a fixed
// summation order, the same whether a descriptor is reduced right behind its producer or with hundreds of others at the end
// of a backward pass (bg_partial_reduce).
constexpr int REDUCE_BATCH = 96;
struct ReduceDesc {
  const float* partial;
  float* out;
  int G, stride, n, first_block;      // first_block: prefix sum of cdiv(n, 32) over the descriptors before this one
};
struct ReduceBatch {
  ReduceDesc d[REDUCE_BATCH];
  int count;
};

__global__ __launch_bounds__(256) void partial_reduce_kernel(ReduceBatch b) {
  __shared__ float red[8][33];
  int lo = 0, hi = b.count - 1;          // the last descriptor whose first_block <= blockIdx.x (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.d[mid].first_block <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const float* partial = b.d[lo].partial;
  float* out = b.d[lo].out;
  const int G = b.d[lo].G, stride = b.d[lo].stride, n = b.d[lo].n;
  const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int c = ((int)blockIdx.x - b.d[lo].first_block) * 32 + col;
  float s = 0.f;
  if (c < n) {
#pragma unroll 8
    for (int g = slice; g < G; g += 8) s += partial[(int64_t)g * stride + c];
  }
  red[slice][col] = s;
  __syncthreads();
  if (slice == 0 && c < n) {
    float t = red[0][col];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][col];
    out[c] += t;
  }
}

// host side: packs descriptors into launches of REDUCE_BATCH.
struct ReduceItem {
  const float* partial;
  float* out;
  int G, stride, n;
};
// Reductions into the same (or an overlapping) `out` must not share a launch: item i goes into round r(i) = 1 + the highest
// round among the earlier items it overlaps, and the rounds run one after another - every `out` receives its partial sums in
// the order the items were given.
static int launch_reduce(const ReduceItem* items, int count, hipStream_t st) {
  std::vector<int> round((size_t)count, 0);
  int rounds = 1;
  for (int i = 1; i < count; ++i) {
    for (int j = 0; j < i; ++j) {
      if (items[i].out < items[j].out + items[j].n && items[j].out < items[i].out + items[i].n && round[j] + 1 > round[i]) round[i] = round[j] + 1;
    }
    if (round[i] + 1 > rounds) rounds = round[i] + 1;
  }
  for (int r = 0; r < rounds; ++r) {
    ReduceBatch b;
    int m = 0, blocks = 0;
    for (int i = 0; i <= count; ++i) {
      if (i < count && round[i] == r) {
        const ReduceItem& it = items[i];
        b.d[m++] = ReduceDesc{it.partial, it.out, it.G, it.stride, it.n, blocks};
        blocks += (int)cdiv(it.n, 32);
      }
      if (m == REDUCE_BATCH || (i == count && m > 0)) {
        b.count = m;
        hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
        m = blocks = 0;
      }
    }
  }
  return 0;
}

// ---- bias gradient (column sums) with the activation's backward in the same pass ------------------------------------------
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float k = 0.7978845608028654f, c = 0.044715f;
  const float u = k * (x + c * x * x * x);
  const float t = 1.f - 2.f / (__expf(2.f * u) + 1.f);     // tanh(u); exp overflow -> 1, underflow -> -1
  return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * k * (1.f + 3.f * c * x * x);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.7071067811865476f));
  return cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

struct BiasActArgs {
  const void* dy;
  const void* pre;
  void* dx;
  float* partial;   // [gridDim.x][C]
  int64_t rows, dy_stride;
  int C;
};

// MODE 0: column sums of dy only;  1: dx = dy * gelu_tanh'(pre);  2: dx = dy * gelu_erf'(pre); sums of dx
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(BiasActArgs a) {
  const int C = a.C;
  const int G = gridDim.x;
  float* part = a.partial + (int64_t)blockIdx.x * C;
  const int c = 4 * (blockIdx.y * 256 + threadIdx.x);      // grid.y column chunks of 1024
  if (c < C) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto one = [&](int64_t r, float (&d)[4], const float (&p)[4]) __attribute__((always_inline)) {
      if constexpr (MODE != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] *= (MODE == 1) ? gelu_tanh_grad(p[j]) : gelu_erf_grad(p[j]);
        store4<T>((T*)a.dx + r * C + c, d);
        // the bias gradient is the sum of the ROUNDED values the next GEMM sees, like autograd's sum over dx
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = to_f32(from_f32<T>(d[j]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += d[j];
    };
    // Four rows at a time, their loads requested before the first is used (round 3): a thread's rows are G apart and each used
    // to be one dependent round trip (the store of dx between two rows' loads keeps the compiler from moving them) - twelve in
    // a row at 3072 rows.  Rows are still added in increasing order: same sums.
    constexpr int UR = 4;
    int64_t r = blockIdx.x;
    for (; r + (UR - 1) * (int64_t)G < a.rows; r += (int64_t)UR * G) {
      float d[UR][4], p[UR][4];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        load4<T>((const T*)a.dy + (r + u * (int64_t)G) * a.dy_stride + c, d[u]);
        if constexpr (MODE != 0) load4<T>((const T*)a.pre + (r + u * (int64_t)G) * C + c, p[u]);
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) one(r + u * (int64_t)G, d[u], p[u]);
    }
    for (; r < a.rows; r += G) {
      float d[4], p[4] = {0.f, 0.f, 0.f, 0.f};
      load4<T>((const T*)a.dy + r * a.dy_stride + c, d);
      if constexpr (MODE != 0) load4<T>((const T*)a.pre + r * C + c, p);
      one(r, d, p);
    }
    store4<float>(part + c, acc);
  }
}

static int ln_groups(int64_t rows) {       // every wave gets the same number of rows (k), at most 512 workgroups of 4 waves
  const int64_t g1 = cdiv(rows, 4);
  if (g1 <= 512) return (int)(g1 < 1 ? 1 : g1);
  const int64_t k = cdiv(g1, 512);
  return (int)cdiv(g1, k);
}
static int bias_groups(int64_t rows) {
  int64_t g = cdiv(rows, 8);      // measured best at (2048..3072) x 3072: more groups cost more in the reduction than they gain
  return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

template <typename TD, typename TY, bool DROP = false>
static void launch_ln_fwd(const LnFwdArgs& a, hipStream_t st) {
  dim3 grid((unsigned)cdiv(a.rows, 4)), block(256);
  const int nv = (int)cdiv(a.C, 256);
  if (nv <= 2) hipLaunchKernelGGL((add_layernorm_fwd_kernel<TD, TY, 2, DROP>), grid, block, 0, st, a);
  else if (nv <= 4) hipLaunchKernelGGL((add_layernorm_fwd_kernel<TD, TY, 4, DROP>), grid, block, 0, st, a);
  else if (nv <= 8) hipLaunchKernelGGL((add_layernorm_fwd_kernel<TD, TY, 8, DROP>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((add_layernorm_fwd_kernel<TD, TY, 16, DROP>), grid, block, 0, st, a);
}

template <typename TY, typename TL, bool COLSUM, bool DROP = false>
static void launch_ln_bwd2(const LnBwdArgs& a0, int G, hipStream_t st) {
  dim3 grid((unsigned)G), block(256);
  LnBwdArgs a = a0;
  size_t lds = (size_t)(COLSUM ? 3 : 2) * a.C * sizeof(float);
  a.wave_lds = (4 * lds <= 65536 && a.C % 4 == 0) ? 1 : 0;
  if (a.wave_lds) lds *= 4;
  const int nv = (int)cdiv(a.C, 256);
  if (nv <= 2) hipLaunchKernelGGL((layernorm_bwd_kernel<TY, TL, 2, COLSUM, DROP>), grid, block, lds, st, a);
  else if (nv <= 4) hipLaunchKernelGGL((layernorm_bwd_kernel<TY, TL, 4, COLSUM, DROP>), grid, block, lds, st, a);
  else if (nv <= 8) hipLaunchKernelGGL((layernorm_bwd_kernel<TY, TL, 8, COLSUM, DROP>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((layernorm_bwd_kernel<TY, TL, 16, COLSUM, DROP>), grid, block, lds, st, a);
}
template <typename TY, typename TL, bool DROP = false>
static void launch_ln_bwd(const LnBwdArgs& a, int G, bool colsum, hipStream_t st) {
  if (colsum) launch_ln_bwd2<TY, TL, true, DROP>(a, G, st);
  else launch_ln_bwd2<TY, TL, false, DROP>(a, G, st);
}

}  // namespace bg

using namespace bg;

static bool dtype_ok(int d) { return d == BG_F32 || d == BG_BF16; }
static bool al16(const void* p) { return p && (((uintptr_t)p) & 15) == 0; }

// -> 0, or an error code with the message set: the row-dropout descriptor of the *_dropout_* entry points
static int check_row_dropout(const bg_row_dropout* d, int64_t rows, const char* who, RowDropArgs* out) {
  BG_REQUIRE(d && d->drop_state, "%s: null dropout descriptor / state", who);
  BG_REQUIRE(d->p > 0.f && d->p < 1.f, "%s: p = %g is not in (0, 1)", who, (double)d->p);
  BG_REQUIRE(d->rows_per_batch > 0 && rows % d->rows_per_batch == 0 && rows / d->rows_per_batch < 4096, "%s: %lld rows are not a whole number (< 4096) of batches of %d",
             who, (long long)rows, d->rows_per_batch);
  BG_REQUIRE(d->text_rows >= 0 && d->text_rows <= (1 << 20), "%s: text_rows %d", who, d->text_rows);
  *out = RowDropArgs{d->drop_state, d->p, d->site, d->row0, d->text_rows, d->rows_per_batch};
  return BG_OK;
}

static int add_layernorm_fwd_impl(const float* x, const void* delta, const void* delta2, int delta_dtype, int64_t rows, int C,
                                  const float* gamma, const float* beta, const float* gamma_b, const float* beta_b, float eps,
                                  float* x_out, void* y, void* y_b, int y_dtype, float* mean, float* rstd, const bg_row_dropout* drop,
                                  void* stream);

extern "C" int bg_add_layernorm_fwd(const float* x, const void* delta, const void* delta2, int delta_dtype, int64_t rows, int C, const float* gamma,
                                    const float* beta, float eps, float* x_out, void* y, int y_dtype, float* mean, float* rstd,
                                    void* stream) {
  return add_layernorm_fwd_impl(x, delta, delta2, delta_dtype, rows, C, gamma, beta, nullptr, nullptr, eps, x_out, y, nullptr, y_dtype, mean, rstd,
                                nullptr, stream);
}

extern "C" int bg_add_layernorm_pair_fwd(const float* x, const void* delta, const void* delta2, int delta_dtype, int64_t rows, int C,
                                         const float* gamma, const float* beta, const float* gamma_b, const float* beta_b, float eps,
                                         float* x_out, void* y, void* y_b, int y_dtype, float* mean, float* rstd, void* stream) {
  return add_layernorm_fwd_impl(x, delta, delta2, delta_dtype, rows, C, gamma, beta, gamma_b, beta_b, eps, x_out, y, y_b, y_dtype, mean, rstd, nullptr,
                                stream);
}

extern "C" int bg_add_layernorm_dropout_fwd(const float* x, const void* delta, int delta_dtype, int64_t rows, int C, const float* gamma,
                                            const float* beta, float eps, float* x_out, void* y, int y_dtype, float* mean, float* rstd,
                                            const bg_row_dropout* drop, void* stream) {
  BG_REQUIRE(delta && drop, "bg_add_layernorm_dropout_fwd: delta and the dropout descriptor are required");
  return add_layernorm_fwd_impl(x, delta, nullptr, delta_dtype, rows, C, gamma, beta, nullptr, nullptr, eps, x_out, y, nullptr, y_dtype, mean, rstd, drop,
                                stream);
}

static int add_layernorm_fwd_impl(const float* x, const void* delta, const void* delta2, int delta_dtype, int64_t rows, int C,
                                  const float* gamma, const float* beta, const float* gamma_b, const float* beta_b, float eps,
                                  float* x_out, void* y, void* y_b, int y_dtype, float* mean, float* rstd, const bg_row_dropout* drop,
                                  void* stream) {
  BG_REQUIRE(x && gamma && beta && y && mean && rstd, "bg_add_layernorm_fwd: null argument");
  BG_REQUIRE((y_b == nullptr) == (gamma_b == nullptr) && (y_b == nullptr) == (beta_b == nullptr) && (!y_b || (al16(gamma_b) && al16(beta_b) && (((uintptr_t)y_b) & 7) == 0)),
             "bg_add_layernorm_pair_fwd: gamma_b, beta_b and y_b are given together (16-byte aligned; 8 for a bf16 y_b), or all NULL");
  BG_REQUIRE(rows > 0 && rows < (1ll << 31), "bg_add_layernorm_fwd: rows=%lld", (long long)rows);
  BG_REQUIRE(C > 0 && C % 4 == 0 && C <= 256 * LN_MAX_NV, "bg_add_layernorm_fwd: C=%d must be a multiple of 4, at most %d", C, 256 * LN_MAX_NV);
  BG_REQUIRE(dtype_ok(y_dtype) && (!delta || dtype_ok(delta_dtype)), "bg_add_layernorm_fwd: dtype");
  BG_REQUIRE(!delta || x_out, "bg_add_layernorm_fwd: delta without x_out");
  BG_REQUIRE(!delta2 || (delta && (((uintptr_t)delta2) & 7) == 0), "bg_add_layernorm_fwd: delta2 without delta, or misaligned");
  BG_REQUIRE(al16(x) && al16(gamma) && al16(beta) && (!delta || (((uintptr_t)delta) & 7) == 0) && (!delta || al16(x_out)) && (((uintptr_t)y) & 7) == 0,
             "bg_add_layernorm_fwd: pointers must be 16-byte aligned (8 for bf16)");
  LnFwdArgs a{x, delta, delta2, gamma, beta, x_out, y, mean, rstd, rows, C, eps, gamma_b, beta_b, y_b, RowDropArgs{}};
  hipStream_t st = (hipStream_t)stream;
  const bool dbf = delta && delta_dtype == BG_BF16;
  if (drop) {
    if (int e = check_row_dropout(drop, rows, "bg_add_layernorm_dropout_fwd", &a.drop)) return e;
    BG_REQUIRE(delta_dtype == y_dtype, "bg_add_layernorm_dropout_fwd: delta and y share a dtype (bf16 / bf16 or fp32 / fp32)");
    if (dbf) launch_ln_fwd<bf16_t, bf16_t, true>(a, st);
    else launch_ln_fwd<float, float, true>(a, st);
    BG_CHECK_LAUNCH("bg_add_layernorm_dropout_fwd");
    return BG_OK;
  }
  if (y_dtype == BG_BF16) {
    if (dbf) launch_ln_fwd<bf16_t, bf16_t>(a, st);
    else launch_ln_fwd<float, bf16_t>(a, st);
  } else {
    if (dbf) launch_ln_fwd<bf16_t, float>(a, st);
    else launch_ln_fwd<float, float>(a, st);
  }
  BG_CHECK_LAUNCH("bg_add_layernorm_fwd");
  return BG_OK;
}

extern "C" size_t bg_layernorm_bwd_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)ln_groups(rows) * 3 * (size_t)C * sizeof(float);
}

extern "C" int bg_layernorm_bwd_groups(int64_t rows) { return rows > 0 ? ln_groups(rows) : 0; }

static int layernorm_bwd_impl(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* gamma,
                              const float* dres, int64_t rows, int C, float* dx, void* dx_lp, int lp_dtype, float* dgamma,
                              float* dbeta, float* dx_colsum, int deferred_colsum, void* workspace, size_t workspace_bytes,
                              const bg_row_dropout* drop, void* stream);

extern "C" int bg_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* gamma,
                                const float* dres, int64_t rows, int C, float* dx, void* dx_lp, int lp_dtype, float* dgamma,
                                float* dbeta, float* dx_colsum, int deferred_colsum, void* workspace, size_t workspace_bytes,
                                void* stream) {
  return layernorm_bwd_impl(dy, dy_dtype, x, mean, rstd, gamma, dres, rows, C, dx, dx_lp, lp_dtype, dgamma, dbeta, dx_colsum, deferred_colsum,
                            workspace, workspace_bytes, nullptr, stream);
}

extern "C" int bg_layernorm_dropout_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* gamma,
                                        const float* dres, int64_t rows, int C, float* dx, void* dx_lp, int lp_dtype, float* dgamma,
                                        float* dbeta, float* dx_colsum, int deferred_colsum, void* workspace, size_t workspace_bytes,
                                        const bg_row_dropout* drop, void* stream) {
  BG_REQUIRE(drop && dx_lp, "bg_layernorm_dropout_bwd: the dropout descriptor and dx_lp (the gradient through the dropout) are required");
  return layernorm_bwd_impl(dy, dy_dtype, x, mean, rstd, gamma, dres, rows, C, dx, dx_lp, lp_dtype, dgamma, dbeta, dx_colsum, deferred_colsum,
                            workspace, workspace_bytes, drop, stream);
}

static int layernorm_bwd_impl(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* gamma,
                              const float* dres, int64_t rows, int C, float* dx, void* dx_lp, int lp_dtype, float* dgamma,
                              float* dbeta, float* dx_colsum, int deferred_colsum, void* workspace, size_t workspace_bytes,
                              const bg_row_dropout* drop, void* stream) {
  BG_REQUIRE(dy && x && mean && rstd && gamma && dx, "bg_layernorm_bwd: null argument");
  BG_REQUIRE((dgamma != nullptr) == (dbeta != nullptr), "bg_layernorm_bwd: dgamma and dbeta are given together, or both NULL (deferred reduction)");
  BG_REQUIRE(deferred_colsum == 0 || !dgamma, "bg_layernorm_bwd: deferred_colsum is for the deferred form (dgamma == NULL)");
  BG_REQUIRE(rows > 0 && rows < (1ll << 31), "bg_layernorm_bwd: rows=%lld", (long long)rows);
  BG_REQUIRE(C > 0 && C % 4 == 0 && C <= 256 * LN_MAX_NV, "bg_layernorm_bwd: C=%d must be a multiple of 4, at most %d", C, 256 * LN_MAX_NV);
  BG_REQUIRE(dtype_ok(dy_dtype) && (!dx_lp || dtype_ok(lp_dtype)), "bg_layernorm_bwd: dtype");
  BG_REQUIRE(al16(x) && al16(gamma) && al16(dx) && (!dres || al16(dres)) && (((uintptr_t)dy) & 7) == 0 && (((uintptr_t)dx_lp) & 7) == 0,
             "bg_layernorm_bwd: pointers must be 16-byte aligned (8 for bf16)");
  const size_t need = bg_layernorm_bwd_workspace_bytes(rows, C);
  if (!workspace || workspace_bytes < need || !al16(workspace)) {
    set_error("bg_layernorm_bwd: workspace %zu < %zu bytes (or not 16-byte aligned)", workspace_bytes, need);
    return BG_ERR_WORKSPACE;
  }
  const int G = ln_groups(rows);
  const bool colsum = dx_colsum != nullptr || deferred_colsum != 0;
  LnBwdArgs a{dy, x, mean, rstd, gamma, dres, dx, dx_lp, (float*)workspace, rows, C, 0, RowDropArgs{}};
  hipStream_t st = (hipStream_t)stream;
  const bool lbf = dx_lp && lp_dtype == BG_BF16;
  if (drop) {
    if (int e = check_row_dropout(drop, rows, "bg_layernorm_dropout_bwd", &a.drop)) return e;
    BG_REQUIRE(dy_dtype == lp_dtype, "bg_layernorm_dropout_bwd: dy and dx_lp share a dtype (bf16 / bf16 or fp32 / fp32)");
    if (lbf) launch_ln_bwd<bf16_t, bf16_t, true>(a, G, colsum, st);
    else launch_ln_bwd<float, float, true>(a, G, colsum, st);
  } else if (dy_dtype == BG_BF16) {
    if (lbf) launch_ln_bwd<bf16_t, bf16_t>(a, G, colsum, st);
    else launch_ln_bwd<bf16_t, float>(a, G, colsum, st);
  } else {
    if (lbf) launch_ln_bwd<float, bf16_t>(a, G, colsum, st);
    else launch_ln_bwd<float, float>(a, G, colsum, st);
  }
  BG_CHECK_LAUNCH("bg_layernorm_bwd");
  if (dgamma) {
    const float* part = (const float*)workspace;
    const int stride = (colsum ? 3 : 2) * C;
    const ReduceItem items[3] = {{part, dgamma, G, stride, C}, {part + C, dbeta, G, stride, C}, {part + 2 * C, dx_colsum, G, stride, C}};
    launch_reduce(items, colsum ? 3 : 2, st);
    BG_CHECK_LAUNCH("bg_layernorm_bwd(reduce)");
  }
  return BG_OK;
}

extern "C" size_t bg_bias_act_bwd_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)bias_groups(rows) * (size_t)C * sizeof(float);
}

extern "C" int bg_bias_act_bwd_groups(int64_t rows) { return rows > 0 ? bias_groups(rows) : 0; }

extern "C" int bg_bias_act_bwd(const void* dy, int dtype, int64_t rows, int C, int64_t dy_row_stride, int act, const void* pre, void* dx,
                               float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  BG_REQUIRE(dy, "bg_bias_act_bwd: null dy");
  BG_REQUIRE(dtype_ok(dtype), "bg_bias_act_bwd: dtype %d", dtype);
  BG_REQUIRE(rows > 0 && rows < (1ll << 31) && C > 0 && C % 4 == 0 && dy_row_stride >= C && dy_row_stride % 4 == 0,
             "bg_bias_act_bwd: rows=%lld C=%d stride=%lld (C and the stride must be multiples of 4)", (long long)rows, C, (long long)dy_row_stride);
  BG_REQUIRE(act >= 0 && act <= 2, "bg_bias_act_bwd: act=%d (0 none, 1 gelu-tanh, 2 gelu-erf)", act);
  BG_REQUIRE(act == 0 || (pre && dx && dy_row_stride == C), "bg_bias_act_bwd: an activation needs pre, dx and contiguous dy");
  const uintptr_t am = dtype == BG_F32 ? 15 : 7;
  BG_REQUIRE((((uintptr_t)dy) & am) == 0 && (((uintptr_t)pre) & am) == 0 && (((uintptr_t)dx) & am) == 0, "bg_bias_act_bwd: misaligned pointer");
  const size_t need = bg_bias_act_bwd_workspace_bytes(rows, C);
  if (!workspace || workspace_bytes < need || !al16(workspace)) {
    set_error("bg_bias_act_bwd: workspace %zu < %zu bytes (or not 16-byte aligned)", workspace_bytes, need);
    return BG_ERR_WORKSPACE;
  }
  const int G = bias_groups(rows);
  BiasActArgs a{dy, pre, dx, (float*)workspace, rows, dy_row_stride, C};
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)G, (unsigned)cdiv(C, 1024)), block(256);
#define BG_LAUNCH_BA(T)                                                                             \
  do {                                                                                              \
    if (act == 0) hipLaunchKernelGGL((bias_act_bwd_kernel<T, 0>), grid, block, 0, st, a);           \
    else if (act == 1) hipLaunchKernelGGL((bias_act_bwd_kernel<T, 1>), grid, block, 0, st, a);      \
    else hipLaunchKernelGGL((bias_act_bwd_kernel<T, 2>), grid, block, 0, st, a);                    \
  } while (0)
  if (dtype == BG_F32) BG_LAUNCH_BA(float);
  else BG_LAUNCH_BA(bf16_t);
#undef BG_LAUNCH_BA
  BG_CHECK_LAUNCH("bg_bias_act_bwd");
  if (dbias) {
    const ReduceItem item{(const float*)workspace, dbias, G, C, C};
    launch_reduce(&item, 1, st);
    BG_CHECK_LAUNCH("bg_bias_act_bwd(reduce)");
  }
  return BG_OK;
}

extern "C" int bg_partial_reduce(const bg_reduce_desc* descs, int count, void* stream) {
  BG_REQUIRE(count >= 0 && (descs || count == 0), "bg_partial_reduce: count=%d", count);
  if (count == 0) return BG_OK;
  std::vector<ReduceItem> items((size_t)count);
  for (int i = 0; i < count; ++i) {
    const bg_reduce_desc& d = descs[i];
    BG_REQUIRE(d.partial && d.out && d.groups > 0 && d.n > 0 && d.stride >= d.n, "bg_partial_reduce: descriptor %d: groups=%d n=%d stride=%d", i,
               d.groups, d.n, d.stride);
    BG_REQUIRE((((uintptr_t)d.partial) & 3) == 0 && (((uintptr_t)d.out) & 3) == 0, "bg_partial_reduce: descriptor %d: misaligned pointer", i);
    items[(size_t)i] = ReduceItem{d.partial, d.out, d.groups, d.stride, d.n};
  }
  launch_reduce(items.data(), count, (hipStream_t)stream);
  BG_CHECK_LAUNCH("bg_partial_reduce");
  return BG_OK;
}
