// Top-n selection over the vocabulary: one streaming pass per row (HBM-bound).
//
// Replaces the reference's full argsort per row (BranchPredictions/SamplingStrategies.py:149-158,197-205,
// 236-240) and its scatter+cat label blocking (:258-265).  See include/branchgan.h: bg_topn_select.
//
// Design (MI355X): one 256-thread workgroup per (row, split).  Each lane streams 16-byte vectors of the
// row; each wave keeps its current best n items as a sorted list spread over its lanes (lane i = i-th
// best) and a wave-uniform threshold = the n-th best key.  An element enters the slow path only if it is
// not below the threshold (≈ n·ln(V/n) times per wave); everything else costs one compare.  Items are
// 64-bit composites (order-preserving key bits << 32 | ~index) so one integer compare gives the total
// order "larger key first, NaN largest, equal keys by increasing index".  Waves merge through LDS by rank.
#include "topn_common.h"

namespace bg {

struct TopnArgs {
  const void* scores;
  const int64_t* row_map;
  const float* noise;
  const int64_t* blocked;
  int64_t V, score_stride, noise_stride, seg_len;
  int n, n_blocked, splits;
  float blocked_value;
  int64_t* out_ids;
  float* out_keys;
  u64* partial;  // [rows][splits][n] when splits > 1
};

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };

template <typename T, bool HAS_NOISE>
__global__ __launch_bounds__(256) void topn_partial_kernel(TopnArgs a) {
  constexpr int VEC = VecOf<T>::N;
  __shared__ u64 lds_items[4 * BG_MAX_TOPN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const int split = blockIdx.y;
  const int64_t in_row = a.row_map ? a.row_map[row] : row;
  const T* s = (const T*)a.scores + in_row * a.score_stride;
  const float* e = HAS_NOISE ? a.noise + row * a.noise_stride : nullptr;
  const int64_t begin = (int64_t)split * a.seg_len;
  const int64_t end = min(a.V, begin + a.seg_len);
  const int n = a.n;

  uint32_t blk[BG_MAX_BLOCKED];
#pragma unroll
  for (int i = 0; i < BG_MAX_BLOCKED; ++i) {
    int64_t b = (i < a.n_blocked) ? a.blocked[row * a.n_blocked + i] : -1;
    blk[i] = (b >= 0 && b < a.V) ? (uint32_t)b : 0xFFFFFFFFu;
  }

  u64 item = 0, tau = 0;        // 0 = empty (below every real item)
  float tau_key = order_bits_inv(0);  // NaN: "key < tau_key" is false for every key => everything is a candidate

  // first 16-byte-aligned vector at or before `begin`
  const int mis = (int)((((uintptr_t)(s + begin)) & 15) / sizeof(T));
  const int64_t vbegin = begin - mis;
  const bool noise_vec_ok = HAS_NOISE ? ((((uintptr_t)(e + vbegin)) & 15) == 0) : true;

  for (int64_t cb = vbegin; cb < end; cb += 256 * VEC) {
    const int64_t i0 = cb + (int64_t)tid * VEC;
    float key[VEC];
    bool cand[VEC];
    const bool full = (i0 >= begin) && (i0 + VEC <= end);
    if (full) {
      float sc[VEC];
      if constexpr (VEC == 4) {
        f32x4 v = *(const f32x4*)(s + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] = v[j];
      } else {
        bf16x8 v = *(const bf16x8*)(s + i0);
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = (float)v[j];
      }
      for (int b = 0; b < a.n_blocked; ++b) {
        uint32_t d = blk[b] - (uint32_t)i0;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (d == (uint32_t)j) sc[j] = a.blocked_value;
      }
      if constexpr (HAS_NOISE) {
        float en[VEC];
        if (noise_vec_ok) {
#pragma unroll
          for (int q = 0; q < VEC / 4; ++q) {
            f32x4 v = *(const f32x4*)(e + i0 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) en[4 * q + j] = v[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) en[j] = e[i0 + j];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) key[j] = sc[j] / en[j];
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) key[j] = sc[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) cand[j] = !(key[j] < tau_key);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int64_t i = i0 + j;
        const bool ok = (i >= begin) && (i < end);
        float sc = 0.f;
        if (ok) {
          sc = to_f32(s[i]);
          for (int b = 0; b < a.n_blocked; ++b)
            if (blk[b] == (uint32_t)i) sc = a.blocked_value;
          if constexpr (HAS_NOISE) sc = sc / e[i];
        }
        key[j] = sc;
        cand[j] = ok && !(sc < tau_key);
      }
    }
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) any |= cand[j];
    if (__ballot(any) == 0) continue;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      u64 bal = __ballot(cand[j]);
      if (bal == 0) continue;
      const u64 mine = make_item(key[j], (uint32_t)(i0 + j));
      while (bal) {
        const int L = __ffsll((unsigned long long)bal) - 1;
        bal &= bal - 1;
        const u64 c = readlane64(mine, L);
        if (c > tau) {
          const int pos = __popcll(__ballot(item > c));  // sorted list: lanes 0..pos-1 beat c, pos < n
          const u64 up = __shfl_up(item, 1);
          if (lane == pos) item = c;
          else if (lane > pos) item = up;
          if (lane >= n) item = 0;
          tau = readlane64(item, n - 1);
          tau_key = order_bits_inv((uint32_t)(tau >> 32));
          if (tau == 0) tau_key = order_bits_inv(0);
        }
      }
    }
  }

  if (lane < n) lds_items[wave * n + lane] = item;
  __syncthreads();
  const int64_t out_row = row;
  if (a.splits > 1) {
    u64* part = a.partial + (out_row * a.splits + split) * n;
    // pre-clear: a segment can hold fewer than n real items
    if (tid < n) part[tid] = 0;
    __syncthreads();
    rank_merge_store(lds_items, 4 * n, n, nullptr, nullptr, part);
  } else {
    rank_merge_store(lds_items, 4 * n, n, a.out_ids + out_row * n, a.out_keys ? a.out_keys + out_row * n : nullptr, nullptr);
  }
}

__global__ __launch_bounds__(256) void topn_merge_kernel(const u64* partial, int splits, int n, int64_t* out_ids, float* out_keys) {
  __shared__ u64 items[16 * BG_MAX_TOPN];
  const int64_t row = blockIdx.x;
  const int count = splits * n;
  for (int t = threadIdx.x; t < count; t += blockDim.x) items[t] = partial[row * count + t];
  __syncthreads();
  rank_merge_store(items, count, n, out_ids + row * n, out_keys ? out_keys + row * n : nullptr, nullptr);
}

static int choose_splits(int64_t rows, int64_t V) {
  int64_t s = cdiv(2048, rows);
  int64_t by_len = V / 4096;
  if (s > by_len) s = by_len;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace bg

using namespace bg;

extern "C" size_t bg_topn_workspace_bytes(int64_t rows, int64_t V, int n) {
  if (rows <= 0 || V <= 0 || n <= 0) return 0;
  int splits = choose_splits(rows, V);
  return splits > 1 ? (size_t)rows * splits * n * sizeof(u64) : 0;
}

extern "C" int bg_topn_select(const void* scores, int dtype, int64_t rows, int64_t V, int64_t score_row_stride,
                              const int64_t* row_map, const float* noise, int64_t noise_row_stride,
                              const int64_t* blocked, int n_blocked, float blocked_value, int n, int64_t* out_ids,
                              float* out_keys, void* workspace, size_t workspace_bytes, void* stream) {
  BG_REQUIRE(scores && out_ids, "bg_topn_select: null scores/out_ids");
  BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16, "bg_topn_select: dtype %d", dtype);
  BG_REQUIRE(rows > 0 && rows < (1ll << 31), "bg_topn_select: rows=%lld", (long long)rows);
  BG_REQUIRE(V > 0 && V < (1ll << 31), "bg_topn_select: V=%lld", (long long)V);
  BG_REQUIRE(n >= 1 && n <= BG_MAX_TOPN && n <= V, "bg_topn_select: n=%d must be in [1, min(%d, V)]", n, BG_MAX_TOPN);
  BG_REQUIRE(score_row_stride >= V, "bg_topn_select: score_row_stride < V");
  BG_REQUIRE(!noise || noise_row_stride >= V, "bg_topn_select: noise_row_stride < V");
  BG_REQUIRE(n_blocked >= 0 && n_blocked <= BG_MAX_BLOCKED && (n_blocked == 0 || blocked), "bg_topn_select: n_blocked=%d", n_blocked);
  const int splits = choose_splits(rows, V);
  TopnArgs a;
  a.scores = scores; a.row_map = row_map; a.noise = noise; a.blocked = blocked;
  a.V = V; a.score_stride = score_row_stride; a.noise_stride = noise_row_stride;
  a.seg_len = cdiv(cdiv(V, splits), 2048) * 2048;
  a.n = n; a.n_blocked = n_blocked; a.splits = splits; a.blocked_value = blocked_value;
  a.out_ids = out_ids; a.out_keys = out_keys; a.partial = (u64*)workspace;
  if (splits > 1) {
    size_t need = (size_t)rows * splits * n * sizeof(u64);
    if (!workspace || workspace_bytes < need) {
      set_error("bg_topn_select: workspace %zu < %zu bytes", workspace_bytes, need);
      return BG_ERR_WORKSPACE;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)rows, (unsigned)splits), block(256);
  if (dtype == BG_F32) {
    if (noise) hipLaunchKernelGGL((topn_partial_kernel<float, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((topn_partial_kernel<float, false>), grid, block, 0, st, a);
  } else {
    if (noise) hipLaunchKernelGGL((topn_partial_kernel<bf16_t, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((topn_partial_kernel<bf16_t, false>), grid, block, 0, st, a);
  }
  BG_CHECK_LAUNCH("bg_topn_select");
  if (splits > 1) {
    hipLaunchKernelGGL(topn_merge_kernel, dim3((unsigned)rows), block, 0, st, (const u64*)workspace, splits, n, out_ids, out_keys);
    BG_CHECK_LAUNCH("bg_topn_select(merge)");
  }
  return BG_OK;
}
