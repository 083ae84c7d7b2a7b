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
#include <mutex>
#include <type_traits>

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

template <typename T, bool HAS_NOISE> struct Tile;      // VEC elements per 16-byte load, U loads in flight per lane
// PF: the next chunk's loads are issued before the current one is filtered (register double buffer)
template <> struct Tile<float, false> { static constexpr int VEC = 4, U = 8; static constexpr bool PF = false; };
template <> struct Tile<float, true> { static constexpr int VEC = 4, U = 4; static constexpr bool PF = false; };
template <> struct Tile<bf16_t, false> { static constexpr int VEC = 8, U = 4; static constexpr bool PF = false; };
template <> struct Tile<bf16_t, true> { static constexpr int VEC = 8, U = 2; static constexpr bool PF = false; };

template <typename T, bool HAS_NOISE>
__global__ __launch_bounds__(256) void topn_partial_kernel(TopnArgs a) {
  constexpr int VEC = Tile<T, HAS_NOISE>::VEC, U = Tile<T, HAS_NOISE>::U;
  constexpr bool PF = Tile<T, HAS_NOISE>::PF;
  constexpr int CH = 256 * VEC * U;     // elements per workgroup chunk; U*VEC <= 32 candidate bits per lane
  __shared__ u64 lds_items[4 * BG_MAX_TOPN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const int split = blockIdx.y;
  const int64_t in_row = a.row_map ? a.row_map[row] : row;
  const T* s = (const T*)a.scores + in_row * a.score_stride;
  const float* e = HAS_NOISE ? a.noise + row * a.noise_stride : nullptr;
  const int64_t begin = min(a.V, (int64_t)split * a.seg_len);
  const int64_t end = min(a.V, begin + a.seg_len);
  const int n = a.n;

  uint32_t blk[BG_MAX_BLOCKED];
#pragma unroll
  for (int i = 0; i < BG_MAX_BLOCKED; ++i) {
    int64_t b = (i < a.n_blocked) ? a.blocked[row * a.n_blocked + i] : -1;
    blk[i] = (b >= 0 && b < a.V) ? (uint32_t)b : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < i; ++k)
      if (blk[k] == blk[i]) blk[i] = 0xFFFFFFFFu;     // a repeated id counts once
  }
  auto is_blocked = [&](int64_t i) -> bool {
    bool hit = false;
#pragma unroll
    for (int b = 0; b < BG_MAX_BLOCKED; ++b) hit |= (b < a.n_blocked) && (blk[b] == (uint32_t)i);
    return hit;
  };
  // exact item of element i (the streaming loop below only filters; whatever passes is re-read here)
  auto item_of = [&](int64_t i) -> u64 {
    if (is_blocked(i)) return 0;                       // blocked elements enter once, in the prologue
    float key = to_f32(s[i]);
    if constexpr (HAS_NOISE) key = key / e[i];
    return make_item(key, (uint32_t)i);
  };

  WaveList wl;
  wl.init(n, lane);

  // prologue: the blocked ids of this segment take their replacement score (wave 0 owns them)
  if (wave == 0) {
#pragma unroll
    for (int b = 0; b < BG_MAX_BLOCKED; ++b) {
      if (b >= a.n_blocked) break;
      const int64_t i = blk[b];
      if (blk[b] == 0xFFFFFFFFu || i < begin || i >= end) continue;    // wave-uniform
      float key = a.blocked_value;
      if constexpr (HAS_NOISE) key = key / e[i];
      wl.offer(make_item(key, (uint32_t)i));
    }
  }

  // per-element path for the unaligned head and the ragged tail
  auto edge = [&](int64_t lo, int64_t hi) {
    for (int64_t base = lo; base < hi; base += 256) {
      const int64_t i = base + tid;
      u64 mine = 0;
      if (i < hi) {
        float key = to_f32(s[i]);
        if constexpr (HAS_NOISE) key = key / e[i];
        if (!(key < wl.tau_key) && !is_blocked(i)) mine = make_item(key, (uint32_t)i);
      }
      wl.offer_lanes(mine);
    }
  };

  const int mis = (int)((((uintptr_t)(s + begin)) & 15) / sizeof(T));
  int64_t abegin = min(end, begin + ((VEC - mis) % VEC));
  if (HAS_NOISE && (((uintptr_t)(e + abegin)) & 15) != 0) abegin = end;   // noise not co-aligned: everything per element
  edge(begin, abegin);

  // ---- streaming body: whole chunks of CH elements, U independent 16-byte loads per lane issued before anything waits ----
  typedef typename std::conditional<VEC == 4, f32x4, bf16x8>::type SV;
  constexpr int NQ = VEC / 4;                       // noise vectors per score vector
  struct Raw {
    SV sc[U];
    f32x4 en[HAS_NOISE ? U * NQ : 1];
  };
  // whole 16-byte vectors end at vend; a vector past it is not loaded and reads as -inf (never a candidate of a full list)
  const int64_t vend = abegin + ((end - abegin) / VEC) * VEC;
  auto load = [&](Raw& r, int64_t at) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i0 = at + (int64_t)(u * 256 + tid) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) r.sc[u][j] = -INFINITY;
      if (i0 < vend) r.sc[u] = __builtin_nontemporal_load((const SV*)(s + i0));
    }
    if constexpr (HAS_NOISE) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int64_t i0 = at + (int64_t)(u * 256 + tid) * VEC;
          r.en[u * NQ + q] = f32x4{1.f, 1.f, 1.f, 1.f};
          if (i0 < vend) r.en[u * NQ + q] = __builtin_nontemporal_load((const f32x4*)(e + i0 + 4 * q));
        }
    }
  };
  int64_t cb = abegin;
  Raw cur, nxt;
  if (PF && cb < vend) load(cur, cb);
  for (; cb < vend; cb += CH) {
    if constexpr (PF) {
      if (cb + CH < vend) load(nxt, cb + CH);       // the next chunk is in flight while this one is filtered
    } else {
      load(cur, cb);
    }
    float key[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        key[u][j] = (float)cur.sc[u][j];
        if constexpr (HAS_NOISE) key[u][j] = key[u][j] / cur.en[u * NQ + (j >> 2)][j & 3];
      }
    auto index_of = [&](int k) -> int64_t { return cb + (int64_t)((k / VEC) * 256 + tid) * VEC + (k % VEC); };
    float thr = wl.tau_key;
    if (cb == abegin) {
      // First chunk: the list is empty (or holds only prologue items).  The n-th largest of the 64 per-lane maxima is a
      // valid lower bound of this chunk's n-th best key (n real elements reach it), so it filters the chunk before
      // any item is built.  v_max drops NaNs, which only makes the bound more conservative.
      float best = key[0][0];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) best = fmaxf(best, key[u][j]);
      // up to n_blocked of those maxima may belong to blocked elements (whose streamed score is not their real one)
      const int nb = n + a.n_blocked;
      if (nb <= 64) {
        // nb-th largest of the 64 lane values = the largest value with at least nb lanes at or above it: bitwise search on the ordered
        // bit pattern, one v_cmp + scalar popcount per bit
        const uint32_t ob = order_bits(best);
        uint32_t sel = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
          const uint32_t cand = sel | (1u << bit);
          if (__popcll(__ballot(ob >= cand)) >= nb) sel = cand;
        }
        const float bound = order_bits_inv(sel);
        if (!(bound < thr)) thr = bound;
      }
    }
    // one v_cmp per element straight into a wave mask (SGPR pair); the masks are OR-ed on the scalar unit and per-lane
    // candidate bits are only built for a 16-byte group in which some lane has a candidate
    uint32_t cmask = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool c[VEC];
      u64 any = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        c[j] = !(key[u][j] < thr);
        any |= __ballot(c[j]);
      }
      if (any) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) cmask |= c[j] ? (1u << (u * VEC + j)) : 0u;
      }
    }
    // drain: one candidate per lane per round, re-read from memory (L1/L2 hit) so no register array is indexed dynamically
    while (__ballot(cmask != 0)) {
      u64 mine = 0;
      if (cmask) {
        const int k = __ffs(cmask) - 1;
        cmask &= cmask - 1;
        const int64_t i = index_of(k);
        if (i < vend) mine = item_of(i);
      }
      wl.offer_lanes(mine);
    }
    if constexpr (PF) cur = nxt;
  }
  edge(vend, end);

  const u64 item = wl.item;
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

// One body for both entry points: `partial` is the caller's workspace (bg_topn_select) or this device's copy of the
// file-scope scratch array (bg_topn_select_static); nothing here allocates or synchronises.
static int topn_launch(const char* who, const void* scores, int dtype, int64_t rows, int64_t V, int64_t score_row_stride,
                       const int64_t* row_map, const float* noise, int64_t noise_row_stride, const int64_t* blocked,
                       int n_blocked, float blocked_value, int n, int64_t* out_ids, float* out_keys, u64* partial,
                       size_t partial_bytes, void* stream) {
  BG_REQUIRE(scores && out_ids, "%s: null scores/out_ids", who);
  BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16, "%s: dtype %d", who, dtype);
  BG_REQUIRE(rows > 0 && rows < (1ll << 31), "%s: rows=%lld", who, (long long)rows);
  BG_REQUIRE(V > 0 && V < (1ll << 31), "%s: V=%lld", who, (long long)V);
  BG_REQUIRE(n >= 1 && n <= BG_MAX_TOPN && n <= V, "%s: n=%d must be in [1, min(%d, V)]", who, n, BG_MAX_TOPN);
  BG_REQUIRE(score_row_stride >= V, "%s: score_row_stride < V", who);
  BG_REQUIRE(!noise || noise_row_stride >= V, "%s: noise_row_stride < V", who);
  BG_REQUIRE(n_blocked >= 0 && n_blocked <= BG_MAX_BLOCKED && (n_blocked == 0 || blocked), "%s: n_blocked=%d", who, n_blocked);
  const int splits = choose_splits(rows, V);
  TopnArgs a;
  a.scores = scores; a.row_map = row_map; a.noise = noise; a.blocked = blocked;
  a.V = V; a.score_stride = score_row_stride; a.noise_stride = noise_row_stride;
  a.seg_len = cdiv(cdiv(V, splits), 2048) * 2048;
  a.n = n; a.n_blocked = n_blocked; a.splits = splits; a.blocked_value = blocked_value;
  a.out_ids = out_ids; a.out_keys = out_keys; a.partial = partial;
  if (splits > 1) {
    size_t need = (size_t)rows * splits * n * sizeof(u64);
    if (!partial || partial_bytes < need) {
      set_error("%s: %s holds %zu bytes, rows=%lld x splits=%d x n=%d partial items need %zu", who,
                partial ? "the partial buffer" : "a NULL partial buffer", partial_bytes, (long long)rows, splits, n, need);
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
  BG_CHECK_LAUNCH(who);
  if (splits > 1) {
    hipLaunchKernelGGL(topn_merge_kernel, dim3((unsigned)rows), block, 0, st, (const u64*)partial, splits, n, out_ids, out_keys);
    BG_CHECK_LAUNCH(who);
  }
  return BG_OK;
}

extern "C" int bg_topn_select(const void* scores, int dtype, int64_t rows, int64_t V, int64_t score_row_stride,
                              const int64_t* row_map, const float* noise, int64_t noise_row_stride,
                              const int64_t* blocked, int n_blocked, float blocked_value, int n, int64_t* out_ids,
                              float* out_keys, void* workspace, size_t workspace_bytes, void* stream) {
  return topn_launch("bg_topn_select", scores, dtype, rows, V, score_row_stride, row_map, noise, noise_row_stride, blocked,
                     n_blocked, blocked_value, n, out_ids, out_keys, (u64*)workspace, workspace_bytes, stream);
}

// ---- the entry point that allocates nothing (include/branchgan.h: bg_topn_select_static) ----
// The per-(row, split) partial lists of topn_partial_kernel live in a file-scope device array instead of a caller
// workspace.  In host code the array's NAME is only a stand-in for the symbol: each GPU holds its own copy at its own
// address, which hipGetSymbolAddress returns once per device (with that device current); host code uses that pointer.
static __device__ __align__(256) u64 scratch[BG_TOPN_STATIC_ITEMS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
u64* g_scratch_ptr[kMaxDevices];     // per device: address of ITS copy of `scratch` (null until the first call on it)
}  // namespace

extern "C" int bg_topn_select_static(const void* scores, int dtype, int64_t rows, int64_t V, int64_t score_row_stride,
                                     const int64_t* row_map, const float* noise, int64_t noise_row_stride,
                                     const int64_t* blocked, int n_blocked, float blocked_value, int n, int64_t* out_ids,
                                     float* out_keys, void* stream) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    set_error("bg_topn_select_static: no current HIP device (or device index %d >= %d)", dev, kMaxDevices);
    return BG_ERR_LAUNCH;
  }
  u64* ptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    ptr = g_scratch_ptr[dev];
    if (!ptr) {
      void* sym = nullptr;
      hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !sym) {
        set_error("bg_topn_select_static: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
        return BG_ERR_LAUNCH;
      }
      ptr = g_scratch_ptr[dev] = (u64*)sym;
    }
  }
  // the partial kernel writes every slot of a (row, split) list it hands to the merge kernel (it clears the list before the
  // rank-merge), so the last call's contents never reach this call's result and no memset is needed
  return topn_launch("bg_topn_select_static", scores, dtype, rows, V, score_row_stride, row_map, noise, noise_row_stride,
                     blocked, n_blocked, blocked_value, n, out_ids, out_keys, ptr,
                     (size_t)BG_TOPN_STATIC_ITEMS * sizeof(u64), stream);
}
