// Per-row rank statistics over the vocabulary - the (rows, V) work of the evaluation metrics, without a sort.
//
// The reference sorts every row of the (B, S, V) probability tensor three times per evaluation batch
// (Training/Metrics/GeneratorMetrics.py:24-26 `_calcRankProbMasses`: full sort, then sums of the r largest for r up to 1024;
// :29-50 `_calculateInputIDsPlacementMetric`: full argsort, a (B, S, V) one-hot of the label, a (B, S, V) arange; :64-66,
// 109-111 label gather, row min / max) and Training/Utils.py:20-21 adds the entropy.  See include/branchgan.h: bg_rank_stats.
//
// One 512-thread workgroup per row.  Pass 1 streams the row from HBM once (16-byte loads) for everything that needs no order
// - label value, the label's position in a stable descending sort (a count), min, max, entropy - and for the first level of
// an MSD radix select on the order-preserving integer image of the floats (11 + 11 + 10 bits): a 2048-bin histogram that
// tells which buckets hold the requested ranks.  Pass 2 (row now in L2 / MALL) compacts the keys of those buckets and
// everything above them into LDS - for next-token distributions little more than the largest requested rank - where they are
// sorted and prefix-summed.  Rows too flat for that (more than 4096 keys in the selected buckets) take two more radix levels
// and a summing pass over the row instead.  Histogram counters are LDS atomics on integers (exact, order-free); every
// floating-point sum runs in a fixed order, so results are bit-reproducible.  HBM-bound: algorithmic bytes = one read of the row.
#include "common.h"

namespace bg {

constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_BINS = 2048;      // 11-bit digits (the last level uses 1024 of them)
constexpr int RS_CAND = 4096;      // keys the fast path sorts in LDS
constexpr int RS_GROUP = 4;        // ranks the general path refines per sweep (LDS: RS_GROUP histograms of 2048 bins = 32 KB,
                                   // four workgroups per CU)

struct RankStatsArgs {
  const float* x;
  const int64_t* labels;
  int64_t V, stride;
  int n_ranks;
  int ranks[BG_MAX_RANKS];
  float eps;
  float *label_value, *row_min, *row_max, *entropy, *top_mass;
  int64_t* placement;
};

// ascending unsigned order == ascending float order (-0 and +0 share a key, as they compare equal; NaNs with the sign bit
// clear sort above +inf)
__device__ __forceinline__ uint32_t order_key(float v) {
  uint32_t u = __float_as_uint(v);
  u = (u == 0x80000000u) ? 0u : u;
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// f(value, index) over the row: aligned 16-byte loads, scalar head / tail (V and the row stride need not be multiples of 4).
// Four vectors per thread are requested before the first is used: at two workgroups per CU a single 16-byte load per thread
// in flight covers a quarter of the bytes HBM latency needs outstanding.
template <typename F>
__device__ __forceinline__ void row_foreach(const float* x, int64_t V, F f) {
  constexpr int UN = 4;
  const int mis = (int)((((uintptr_t)x) & 15) / sizeof(float));
  for (int64_t cb = -mis; cb < V; cb += (int64_t)UN * RS_THREADS * 4) {
    f32x4 raw[UN];
    bool whole[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i0 = cb + ((int64_t)u * RS_THREADS + threadIdx.x) * 4;
      whole[u] = i0 >= 0 && i0 + 4 <= V;
      if (whole[u]) raw[u] = *(const f32x4*)(x + i0);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i0 = cb + ((int64_t)u * RS_THREADS + threadIdx.x) * 4;
      if (whole[u]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f(raw[u][j], i0 + j);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (i0 + j >= 0 && i0 + j < V) f(x[i0 + j], i0 + j);
      }
    }
  }
}

// The same walk with every lane of the wave calling f together (valid = false outside the row): for bodies that use
// wave-wide operations (ballots) or wave-uniform state.
template <typename F>
__device__ __forceinline__ void row_foreach_converged(const float* x, int64_t V, F f) {
  constexpr int UN = 4;
  const int mis = (int)((((uintptr_t)x) & 15) / sizeof(float));
  for (int64_t cb = -mis; cb < V; cb += (int64_t)UN * RS_THREADS * 4) {
    float v[UN][4];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i0 = cb + ((int64_t)u * RS_THREADS + threadIdx.x) * 4;
      if (i0 >= 0 && i0 + 4 <= V) {
        const f32x4 raw = *(const f32x4*)(x + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[u][j] = raw[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[u][j] = (i0 + j >= 0 && i0 + j < V) ? x[i0 + j] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i0 = cb + ((int64_t)u * RS_THREADS + threadIdx.x) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) f(v[u][j], i0 + j >= 0 && i0 + j < V);
    }
  }
}

// fixed-order block reductions: lanes by xor-shuffle, then the 8 wave results in wave order; result in every thread
template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, T* scratch, Op op) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = scratch[0];
#pragma unroll
  for (int w = 1; w < RS_WAVES; ++w) r = op(r, scratch[w]);
  return r;
}

// The bin holding the r-th largest element of a histogram (r >= 1, r <= its total) and the count in the bins above it.
// Every thread owns `PER` consecutive bins; inclusive prefix sums over threads give each thread the count above its bins.
template <int PER>
__device__ __forceinline__ void find_rank(const uint32_t* hist, uint32_t r, uint32_t* wave_tot, uint32_t* out /* [2] */) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t h[PER], tot = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) { h[j] = hist[t * PER + j]; tot += h[j]; }
  uint32_t inc = tot;                                   // inclusive prefix over lanes, then over waves
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) {
    if (w < wave) before += wave_tot[w];
    total += wave_tot[w];
  }
  uint32_t above = total - (before + inc);              // elements in bins of higher threads
#pragma unroll
  for (int j = PER - 1; j >= 0; --j) {
    if (above < r && r <= above + h[j]) { out[0] = (uint32_t)(t * PER + j); out[1] = above; }
    above += h[j];
  }
  __syncthreads();
}

// Bitonic sort (descending) of 2048 keys held four per thread, element index e = 4 t + r: partners four or more apart sit in
// another lane (strides 4 .. 128: a wave shuffle) or another wave (strides 256 .. 1024: through `xch`, 2048 words of LDS,
// two barriers); strides 1 and 2 stay in the thread's own registers.  6 of the 66 stages touch LDS.
__device__ __forceinline__ void sort2048_desc(uint32_t (&k)[4], uint32_t* xch) {
  const uint32_t t = threadIdx.x;
#pragma unroll
  for (uint32_t size = 2; size <= 2048; size <<= 1) {
#pragma unroll
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      uint32_t other[4];
      if (stride >= 256) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[4 * t + r] = k[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) other[r] = xch[4 * (t ^ (stride >> 2)) + r];
      } else if (stride >= 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) other[r] = (uint32_t)__shfl_xor((int)k[r], (int)(stride >> 2));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) other[r] = k[r ^ (int)stride];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t e = 4 * t + r;
        const bool take_max = ((e & stride) == 0) == ((e & size) == 0);
        const uint32_t hi = k[r] > other[r] ? k[r] : other[r], lo = k[r] > other[r] ? other[r] : k[r];
        k[r] = take_max ? hi : lo;
      }
    }
  }
}

// find_rank for several ranks of ONE histogram: one scan, every thread tests all ranks against its bins.
template <int PER>
__device__ __forceinline__ void find_ranks(const uint32_t* hist, const uint32_t (&r)[BG_MAX_RANKS], int nr, uint32_t* wave_tot, uint32_t* out) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t h[PER], tot = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) { h[j] = hist[t * PER + j]; tot += h[j]; }
  uint32_t inc = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) {
    if (w < wave) before += wave_tot[w];
    total += wave_tot[w];
  }
  uint32_t above = total - (before + inc);
#pragma unroll
  for (int j = PER - 1; j >= 0; --j) {
#pragma unroll
    for (int i = 0; i < BG_MAX_RANKS; ++i)
      if (i < nr && above < r[i] && r[i] <= above + h[j]) { out[2 * i] = (uint32_t)(t * PER + j); out[2 * i + 1] = above; }
    above += h[j];
  }
  __syncthreads();
}

__global__ __launch_bounds__(RS_THREADS) void rank_stats_kernel(RankStatsArgs a) {
  // [0, 8192): histograms (level 1: 4 lane-interleaved copies of 2048 bins; levels 2, 3: one histogram per rank of a group)
  __shared__ uint32_t lds[RS_GROUP * RS_BINS + 64];
  uint32_t* hist = lds;
  uint32_t* scratch = lds + RS_GROUP * RS_BINS;     // 8 words of reduction scratch, 8 of wave totals, 2 per rank of results
  uint32_t* wave_tot = scratch + 8;
  uint32_t* sel = scratch + 16;                         // sel[2 i] = bin, sel[2 i + 1] = count above, for rank i

  const int t = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float* x = a.x + row * a.stride;
  const int64_t V = a.V;
  const int64_t label = a.labels ? a.labels[row] : -1;
  const bool has_label = label >= 0 && label < V;
  const float xl = has_label ? x[label] : 0.f;
  const uint32_t kl = order_key(xl);
  const int nr = a.n_ranks;

  for (int i = t; i < 4 * RS_BINS; i += RS_THREADS) hist[i] = 0;
  __syncthreads();

  // ---- pass 1 ---------------------------------------------------------------------------------------------------------
  float mn = INFINITY, mx = -INFINITY, ent = 0.f;
  uint32_t n_gt = 0, n_eq_before = 0;
  uint32_t* my_hist = hist + (t & 3) * RS_BINS;      // probabilities crowd into a few exponent bins: spread the same-address atomics
  const float eps = a.eps;
  row_foreach(x, V, [&](float v, int64_t j) {
    const uint32_t k = order_key(v);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    ent += v * __logf(v + eps);            // v_log_f32: 1 ulp, far inside the metric's tolerance
    n_gt += k > kl;
    n_eq_before += (k == kl) & (j < label);
    if (nr > 0) atomicAdd(&my_hist[k >> 21], 1u);
  });
  mn = block_reduce(mn, (float*)scratch, [](float p, float q) { return fminf(p, q); });
  mx = block_reduce(mx, (float*)scratch, [](float p, float q) { return fmaxf(p, q); });
  ent = block_reduce(ent, (float*)scratch, [](float p, float q) { return p + q; });
  n_gt = block_reduce(n_gt, scratch, [](uint32_t p, uint32_t q) { return p + q; });
  n_eq_before = block_reduce(n_eq_before, scratch, [](uint32_t p, uint32_t q) { return p + q; });
  if (t == 0) {
    if (a.label_value) a.label_value[row] = has_label ? xl : __uint_as_float(0x7FC00000u);
    if (a.placement) a.placement[row] = has_label ? (int64_t)n_gt + (int64_t)n_eq_before : -1;
    if (a.row_min) a.row_min[row] = mn;
    if (a.row_max) a.row_max[row] = mx;
    if (a.entropy) a.entropy[row] = -ent;
  }
  if (nr == 0) return;

  // ---- radix select of the r-th largest value for every requested rank --------------------------------------------------
  __syncthreads();
  for (int i = t; i < RS_BINS; i += RS_THREADS) {
    uint32_t c = hist[i];
#pragma unroll
    for (int k = 1; k < 4; ++k) c += hist[k * RS_BINS + i];
    hist[i] = c;
  }
  __syncthreads();
  uint32_t want[BG_MAX_RANKS], prefix[BG_MAX_RANKS], above[BG_MAX_RANKS];      // uniform across the block
#pragma unroll
  for (int i = 0; i < BG_MAX_RANKS; ++i) want[i] = i < nr ? (uint32_t)(a.ranks[i] < V ? a.ranks[i] : V) : 0u;
  find_ranks<RS_BINS / RS_THREADS>(hist, want, nr, wave_tot, sel);
#pragma unroll
  for (int i = 0; i < BG_MAX_RANKS; ++i) {
    if (i < nr) { prefix[i] = sel[2 * i]; above[i] = sel[2 * i + 1]; }
  }
  __syncthreads();

  // ---- fast path: everything a rank can need lies in the buckets at and above the lowest selected one.  When those hold at
  // most RS_CAND keys (for next-token distributions: the requested ranks plus a few hundred), ONE more pass over the row
  // compacts them into LDS, a bitonic sort orders them, and each top mass is a prefix sum of the sorted keys.
  uint32_t b_min = prefix[0], n_cand = 0;
#pragma unroll
  for (int i = 0; i < BG_MAX_RANKS; ++i) {
    if (i < nr) {
      b_min = prefix[i] < b_min ? prefix[i] : b_min;
      const uint32_t c = above[i] + hist[prefix[i]];
      n_cand = c > n_cand ? c : n_cand;
    }
  }
  __syncthreads();                                      // hist[] was read above and is overwritten below
  if (n_cand <= RS_CAND) {
    // Compaction without atomics: wave w appends to its own region [w * RS_CAND / 8, ...) at a wave-uniform count kept in a
    // register (ballot + popcount), then the eight regions are packed into one run.  A wave that meets more than its region
    // holds sends the row down the general path.
    constexpr uint32_t REGION = RS_CAND / RS_WAVES;
    const int lane = t & 63, wave = t >> 6;
    uint32_t* region = lds + wave * REGION;
    uint32_t wcount = 0;
    row_foreach_converged(x, V, [&](float v, bool valid) {
      const uint32_t k = order_key(v);
      const bool take = valid && (k >> 21) >= b_min;
      const uint64_t m = __ballot(take);
      const uint32_t at = wcount + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (take && at < REGION) region[at] = k;
      wcount += (uint32_t)__popcll(m);
    });
    if (lane == 0) wave_tot[wave] = wcount;
    __syncthreads();
    uint32_t start = 0, worst = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      if (w < wave) start += wave_tot[w];
      worst = wave_tot[w] > worst ? wave_tot[w] : worst;
    }
    if (worst <= REGION) {
      uint32_t* cand = lds + RS_CAND;                     // [0, n_pad) keys, packed behind the regions
      uint32_t n_pad = 64;
      while (n_pad < n_cand) n_pad <<= 1;
      for (uint32_t i = t; i < n_pad; i += RS_THREADS) cand[i] = 0u;                // key 0 sorts below every real value
      __syncthreads();
      for (uint32_t j = (uint32_t)lane; j < wcount; j += 64) cand[start + j] = region[j];
      __syncthreads();
      if (n_pad <= 2048) {                                // the usual case: sort in registers, four keys per thread
        uint32_t k4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) k4[r] = 4 * (uint32_t)t + r < n_pad ? cand[4 * t + r] : 0u;
        sort2048_desc(k4, lds);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) cand[4 * t + r] = k4[r];
        n_pad = 2048;
        __syncthreads();
      } else {
        // Bitonic sort, descending, in LDS.  Every wave owns a contiguous n_pad / 8 keys: the stages whose partners lie inside one such
        // chunk (all of them up to that size, and the tail of every later size) run without a workgroup barrier - LDS operations
        // of one wave execute in order - which leaves 15 barriers for 4096 keys instead of 78.
        {
          volatile uint32_t* vc = cand;
          const uint32_t chunk = n_pad / RS_WAVES, wbase = (uint32_t)(t >> 6) * chunk;
          auto exchange = [&](uint32_t lo, uint32_t hi, uint32_t size) {
            const uint32_t p = vc[lo], q = vc[hi];
            const bool desc = (lo & size) == 0;
            if (desc ? p < q : p > q) { vc[lo] = q; vc[hi] = p; }
          };
          for (uint32_t size = 2; size <= n_pad; size <<= 1) {
            uint32_t stride = size >> 1;
            for (; 2 * stride > chunk; stride >>= 1) {                                   // partners in different chunks
              for (uint32_t i = t; i < (n_pad >> 1); i += RS_THREADS) {
                const uint32_t lo = 2 * i - (i & (stride - 1));
                exchange(lo, lo + stride, size);
              }
              __syncthreads();
            }
            for (; stride > 0; stride >>= 1) {                                           // inside the wave's own chunk
              for (uint32_t pi = (uint32_t)lane; pi < (chunk >> 1); pi += 64) {
                const uint32_t lo = wbase + 2 * pi - (pi & (stride - 1));
                exchange(lo, lo + stride, size);
              }
              __builtin_amdgcn_wave_barrier();
            }
            if (2 * size > chunk) __syncthreads();                                       // the next size reaches across chunks
          }
          __syncthreads();
        }
      }
      // top masses: thread t sums cand[t * per, (t + 1) * per) below each rank, lanes by shuffle, waves in order
      const uint32_t per = (n_pad + RS_THREADS - 1) / RS_THREADS;
      float* part = (float*)lds;                                                        // [wave][rank]; the regions are free again
  #pragma unroll
      for (int i = 0; i < BG_MAX_RANKS; ++i) {
        if (i < nr) {
          float sm = 0.f;
          for (uint32_t j = t * per; j < (t + 1) * per && j < want[i]; ++j) sm += key_value(cand[j]);
  #pragma unroll
          for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
          if (lane == 0) part[(t >> 6) * BG_MAX_RANKS + i] = sm;
        }
      }
      __syncthreads();
      if (t < nr) {
        float sm = part[t];
  #pragma unroll
        for (int w = 1; w < RS_WAVES; ++w) sm += part[w * BG_MAX_RANKS + t];
        a.top_mass[row * nr + t] = sm;
      }
      return;
    }
    __syncthreads();                                    // a region overflowed: the general path reuses the histogram words
  }

  // ---- general path (flat rows: more than RS_CAND keys share the selected buckets): two more radix levels and a summing pass
  // over the row, RS_GROUP ranks at a time
#pragma unroll
  for (int g0 = 0; g0 < BG_MAX_RANKS; g0 += RS_GROUP) {
    if (g0 >= nr) break;
    // level 2: the next 11 bits among the elements whose top 11 bits equal the rank's bucket
    __syncthreads();
    for (int i = t; i < RS_GROUP * RS_BINS; i += RS_THREADS) hist[i] = 0;
    __syncthreads();
    row_foreach(x, V, [&](float v, int64_t) {
      const uint32_t k = order_key(v);
#pragma unroll
      for (int ii = 0; ii < RS_GROUP; ++ii)
        if (g0 + ii < nr && (k >> 21) == prefix[g0 + ii]) atomicAdd(&hist[ii * RS_BINS + ((k >> 10) & 0x7FFu)], 1u);
    });
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < RS_GROUP; ++ii)
      if (g0 + ii < nr) find_rank<RS_BINS / RS_THREADS>(hist + ii * RS_BINS, want[g0 + ii] - above[g0 + ii], wave_tot, sel + 2 * ii);
#pragma unroll
    for (int ii = 0; ii < RS_GROUP; ++ii) {
      if (g0 + ii < nr) { prefix[g0 + ii] = (prefix[g0 + ii] << 11) | sel[2 * ii]; above[g0 + ii] += sel[2 * ii + 1]; }
    }
    __syncthreads();

    // level 3: the last 10 bits
    for (int i = t; i < RS_GROUP * RS_BINS; i += RS_THREADS) hist[i] = 0;
    __syncthreads();
    row_foreach(x, V, [&](float v, int64_t) {
      const uint32_t k = order_key(v);
#pragma unroll
      for (int ii = 0; ii < RS_GROUP; ++ii)
        if (g0 + ii < nr && (k >> 10) == prefix[g0 + ii]) atomicAdd(&hist[ii * RS_BINS + (k & 0x3FFu)], 1u);
    });
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < RS_GROUP; ++ii)
      if (g0 + ii < nr) find_rank<1024 / RS_THREADS>(hist + ii * RS_BINS, want[g0 + ii] - above[g0 + ii], wave_tot, sel + 2 * ii);
#pragma unroll
    for (int ii = 0; ii < RS_GROUP; ++ii) {                       // prefix = key of the r-th largest
      if (g0 + ii < nr) { prefix[g0 + ii] = (prefix[g0 + ii] << 10) | sel[2 * ii]; above[g0 + ii] += sel[2 * ii + 1]; }
    }
  }

  // ---- last pass: the mass above each threshold; the threshold value itself fills the remaining (r - above) places ----------
  float acc[BG_MAX_RANKS];
#pragma unroll
  for (int i = 0; i < BG_MAX_RANKS; ++i) acc[i] = 0.f;
  row_foreach(x, V, [&](float v, int64_t) {
    const uint32_t k = order_key(v);
#pragma unroll
    for (int i = 0; i < BG_MAX_RANKS; ++i)
      if (i < nr && k > prefix[i]) acc[i] += v;
  });
#pragma unroll
  for (int i = 0; i < BG_MAX_RANKS; ++i) {
    if (i < nr) {
      const float s = block_reduce(acc[i], (float*)scratch, [](float p, float q) { return p + q; });
      if (t == 0) a.top_mass[row * nr + i] = s + (float)(want[i] - above[i]) * key_value(prefix[i]);
    }
  }
}

}  // namespace bg

using namespace bg;

extern "C" int bg_rank_stats(const float* x, int64_t rows, int64_t V, int64_t row_stride, const int64_t* labels, const int32_t* ranks,
                             int n_ranks, float entropy_eps, float* label_value, int64_t* placement, float* row_min, float* row_max,
                             float* entropy, float* top_mass, void* stream) {
  BG_REQUIRE(x, "bg_rank_stats: null x");
  BG_REQUIRE(rows > 0 && rows < (1ll << 31) && V > 0 && V < (1ll << 31) && row_stride >= V, "bg_rank_stats: rows=%lld V=%lld stride=%lld",
             (long long)rows, (long long)V, (long long)row_stride);
  BG_REQUIRE((((uintptr_t)x) & 3) == 0, "bg_rank_stats: misaligned x");
  BG_REQUIRE(n_ranks >= 0 && n_ranks <= BG_MAX_RANKS && (n_ranks == 0 || (ranks && top_mass)), "bg_rank_stats: n_ranks=%d (at most %d, with ranks and top_mass)",
             n_ranks, BG_MAX_RANKS);
  BG_REQUIRE(!placement || labels, "bg_rank_stats: placement without labels");
  RankStatsArgs a{};
  a.x = x; a.labels = labels; a.V = V; a.stride = row_stride; a.n_ranks = n_ranks; a.eps = entropy_eps;
  for (int i = 0; i < n_ranks; ++i) {
    BG_REQUIRE(ranks[i] >= 1, "bg_rank_stats: ranks[%d]=%d must be at least 1", i, ranks[i]);
    a.ranks[i] = ranks[i];
  }
  a.label_value = label_value; a.placement = placement; a.row_min = row_min; a.row_max = row_max; a.entropy = entropy; a.top_mass = top_mass;
  hipLaunchKernelGGL(rank_stats_kernel, dim3((unsigned)rows), dim3(RS_THREADS), 0, (hipStream_t)stream, a);
  BG_CHECK_LAUNCH("bg_rank_stats");
  return BG_OK;
}
