// Top-p (nucleus) filter + exponential race, without sorting.
//
// Replaces BranchPredictions/SamplingStrategies.py:161-187 (top_p_sampling: sort, cumsum, shifted mask, argsort of the
// sort indices, gather, renormalise, then the p/e race with another full argsort).  See include/branchgan.h: bg_topp_select.
//
// The reference keeps sorted position j iff cumsum[j+1] <= p, i.e. every element ranked strictly better than e*, the LAST
// element whose prefix sum still fits (rank = value descending, index ascending).  e* is found by a 3-level radix select over
// the float bit pattern (11 + 11 + 10 bits) whose histograms carry a count and a double-precision sum per bucket, exactly the
// accumulation type ATen's CPU cumsum uses (double accumulate, fp32 compare).  Ties at the cut are resolved in index order.
// The final pass streams the row once more with the noise: key = (kept ? p : 0) / kept_mass / e, top-n as in topn.hip.
// One 256-thread workgroup per row; the re-reads of the 200 KB row come from L2 / Infinity Cache, not HBM.
#include <stdlib.h>

#include "topn_common.h"

namespace bg {

struct ToppArgs {
  const float* probs;
  const float* noise;
  const int64_t* blocked;
  int64_t V, stride, noise_stride;
  int n, n_blocked;
  float top_p;
  int64_t* out_ids;
  float* out_keys;
};

struct Cross {       // result of one histogram scan (wave 0 -> LDS -> everybody)
  int found, bucket;
  unsigned cnt_before;
  double cum_before;
};

__device__ __forceinline__ double shfl_up_d(double v, int d) { return __shfl_up(v, d); }

// Scans `nb` buckets from the highest downwards, starting from (cum0, cnt0); finds the first bucket whose inclusion makes
// the fp32-rounded prefix sum exceed p.  Executed by wave 0; nb is a multiple of 64.
__device__ void scan_buckets(const unsigned* cnt, const double* sum, int nb, double cum0, unsigned cnt0, float p32, Cross* out, int lane, bool force) {
  const int per = nb / 64;
  double s = 0.0;
  unsigned c = 0;
  for (int j = 0; j < per; ++j) {
    const int b = nb - 1 - (lane * per + j);
    s += sum[b];
    c += cnt[b];
  }
  double ps = s;
  unsigned pc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double os = shfl_up_d(ps, d);
    const unsigned oc = __shfl_up(pc, d);
    if (lane >= d) { ps += os; pc += oc; }
  }
  double cum = cum0 + (ps - s);
  unsigned cn = cnt0 + (pc - c);
  int found = 0, fb = 0, lb = -1;
  double fcum = 0.0, lcum = 0.0;
  unsigned fcnt = 0, lcnt = 0;
  for (int j = 0; j < per && !found; ++j) {
    const int b = nb - 1 - (lane * per + j);
    if (cnt[b] == 0) continue;
    lb = b; lcum = cum; lcnt = cn;                      // last non-empty bucket seen and the state before it
    const double nxt = cum + sum[b];
    if ((float)nxt > p32) { found = 1; fb = b; fcum = cum; fcnt = cn; }
    else { cum = nxt; cn += cnt[b]; }
  }
  const unsigned long long bal = __ballot(found);
  if (bal == 0) {
    if (force) {   // a deeper level must cross (its parent bucket did); a last-bit rounding difference of the re-summed
                   // bucket must not lose it: cross at the smallest non-empty bucket
      const unsigned long long ne = __ballot(lb >= 0);
      const int last = 63 - __clzll(ne);
      if (lane == last) { out->found = 1; out->bucket = lb; out->cnt_before = lcnt; out->cum_before = lcum; }
    } else if (lane == 63) { out->found = 0; out->bucket = -1; out->cnt_before = cn; out->cum_before = cum; }
  } else {
    const int first = __ffsll(bal) - 1;
    if (lane == first) { out->found = 1; out->bucket = fb; out->cnt_before = fcnt; out->cum_before = fcum; }
  }
}

__global__ __launch_bounds__(256) void topp_kernel(ToppArgs a) {
  __shared__ unsigned h_cnt[2048];
  __shared__ double h_sum[2048];
  __shared__ u64 lds_items[4 * BG_MAX_TOPN];
  __shared__ Cross cross;
  __shared__ unsigned red_u[4];
  __shared__ unsigned sh_theta, sh_ntheta;
  __shared__ long long sh_idx_cut;
  __shared__ float sh_mass;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const float* p = a.probs + row * a.stride;
  const float* e = a.noise + row * a.noise_stride;
  const int V = (int)a.V, n = a.n;
  const float p32 = a.top_p;

  uint32_t blk[BG_MAX_BLOCKED];
#pragma unroll
  for (int i = 0; i < BG_MAX_BLOCKED; ++i) {
    int64_t b = (i < a.n_blocked) ? a.blocked[row * a.n_blocked + i] : -1;
    blk[i] = (b >= 0 && b < a.V) ? (uint32_t)b : 0xFFFFFFFFu;
  }
  auto value = [&](int i) -> float {
    float v = p[i];
    for (int b = 0; b < a.n_blocked; ++b)
      if (blk[b] == (uint32_t)i) v = 0.f;
    return v + 0.0f;   // -0 -> +0
  };
  auto clear_hist = [&](int nb) {
    for (int i = tid; i < nb; i += 256) { h_cnt[i] = 0; h_sum[i] = 0.0; }
    __syncthreads();
  };

  // ---- level 1: bits[31:21] -------------------------------------------------------------------------------------------
  clear_hist(2048);
  for (int i = tid; i < V; i += 256) {
    const float v = value(i);
    const unsigned b = __float_as_uint(v) >> 21;
    atomicAdd(&h_cnt[b], 1u);
    atomicAdd(&h_sum[b], (double)v);
  }
  __syncthreads();
  if (wave == 0) scan_buckets(h_cnt, h_sum, 2048, 0.0, 0u, p32, &cross, lane, false);
  __syncthreads();
  bool crossing = cross.found != 0;
  unsigned prefix = 0;          // bit pattern of the crossing value, refined level by level
  double cum_gt = cross.cum_before;
  unsigned cnt_gt = cross.cnt_before;
  unsigned n_eq = 0;
  if (crossing) {
    const unsigned b1 = (unsigned)cross.bucket;
    // ---- level 2: bits[20:10] inside bucket b1 ----------------------------------------------------------------------------
    clear_hist(2048);
    for (int i = tid; i < V; i += 256) {
      const float v = value(i);
      const unsigned u = __float_as_uint(v);
      if ((u >> 21) == b1) { atomicAdd(&h_cnt[(u >> 10) & 0x7FFu], 1u); atomicAdd(&h_sum[(u >> 10) & 0x7FFu], (double)v); }
    }
    __syncthreads();
    if (wave == 0) scan_buckets(h_cnt, h_sum, 2048, cum_gt, cnt_gt, p32, &cross, lane, true);
    __syncthreads();
    const unsigned b2 = (unsigned)cross.bucket;     // found again by construction (the whole of b1 crossed)
    cum_gt = cross.cum_before;
    cnt_gt = cross.cnt_before;
    const unsigned pre22 = (b1 << 11) | b2;
    // ---- level 3: bits[9:0] ----------------------------------------------------------------------------------------------
    clear_hist(1024);
    for (int i = tid; i < V; i += 256) {
      const float v = value(i);
      const unsigned u = __float_as_uint(v);
      if ((u >> 10) == pre22) { atomicAdd(&h_cnt[u & 0x3FFu], 1u); atomicAdd(&h_sum[u & 0x3FFu], (double)v); }
    }
    __syncthreads();
    if (wave == 0) scan_buckets(h_cnt, h_sum, 1024, cum_gt, cnt_gt, p32, &cross, lane, true);
    __syncthreads();
    const unsigned b3 = (unsigned)cross.bucket;
    cum_gt = cross.cum_before;
    cnt_gt = cross.cnt_before;
    prefix = (pre22 << 10) | b3;
    n_eq = h_cnt[b3];
  }
  __syncthreads();

  // ---- how many copies of the crossing value tau still fit (sequential double adds of tau in the reference) -------------
  unsigned c_fit = 0;
  const double tau = crossing ? (double)__uint_as_float(prefix) : 0.0;
  if (crossing) {
    if (tau == 0.0) c_fit = n_eq;
    else {
      double guess = floor(((double)p32 - cum_gt) / tau);
      long long c0 = guess < 0 ? 0 : (guess > (double)n_eq ? (long long)n_eq : (long long)guess);
      while (c0 < (long long)n_eq && (float)(cum_gt + (double)(c0 + 1) * tau) <= p32) ++c0;
      while (c0 > 0 && (float)(cum_gt + (double)c0 * tau) > p32) --c0;
      c_fit = (unsigned)c0;
    }
  }
  // M = cnt_gt + c_fit elements fit; kept = the M - 1 best = everything ranked strictly better than e*.
  // kept = { v > theta } U first r of { v == theta } (index order)
  unsigned theta, r, n_theta;
  double e_star;
  bool none_kept = false;
  if (crossing && c_fit >= 1) {
    theta = prefix; r = c_fit - 1; n_theta = n_eq; e_star = tau;
  } else {
    // e* = the worst-ranked element above tau (or of the whole row when nothing crossed): smallest value > tau, last index
    const unsigned lo = crossing ? prefix : 0u;
    unsigned mn = 0xFFFFFFFFu;
    for (int i = tid; i < V; i += 256) {
      const unsigned u = __float_as_uint(value(i));
      if ((!crossing || u > lo) && u < mn) mn = u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
    if (lane == 0) red_u[wave] = mn;
    __syncthreads();
    if (tid == 0) sh_theta = min(min(red_u[0], red_u[1]), min(red_u[2], red_u[3]));
    __syncthreads();
    theta = sh_theta;
    if (theta == 0xFFFFFFFFu) { none_kept = true; r = 0; n_theta = 0; e_star = 0.0; }
    else {
      unsigned c = 0;
      for (int i = tid; i < V; i += 256) c += (__float_as_uint(value(i)) == theta) ? 1u : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((int)c, o);
      __syncthreads();
      if (lane == 0) red_u[wave] = c;
      __syncthreads();
      if (tid == 0) sh_ntheta = red_u[0] + red_u[1] + red_u[2] + red_u[3];
      __syncthreads();
      n_theta = sh_ntheta;
      r = n_theta - 1;
      e_star = (double)__uint_as_float(theta);
    }
  }
  // kept mass (double, then fp32 like torch.sum's result type): all fitting mass minus e*
  const double fit_mass = cum_gt + (double)c_fit * tau;
  if (tid == 0) sh_mass = none_kept ? 0.0f : (float)(fit_mass - e_star);
  // index of the r-th theta-valued element (ties at the cut keep index order); -1: none of them, V: all of them
  if (tid == 0) sh_idx_cut = (r == 0) ? -1 : (long long)V;
  __syncthreads();
  if (!none_kept && r > 0 && r < n_theta && wave == 0) {
    unsigned running = 0;
    long long cut = V;
    for (int base = 0; base < V; base += 64) {
      const int i = base + lane;
      const bool eq = (i < V) && (__float_as_uint(value(i)) == theta);
      const unsigned long long bal = __ballot(eq);
      const unsigned c = (unsigned)__popcll(bal);
      if (running + c >= r) {
        unsigned need = r - running;   // the need-th set bit of bal
        unsigned long long m = bal;
        for (unsigned k = 1; k < need; ++k) m &= m - 1;
        cut = base + (__ffsll(m) - 1);
        break;
      }
      running += c;
    }
    if (lane == 0) sh_idx_cut = cut;
  }
  __syncthreads();
  const long long idx_cut = sh_idx_cut;
  const float mass = sh_mass;

  // ---- final pass: keys and top-n --------------------------------------------------------------------------------------
  u64 item = 0, tauk = 0;
  float tau_key = order_bits_inv(0);
  for (int base = 0; base < V; base += 256) {
    const int i = base + tid;
    const bool ok = i < V;
    float key = 0.f;
    if (ok) {
      const float v = value(i);
      const unsigned u = __float_as_uint(v);
      const bool kept = !none_kept && (u > theta || (u == theta && (long long)i <= idx_cut));
      key = ((kept ? v : 0.f) / mass) / e[i];
    }
    const bool cand = ok && !(key < tau_key);
    u64 bal = __ballot(cand);
    if (bal == 0) continue;
    const u64 mine = make_item(key, (uint32_t)i);
    while (bal) {
      const int L = __ffsll((unsigned long long)bal) - 1;
      bal &= bal - 1;
      const u64 c = readlane64(mine, L);
      if (c > tauk) {
        wave_insert(item, c, n, lane);
        tauk = readlane64(item, n - 1);
        tau_key = order_bits_inv((uint32_t)(tauk >> 32));
      }
    }
  }
  if (lane < n) lds_items[wave * n + lane] = item;
  __syncthreads();
  rank_merge_store(lds_items, 4 * n, n, a.out_ids + row * n, a.out_keys ? a.out_keys + row * n : nullptr, nullptr);
}


// ---- register-resident variant (V <= 53245) ------------------------------------------------------------------------------
// The streaming kernel above re-reads the row once per radix level, the theta / cut passes and the final pass, and its
// level-1 histogram serialises on the handful of buckets a softmax row falls into (LDS atomics, 50k elements on ~100
// addresses).  Here 1024 threads hold the whole row in registers (13 x 16 B per thread), so HBM sees the row ONCE plus the
// noise once; level 1 is the exponent alone (256 buckets) replicated 16 x by lane so that at most 4 lanes of a wave meet on
// an address, levels 2 / 3 split the mantissa 12 + 11 bits, and a wave whose active lanes all hit ONE bucket (uniform and
// near-uniform rows) adds once.  Same arithmetic as the streaming kernel: count + double sum per bucket, fp32 compare.
// BG_TOPP_TRACE (csrc `make trace`, benchmarks/probes/topp_trace.py): wave 0 of every row's workgroup stamps the wall clock
// (100 MHz) at the phase boundaries.
#ifdef BG_TOPP_TRACE
#define TOPP_TRACE_ROWS 8192
__device__ unsigned long long g_topp_trace[TOPP_TRACE_ROWS][12];
#define TOPP_STAMP(slot)                                                                                 \
  do {                                                                                                   \
    __builtin_amdgcn_s_waitcnt(0);                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < TOPP_TRACE_ROWS) g_topp_trace[blockIdx.x][slot] = wall_clock64(); \
  } while (0)
extern "C" int bg_debug_topp_trace(unsigned long long* host, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_topp_trace), bytes < sizeof(g_topp_trace) ? bytes : sizeof(g_topp_trace));
}
#else
#define TOPP_STAMP(slot)
#endif

#define TOPP_NV 13
#define TOPP_THREADS 1024
#define TOPP_CAP (TOPP_NV * TOPP_THREADS * 4)

// Only the SUMS are histogrammed (one LDS atomic per positive element): a bucket matters to the scan iff its sum is positive -
// zeros never move the prefix sum - and the multiplicity of the crossing value is counted from the registers afterwards.
// scan_buckets for sum-only histograms: "non-empty" = positive sum.  Executed by wave 0; nb is a multiple of 64.
__device__ void scan_sums(const double* sum, int nb, double cum0, float p32, Cross* out, int lane, bool force) {
  const int per = nb / 64;
  double s = 0.0;
  for (int j = 0; j < per; ++j) s += sum[nb - 1 - (lane * per + j)];
  double ps = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double os = shfl_up_d(ps, d);
    if (lane >= d) ps += os;
  }
  double cum = cum0 + (ps - s);
  int found = 0, fb = 0, lb = -1;
  double fcum = 0.0, lcum = 0.0;
  for (int j = 0; j < per && !found; ++j) {
    const int b = nb - 1 - (lane * per + j);
    if (!(sum[b] > 0.0)) continue;
    lb = b; lcum = cum;
    const double nxt = cum + sum[b];
    if ((float)nxt > p32) { found = 1; fb = b; fcum = cum; }
    else cum = nxt;
  }
  const unsigned long long bal = __ballot(found);
  if (bal == 0) {
    if (force) {
      const unsigned long long ne = __ballot(lb >= 0);
      const int last = 63 - __clzll(ne);
      if (lane == last) { out->found = 1; out->bucket = lb; out->cnt_before = 0; out->cum_before = lcum; }
    } else if (lane == 63) { out->found = 0; out->bucket = -1; out->cnt_before = 0; out->cum_before = cum; }
  } else {
    const int first = __ffsll(bal) - 1;
    if (lane == first) { out->found = 1; out->bucket = fb; out->cnt_before = 0; out->cum_before = fcum; }
  }
}

// The row into registers: vector c of thread t covers elements (c * 1024 + t) * 4 - mis + {0..3}; outside [0, V): -1.
// Every vector is REQUESTED before the first one is used (round 3: with the canonicalising `+ 0.0f` inside the branch that
// guarded each load the compiler waited for every vector before it asked for the next - thirteen dependent round trips per
// thread, the 8.5 us "row" phase of profiles/r2_c_topp_phases.txt; 2.6 us now).  No branch around the loads: the address is
// clamped into the row's own 16-byte vectors (a vector that straddles an end of the row is read whole - an aligned 16-byte load
// that holds one byte of the row cannot leave the row's pages) and the elements outside are masked afterwards, by two
// wave-uniform tests (only the first vector can start before the row, only the last ones can end behind it) instead of 104
// per-element range checks (whose masks spilled).
__device__ __forceinline__ void topp_load_row(const float* p, int V, int mis, int tid, float (&x)[TOPP_NV][4]) {
  const char* p_al = (const char*)(p - mis);
  const int last = ((V + mis + 3) / 4 - 1) * 4 - mis;
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
    const unsigned off = (unsigned)(min(max(i0, -mis), last) + mis);
    const f32x4 raw = __builtin_nontemporal_load((const f32x4*)(p_al + 4 * off));
#pragma unroll
    for (int j = 0; j < 4; ++j) x[c][j] = raw[j];
  }
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[c][j] += 0.0f;
    if (c == 0 && mis) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i0 + j < 0) x[c][j] = -1.0f;
    }
    if ((c + 1) * TOPP_THREADS * 4 - mis > V) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i0 + j >= V) x[c][j] = -1.0f;
    }
  }
}

// BATCHED: the row's loads all in flight (topp_load_row); false: one vector after the other - the persistent leftover kernel
// (topp_regs_todo_kernel) has no registers to spare for it (380 B of scratch against 92)
template <bool BATCHED>
__device__ __forceinline__ void topp_regs_row(const ToppArgs& a, const int64_t row) {
  __shared__ double h_sum[4096];
  __shared__ double e_sum[256];
  __shared__ u64 lds_items[(TOPP_THREADS / 64) * BG_MAX_TOPN];
  __shared__ Cross cross;
  __shared__ unsigned red_u[TOPP_THREADS / 64], red_v[TOPP_THREADS / 64];
  __shared__ unsigned sh_theta, sh_ntheta;
  __shared__ long long sh_idx_cut;
  __shared__ float sh_mass;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = a.probs + row * a.stride;
  const float* e = a.noise + row * a.noise_stride;
  const int V = (int)a.V, n = a.n;
  const float p32 = a.top_p;

  uint32_t blk[BG_MAX_BLOCKED];
#pragma unroll
  for (int i = 0; i < BG_MAX_BLOCKED; ++i) {
    int64_t b = (i < a.n_blocked) ? a.blocked[row * a.n_blocked + i] : -1;
    blk[i] = (b >= 0 && b < a.V) ? (uint32_t)b : 0xFFFFFFFFu;
  }
  auto value = [&](int i) -> float {                  // global-memory view of element i (rare tie-at-the-cut pass only)
    float v = p[i];
    for (int b = 0; b < a.n_blocked; ++b)
      if (blk[b] == (uint32_t)i) v = 0.f;
    return v + 0.0f;
  };

  // ---- the row, once: vector c of thread t covers elements (c * 512 + t) * 4 - mis + {0..3}; outside [0, V): -1 (skipped) ----
  TOPP_STAMP(0);
  const int mis = (int)((((uintptr_t)p) & 15) / 4);
  float x[TOPP_NV][4];
  if constexpr (BATCHED) {
    topp_load_row(p, V, mis, tid, x);
  } else {
#pragma unroll
    for (int c = 0; c < TOPP_NV; ++c) {
      const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
      if (i0 >= 0 && i0 + 4 <= V) {
        const f32x4 raw = __builtin_nontemporal_load((const f32x4*)(p + i0));
#pragma unroll
        for (int j = 0; j < 4; ++j) x[c][j] = raw[j] + 0.0f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[c][j] = (i0 + j >= 0 && i0 + j < V) ? p[i0 + j] + 0.0f : -1.0f;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
    for (int b = 0; b < a.n_blocked; ++b) {
      const uint32_t d = blk[b] - (uint32_t)i0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (d == (uint32_t)j && i0 + j >= 0) x[c][j] = 0.f;
    }
  }
  auto clear_hist = [&](int nb) {
    for (int i = tid; i < nb; i += TOPP_THREADS) h_sum[i] = 0.0;
    __syncthreads();
  };
  auto present = [](unsigned u) { return (int)u >= 0; };          // sign bit = padding

  TOPP_STAMP(1);   // row in registers
  // ---- level 1: exponent (bits 30:23), 16 lane-replicas per bucket -----------------------------------------------------
  clear_hist(4096);
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = x[c][j];
      if (v > 0.f) atomicAdd(&h_sum[((__float_as_uint(v) >> 23) & 255u) * 16 + (lane & 15)], (double)v);
    }
  __syncthreads();
  if (tid < 256) {
    double ss = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) ss += h_sum[tid * 16 + r];
    e_sum[tid] = ss;
  }
  __syncthreads();
  TOPP_STAMP(2);   // level-1 histogram
  if (wave == 0) scan_sums(e_sum, 256, 0.0, p32, &cross, lane, false);
  __syncthreads();
  TOPP_STAMP(3);   // level-1 scan
  const bool crossing = cross.found != 0;
  unsigned prefix = 0;
  double cum_gt = cross.cum_before;
  unsigned n_eq = 0;
  if (crossing) {
    const unsigned b1 = (unsigned)cross.bucket;
    // All values of the crossing exponent identical (uniform rows, heavy ties)?  Then that value IS the crossing value and
    // the two mantissa levels - whose atomics would all hit one address - are skipped.
    unsigned mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned u = __float_as_uint(x[c][j]);
        if (x[c][j] > 0.f && (u >> 23) == b1) { mn = min(mn, u); mx = max(mx, u); }
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
      mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    }
    if (lane == 0) { red_u[wave] = mn; red_v[wave] = mx; }
    __syncthreads();
    mn = red_u[0]; mx = red_v[0];
    for (int w = 1; w < TOPP_THREADS / 64; ++w) { mn = min(mn, red_u[w]); mx = max(mx, red_v[w]); }
    __syncthreads();
    if (mn == mx) {
      prefix = mn;
    } else {
      // ---- level 2: mantissa bits 22:11 inside exponent b1 ---------------------------------------------------------------
      clear_hist(4096);
#pragma unroll
      for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = x[c][j];
          const unsigned u = __float_as_uint(v);
          if (v > 0.f && (u >> 23) == b1) atomicAdd(&h_sum[(u >> 11) & 0xFFFu], (double)v);
        }
      __syncthreads();
      TOPP_STAMP(4);   // min/max + level-2 histogram
      if (wave == 0) scan_sums(h_sum, 4096, cum_gt, p32, &cross, lane, true);
      __syncthreads();
      TOPP_STAMP(5);   // level-2 scan
      const unsigned pre20 = (b1 << 12) | (unsigned)cross.bucket;
      cum_gt = cross.cum_before;
      __syncthreads();
      // ---- level 3: bits 10:0 ------------------------------------------------------------------------------------------------
      clear_hist(2048);
#pragma unroll
      for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = x[c][j];
          const unsigned u = __float_as_uint(v);
          if (v > 0.f && (u >> 11) == pre20) atomicAdd(&h_sum[u & 0x7FFu], (double)v);
        }
      __syncthreads();
      TOPP_STAMP(6);   // level-3 histogram
      if (wave == 0) scan_sums(h_sum, 2048, cum_gt, p32, &cross, lane, true);
      __syncthreads();
      TOPP_STAMP(7);   // level-3 scan
      cum_gt = cross.cum_before;
      prefix = (pre20 << 11) | (unsigned)cross.bucket;
    }
  }
  __syncthreads();
  {   // multiplicity of the crossing value (0 when nothing crossed: prefix = 0 then counts the zeros, unused)
    unsigned cc = 0;
#pragma unroll
    for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) cc += (__float_as_uint(x[c][j]) == prefix) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cc += __shfl_xor((int)cc, o);
    if (lane == 0) red_u[wave] = cc;
    __syncthreads();
    if (tid == 0) {
      unsigned t2 = 0;
      for (int w = 0; w < TOPP_THREADS / 64; ++w) t2 += red_u[w];
      sh_ntheta = t2;
    }
    __syncthreads();
    n_eq = sh_ntheta;
    __syncthreads();
  }

  // ---- how many copies of the crossing value tau still fit (as in the streaming kernel) -------------------------------------
  unsigned c_fit = 0;
  const double tau = crossing ? (double)__uint_as_float(prefix) : 0.0;
  if (crossing) {
    if (tau == 0.0) c_fit = n_eq;
    else {
      double guess = floor(((double)p32 - cum_gt) / tau);
      long long c0 = guess < 0 ? 0 : (guess > (double)n_eq ? (long long)n_eq : (long long)guess);
      while (c0 < (long long)n_eq && (float)(cum_gt + (double)(c0 + 1) * tau) <= p32) ++c0;
      while (c0 > 0 && (float)(cum_gt + (double)c0 * tau) > p32) --c0;
      c_fit = (unsigned)c0;
    }
  }
  unsigned theta, r, n_theta;
  double e_star;
  bool none_kept = false;
  if (crossing && c_fit >= 1) {
    theta = prefix; r = c_fit - 1; n_theta = n_eq; e_star = tau;
  } else {
    const unsigned lo = crossing ? prefix : 0u;
    unsigned mn = 0xFFFFFFFFu;
#pragma unroll
    for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned u = __float_as_uint(x[c][j]);
        if (present(u) && (!crossing || u > lo) && u < mn) mn = u;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
    if (lane == 0) red_u[wave] = mn;
    __syncthreads();
    if (tid == 0) {
      unsigned m2 = red_u[0];
      for (int w = 1; w < TOPP_THREADS / 64; ++w) m2 = min(m2, red_u[w]);
      sh_theta = m2;
    }
    __syncthreads();
    theta = sh_theta;
    if (theta == 0xFFFFFFFFu) { none_kept = true; r = 0; n_theta = 0; e_star = 0.0; }
    else {
      unsigned cc = 0;
#pragma unroll
      for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) cc += (__float_as_uint(x[c][j]) == theta) ? 1u : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cc += __shfl_xor((int)cc, o);
      __syncthreads();
      if (lane == 0) red_u[wave] = cc;
      __syncthreads();
      if (tid == 0) {
        unsigned t2 = 0;
        for (int w = 0; w < TOPP_THREADS / 64; ++w) t2 += red_u[w];
        sh_ntheta = t2;
      }
      __syncthreads();
      n_theta = sh_ntheta;
      r = n_theta - 1;
      e_star = (double)__uint_as_float(theta);
    }
  }
  const double fit_mass = cum_gt + (double)c_fit * tau;
  if (tid == 0) sh_mass = none_kept ? 0.0f : (float)(fit_mass - e_star);
  if (tid == 0) sh_idx_cut = (r == 0) ? -1 : (long long)V;
  __syncthreads();
  if (!none_kept && r > 0 && r < n_theta && wave == 0) {     // ties at the cut: the r-th theta-valued element in index order
    unsigned running = 0;
    long long cut = V;
    for (int base = 0; base < V; base += 64) {
      const int i = base + lane;
      const bool eq = (i < V) && (__float_as_uint(value(i)) == theta);
      const unsigned long long bal = __ballot(eq);
      const unsigned c = (unsigned)__popcll(bal);
      if (running + c >= r) {
        unsigned need = r - running;
        unsigned long long m = bal;
        for (unsigned k = 1; k < need; ++k) m &= m - 1;
        cut = base + (__ffsll(m) - 1);
        break;
      }
      running += c;
    }
    if (lane == 0) sh_idx_cut = cut;
  }
  __syncthreads();
  const long long idx_cut = sh_idx_cut;
  const float mass = sh_mass;

  TOPP_STAMP(8);   // multiplicity, fit, theta, cut
  // ---- final pass: keys from the registers and the noise (one HBM read), top-n per wave, merge ----------------------------------
  WaveList wl;
  wl.init(n, lane);
  const bool noise_vec = ((((uintptr_t)(e - mis)) & 15) == 0);
  auto load_noise = [&](int c, float (&en)[4]) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
    if (noise_vec && i0 >= 0 && i0 + 4 <= V) {
      const f32x4 raw = __builtin_nontemporal_load((const f32x4*)(e + i0));
#pragma unroll
      for (int j = 0; j < 4; ++j) en[j] = raw[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) en[j] = (i0 + j >= 0 && i0 + j < V) ? e[i0 + j] : 1.0f;
    }
  };
  constexpr int PF = 4;                                  // noise vectors in flight per lane
  float en_ring[PF][4];
#pragma unroll
  for (int c = 0; c < PF && c < TOPP_NV; ++c) load_noise(c, en_ring[c]);
  const float zero_over_mass = 0.f / mass;               // +0, or NaN when nothing is kept
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    float en[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) en[j] = en_ring[c % PF][j];
    if (c + PF < TOPP_NV) load_noise(c + PF, en_ring[c % PF]);
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
    // candidates by the full 64-bit order (key, then index): rows that keep few tokens are mostly zero keys, and a zero must
    // not stay a candidate just because it ties with the list's zero threshold
    u64 it[4];
    uint32_t cmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = x[c][j];
      const unsigned u = __float_as_uint(v);
      const bool in = i0 + j >= 0 && i0 + j < V;
      const bool kept = !none_kept && (u > theta || (u == theta && (long long)(i0 + j) <= idx_cut));
      // a dropped token's key is (0 / mass) / e: +-0 unless e is 0 or NaN (then NaN) - no division needed for the usual case
      const float key = kept ? (v / mass) / en[j] : ((zero_over_mass == 0.f && en[j] == en[j] && en[j] != 0.f && fabsf(en[j]) != INFINITY) ? 0.f : zero_over_mass / en[j]);
      it[j] = make_item(key, (uint32_t)(i0 + j));
      cmask |= (in && it[j] > wl.tau) ? (1u << j) : 0u;
    }
    while (__ballot(cmask != 0)) {
      u64 mine = 0;
      if (cmask) {
        const int j = __ffs(cmask) - 1;
        cmask &= cmask - 1;
        mine = j == 0 ? it[0] : (j == 1 ? it[1] : (j == 2 ? it[2] : it[3]));
      }
      wl.offer_lanes(mine);
    }
    __builtin_amdgcn_sched_barrier(0);     // keep PF noise vectors in flight: hoisting all 13 loads would spill the row
  }
  TOPP_STAMP(9);   // noise pass + per-wave lists
  if (lane < n) lds_items[wave * n + lane] = wl.item;
  __syncthreads();
  rank_merge_store(lds_items, (TOPP_THREADS / 64) * n, n, a.out_ids + row * n, a.out_keys ? a.out_keys + row * n : nullptr, nullptr);
  TOPP_STAMP(10);  // merge + store
}

__global__ __launch_bounds__(TOPP_THREADS) void topp_regs_kernel(ToppArgs a) { topp_regs_row<true>(a, blockIdx.x); }

// The rows topp_list_kernel could not do: todo[0] of them, listed in todo[1 ..]; a few persistent workgroups walk the list
// (normally empty: the launch then costs a few microseconds, not a workgroup of 1024 threads per row).
__global__ __launch_bounds__(TOPP_THREADS) void topp_regs_todo_kernel(ToppArgs a, const int* todo) {
  const int count = todo[0];
  for (int i = blockIdx.x; i < count; i += gridDim.x) {
    topp_regs_row<false>(a, todo[1 + i] & 0xFFFFFF);
    __syncthreads();                                     // the next row reuses the shared arrays
  }
}

__global__ void topp_reset_kernel(int* todo) { todo[0] = 0; }

// ---- register-resident variant, list form (round 2) -----------------------------------------------------------------------------
// The first form above spends 77 us on a row of 50304 (benchmarks/probes/topp_trace.py: 8 us for the row, 8 for the noise -
// and 60 in between).  Every instruction executed per ELEMENT of the register-resident row costs 0.35 us (16 waves share the
// CU's 64 lanes), a CU streams a 200 KB row in no less than 8 us, and one workgroup per CU overlaps nothing by itself.  So:
//  * level 1 (the exponent; exact integer bucket sums) is the only histogram over the registers.  Everything at or above the
//    crossing exponent - the only elements that can be kept or decide the cut, a few thousand - is compacted into LDS as
//    (index, value bits) pairs, each wave filling a region of its own (ballot + popcount, no atomics);
//  * from there the row's 52 registers are free: all thirteen noise vectors are requested at once and arrive while the
//    remaining phases run on the list - 8-bit radix levels (count + double sum) until the crossing bucket holds at most 64
//    values, those ranked by one wave (crossing value, multiplicity, mass above, next value up), kept entries moved to the
//    front of each region, their keys (v / mass) / e formed with gathered noise;
//  * the streamed noise row is only checked for 0 / NaN: a dropped token's key is +0 unless its noise is 0 or NaN.
// Rows this cannot do - p beyond the row's mass, more candidates than the list holds (flat rows, heavy ties), a tie at the
// cut that the index order must break, fewer than n kept tokens, nothing kept, zero or NaN noise - are FLAGGED in the caller's
// workspace and redone by the first form in a second launch of a few persistent workgroups: same results.
// level-1 histogram: replicas of a bucket (one per lane & (n - 1)), buckets n u64 apart.  Measured per row at 8192 x 50304
// (round 3): n = 16 7.0 us, 8 7.4, 32 8.2, 16 replicas at a stride of 17 (to spread the banks) 8.8 - the phase runs at the LDS's
// rate of 64-bit atomics (~3 per clock per CU), not at its bank conflicts.
#define TOPP_HSTRIDE 16
#define TOPP_NOISE_EARLY 7                        // noise vectors requested before the candidate pass (registers)
#define TOPP_REGION 768                           // list entries per wave (16 regions: 96 KB)
// 96 KB of regions + 48 KB of their noise + 8 KB of histograms and lists: more static LDS than any part but gfx950 (160 KB per CU) offers.  The
// library is written for gfx950 only (csrc/Makefile); a build for another --offload-arch stops here with a message instead of
// failing somewhere inside the linker.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "topp_list_kernel needs the 160 KB LDS of gfx950 (MI355X): build with --offload-arch=gfx950"
#endif
__global__ __launch_bounds__(TOPP_THREADS) void topp_list_kernel(ToppArgs a, int* todo) {
  __shared__ u64 big[(TOPP_THREADS / 64) * TOPP_REGION];     // level 1: 256 exponents x 16 replicas of u64 sums; then the regions
  __shared__ float noise_l[(TOPP_THREADS / 64) * TOPP_REGION]; // the noise of the regions' entries, position for position
  __shared__ double e_sum[256];
  __shared__ double h_sum[256];
  __shared__ unsigned h_cnt[256];
  __shared__ unsigned list[64];
  __shared__ unsigned list_n;
  __shared__ u64 lds_items[(TOPP_THREADS / 64) * BG_MAX_TOPN];
  __shared__ Cross cross;
  __shared__ unsigned red_u[TOPP_THREADS / 64], red_c[TOPP_THREADS / 64];
  __shared__ unsigned sh_bad, sh_kept;
  __shared__ unsigned l_prefix, l_neq, l_next, l_next_cnt;
  __shared__ double l_cum;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row = blockIdx.x;
  const float* p = a.probs + row * a.stride;
  const float* e = a.noise + row * a.noise_stride;
  const int V = (int)a.V, n = a.n;
  const float p32 = a.top_p;
  // workgroup-uniform: the first form redoes this row.  The reason rides in the top byte (rows < 2^24: bg_topp_select checks)
  auto give_up = [&](int why) {
    if (tid == 0) todo[1 + atomicAdd(&todo[0], 1)] = (int)row | (why << 24);
  };

  uint32_t blk[BG_MAX_BLOCKED];
#pragma unroll
  for (int i = 0; i < BG_MAX_BLOCKED; ++i) {
    int64_t b = (i < a.n_blocked) ? a.blocked[row * a.n_blocked + i] : -1;
    blk[i] = (b >= 0 && b < a.V) ? (uint32_t)b : 0xFFFFFFFFu;
  }

  // ---- the row, once: vector c of thread t covers elements (c * 1024 + t) * 4 - mis + {0..3}; outside [0, V): -1 (skipped) ----
  TOPP_STAMP(0);
  const int mis = (int)((((uintptr_t)p) & 15) / 4);
  float x[TOPP_NV][4];
  topp_load_row(p, V, mis, tid, x);
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
    for (int b = 0; b < a.n_blocked; ++b) {
      const uint32_t d = blk[b] - (uint32_t)i0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (d == (uint32_t)j && i0 + j >= 0) x[c][j] = 0.f;
    }
  }
  TOPP_STAMP(1);   // row in registers

  // ---- level 1: exponent (bits 30:23), 16 lane-replicas per bucket.  The values of one exponent are 24-bit integers times one
  // power of two, so a bucket's sum is an INTEGER sum (below 2^40 for 53245 elements): exact whatever the order.
  for (int i = tid; i < 256 * TOPP_HSTRIDE; i += TOPP_THREADS) big[i] = 0ull;
  if (tid == 0) { sh_bad = 0u; sh_kept = 0u; list_n = 0u; }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = x[c][j];
      const unsigned u = __float_as_uint(v);
      const unsigned ex = (u >> 23) & 255u;
      if (v > 0.f) atomicAdd(&big[ex * TOPP_HSTRIDE + (lane & (TOPP_HSTRIDE - 1))], (u64)((u & 0x7FFFFFu) | (ex ? 0x800000u : 0u)));
    }
  __syncthreads();
  if (tid < 256) {
    u64 ss = 0;
#pragma unroll
    for (int r = 0; r < TOPP_HSTRIDE; ++r) ss += big[tid * TOPP_HSTRIDE + r];
    e_sum[tid] = ldexp((double)ss, (tid ? tid : 1) - 150);     // exact: ss < 2^53
  }
  __syncthreads();
  TOPP_STAMP(2);   // level-1 histogram
  if (wave == 0) scan_sums(e_sum, 256, 0.0, p32, &cross, lane, false);
  __syncthreads();
  TOPP_STAMP(3);   // level-1 scan
  if (cross.found == 0) { give_up(1); return; }         // p is not below the row's mass: everything but the last element is kept
  const unsigned b1 = (unsigned)cross.bucket;
  double cum_gt = cross.cum_before;
  if (b1 == 0) { give_up(2); return; }                  // the cut falls among the denormals

  // The noise row is requested while the row is still needed for one more pass - as many vectors as fit beside it in the
  // registers now, the rest behind that pass - and arrives behind the phases below.  It has the probabilities' misalignment
  // (checked on the host), so vector c is one aligned 16-byte load; a vector that straddles the row's ends is read whole (the
  // bytes outside belong to the neighbouring rows or the 16-byte block the row ends in)
  f32x4 en[TOPP_NV];
  const char* e_al = (const char*)(e - mis);
  auto load_noise = [&](int c) __attribute__((always_inline)) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
    en[c] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
    if (i0 > -4 && i0 < V) en[c] = __builtin_nontemporal_load((const f32x4*)(e_al + (unsigned)((c * TOPP_THREADS + tid) * 16)));
  };
#pragma unroll
  for (int c = 0; c < TOPP_NOISE_EARLY; ++c) load_noise(c);
  // ---- everything at or above the crossing exponent goes to this wave's region of the list (no atomics) -----------------------
  u64* region = big + wave * TOPP_REGION;
  unsigned mine = 0;                                   // wave-uniform count
  const unsigned lowest = b1 << 23;                  // positive floats order like their bit patterns
  // which of this thread's 52 elements were listed (bit 4 c + j): the key pass walks the same ballots again, so an entry's
  // position in the region - and with it the thread that holds the entry's noise in its registers - needs no look-up
  u64 listed = 0;
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) any |= (int)__float_as_uint(x[c][j]) >= (int)lowest;    // the padding (-1) is negative as an int
    if (__ballot(any) == 0) continue;                    // most vectors of a peaked row hold no candidate
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned u = __float_as_uint(x[c][j]);
      const bool hit = (int)u >= (int)lowest;
      const unsigned long long bal = __ballot(hit);
      if (bal == 0) continue;
      const unsigned pos = mine + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
      if (hit && pos < TOPP_REGION) region[pos] = ((u64)(uint32_t)(i0 + j) << 32) | u;
      listed |= hit ? (1ull << (4 * c + j)) : 0ull;
      mine += (unsigned)__popcll(bal);
    }
  }
  if (mine > TOPP_REGION && lane == 0) atomicOr(&sh_bad, 1u);
  // the registers of the row are free from here: the rest of the noise row is requested now
#pragma unroll
  for (int c = TOPP_NOISE_EARLY; c < TOPP_NV; ++c) load_noise(c);
  if (tid < 256) { h_sum[tid] = 0.0; h_cnt[tid] = 0u; }
  __syncthreads();
  if (sh_bad) { give_up(3); return; }
  TOPP_STAMP(4);   // candidates listed
  const unsigned cnt = min(mine, (unsigned)TOPP_REGION);

  // ---- radix levels of 8 + 8 + 7 bits below the exponent, on the list, until the crossing bucket is small -----------------------
  unsigned prefix = b1, n_eq = 0, next_up = 0, next_cnt = 0;
  int shift = 23;
  unsigned in_bucket = 0xFFFFFFFFu;
  for (int lvl = 0; lvl < 3 && in_bucket > 64u; ++lvl) {
    const int nshift = lvl == 0 ? 15 : (lvl == 1 ? 7 : 0);
    const unsigned mask = lvl == 2 ? 127u : 255u;
    for (unsigned i = lane; i < cnt; i += 64) {
      const unsigned u = (unsigned)region[i];
      if ((u >> shift) == prefix) {
        atomicAdd(&h_cnt[(u >> nshift) & mask], 1u);
        atomicAdd(&h_sum[(u >> nshift) & mask], (double)__uint_as_float(u));
      }
    }
    __syncthreads();
    if (wave == 0) scan_buckets(h_cnt, h_sum, 256, cum_gt, 0u, p32, &cross, lane, true);
    __syncthreads();
    prefix = (prefix << (lvl == 2 ? 7 : 8)) | (unsigned)cross.bucket;
    shift = nshift;
    cum_gt = cross.cum_before;
    in_bucket = h_cnt[cross.bucket];
    __syncthreads();
    if (in_bucket > 64u && lvl < 2) {
      if (tid < 256) { h_sum[tid] = 0.0; h_cnt[tid] = 0u; }
      __syncthreads();
    }
  }
  if (shift == 0) {
    n_eq = in_bucket;                                  // all 31 bits resolved: the bucket is one value
  } else {
    for (unsigned i = lane; i < cnt; i += 64) {
      const unsigned u = (unsigned)region[i];
      if ((u >> shift) == prefix) {
        const unsigned pos = atomicAdd(&list_n, 1u);
        if (pos < 64u) list[pos] = u;
      }
    }
    __syncthreads();
    if (wave == 0) {
      const int lc = (int)min(list_n, 64u);
      const unsigned ui = lane < lc ? list[lane] : 0u;
      const double vi = (double)__uint_as_float(ui);
      double gt_sum = 0.0;
      unsigned eqc = 0;
      for (int j2 = 0; j2 < lc; ++j2) {
        const unsigned uj = list[j2];
        gt_sum += (uj > ui) ? (double)__uint_as_float(uj) : 0.0;
        eqc += (uj == ui) ? 1u : 0u;
      }
      const bool cr = lane < lc && (float)(cum_gt + gt_sum + (double)eqc * vi) > p32;
      unsigned best = cr ? ui : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
      if (best == 0u) {                                // the parent bucket crossed, a last-bit difference of the re-summed
        best = lane < lc ? ui : 0xFFFFFFFFu;           // values must not lose it: cross at the smallest value
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
      }
      unsigned nxt = (lane < lc && ui > best) ? ui : 0xFFFFFFFFu;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) nxt = min(nxt, (unsigned)__shfl_xor((int)nxt, o));
      const unsigned long long is_best = __ballot(lane < lc && ui == best);
      const unsigned long long is_next = __ballot(lane < lc && ui == nxt);
      if (lane == __ffsll(is_best) - 1) {
        l_prefix = ui; l_neq = eqc; l_cum = cum_gt + gt_sum;
        l_next = nxt == 0xFFFFFFFFu ? 0u : nxt;
        l_next_cnt = (unsigned)__popcll(is_next);
      }
    }
    __syncthreads();
    prefix = l_prefix; n_eq = l_neq; cum_gt = l_cum; next_up = l_next; next_cnt = l_next_cnt;
  }
  TOPP_STAMP(5);   // levels + ranked list

  // ---- how many copies of the crossing value tau still fit (sequential double adds of tau in the reference) -------------
  const double tau = (double)__uint_as_float(prefix);
  unsigned c_fit = 0;
  {
    double guess = floor(((double)p32 - cum_gt) / tau);
    long long c0 = guess < 0 ? 0 : (guess > (double)n_eq ? (long long)n_eq : (long long)guess);
    while (c0 < (long long)n_eq && (float)(cum_gt + (double)(c0 + 1) * tau) <= p32) ++c0;
    while (c0 > 0 && (float)(cum_gt + (double)c0 * tau) > p32) --c0;
    c_fit = (unsigned)c0;
  }
  // M = everything above tau + c_fit copies of tau fit; kept = the M - 1 best = everything ranked strictly better than e*
  unsigned theta, r, n_theta;
  double e_star;
  if (c_fit >= 1) {
    theta = prefix; r = c_fit - 1; n_theta = n_eq; e_star = tau;
  } else if (next_up != 0u) {                          // e* = the worst-ranked element above tau: the ranked list knows it
    theta = next_up; n_theta = next_cnt; r = n_theta - 1; e_star = (double)__uint_as_float(theta);
  } else {
    // smallest value above tau and how often it occurs, from the list (it holds everything at or above tau's exponent)
    unsigned mn = 0xFFFFFFFFu, mc = 0u;
    for (unsigned i = lane; i < cnt; i += 64) {
      const unsigned u = (unsigned)region[i];
      if (u > prefix) { mc = u < mn ? 1u : (mc + (u == mn ? 1u : 0u)); mn = min(mn, u); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned om = (unsigned)__shfl_xor((int)mn, o), oc = (unsigned)__shfl_xor((int)mc, o);
      mc = om < mn ? oc : (om == mn ? mc + oc : mc);
      mn = min(mn, om);
    }
    if (lane == 0) { red_u[wave] = mn; red_c[wave] = mc; }
    __syncthreads();
    mn = red_u[0]; mc = red_c[0];
#pragma unroll 1
    for (int w = 1; w < TOPP_THREADS / 64; ++w) {
      const unsigned om = red_u[w], oc = red_c[w];
      mc = om < mn ? oc : (om == mn ? mc + oc : mc);
      mn = min(mn, om);
    }
    if (mn == 0xFFFFFFFFu) {
      // tau is the row's maximum (its mass alone exceeds p): nothing is kept, the mass is 0 and every key is (0 / 0) / e = NaN -
      // all equal, so the n lowest indices win (NaN sorts first, ties by index)
      if (tid < n) {
        a.out_ids[row * n + tid] = tid;
        if (a.out_keys) a.out_keys[row * n + tid] = __uint_as_float(0x7FC00000u);
      }
      return;
    }
    theta = mn; n_theta = mc; r = n_theta - 1; e_star = (double)__uint_as_float(theta);
  }
  const float mass = (float)(cum_gt + (double)c_fit * tau - e_star);
  if (mass == 0.f) {
    // Nothing is kept (the element above tau is the row's maximum and e* itself): every key is (0 / 0) / e = NaN - all equal,
    // so the n lowest indices win (NaN sorts first, ties by index)
    if (tid < n) {
      a.out_ids[row * n + tid] = tid;
      if (a.out_keys) a.out_keys[row * n + tid] = __uint_as_float(0x7FC00000u);
    }
    return;
  }
  if (!(mass > 0.f) || !(mass < INFINITY)) { give_up(5); return; }
  // r == 0: none of the theta-valued elements is kept.  0 < r < n_theta: a tie AT the cut - the first r of them in index order
  // are kept: their indices go to a small shared list and the r-th smallest is the last one in
  int idx_cut = -1;
  if (r > 0) {
    if (n_theta > 64u) { give_up(4); return; }
    __syncthreads();
    if (tid == 0) list_n = 0u;
    __syncthreads();
    for (unsigned i = lane; i < cnt; i += 64) {
      const u64 ent = region[i];
      if ((unsigned)ent == theta) list[atomicAdd(&list_n, 1u) & 63u] = (unsigned)(ent >> 32);
    }
    __syncthreads();
    if (tid < (int)n_theta) {
      const unsigned mine_idx = list[tid];
      unsigned below = 0;
      for (unsigned j2 = 0; j2 < n_theta; ++j2) below += list[j2] < mine_idx ? 1u : 0u;
      if (below == r - 1) l_next = mine_idx;             // (l_next is free by now)
    }
    __syncthreads();
    idx_cut = (int)l_next;
  }
  const unsigned keep_from = theta + 1u;                 // plus the theta-valued entries up to idx_cut
  TOPP_STAMP(8);   // fit, theta

  // ---- the noise of the listed elements, from the registers it was streamed into: the listing pass is walked again (same
  // ballots, so the same positions) and each thread drops the noise of its own entries beside them in LDS.  (Round 2 gathered
  // e[index] for the kept entries from global memory instead: one 64-byte sector from HBM per entry, of a line the stream of
  // the same row had already passed through - 1.38 x the algorithmic traffic, profiles/r2_pmc_topp.json; now 1.00 x.)
  float* nz = noise_l + wave * TOPP_REGION;
  unsigned walked = 0;
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    const unsigned mc = (unsigned)(listed >> (4 * c)) & 15u;
    if (__ballot(mc != 0u) == 0) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool hit = (mc >> j) & 1u;
      const unsigned long long bal = __ballot(hit);
      if (bal == 0) continue;
      if (hit) nz[walked + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = en[c][j];     // (mine <= TOPP_REGION: a longer list gave up above)
      walked += (unsigned)__popcll(bal);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // ---- keys of the kept entries, top-n per wave: four entries per lane ----------------------------------------------------------
  WaveList wl;
  wl.init(n, lane);
  unsigned kept_w = 0;
  for (unsigned i0 = 0; i0 < cnt; i0 += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned i = i0 + k * 64 + lane;
      const u64 ent = i < cnt ? region[i] : 0ull;
      const bool kept = i < cnt && ((unsigned)ent >= keep_from || ((unsigned)ent == theta && (int)(uint32_t)(ent >> 32) <= idx_cut));
      kept_w += (unsigned)__popcll(__ballot(kept));
      u64 it = 0;
      if (kept) {
        it = make_item((__uint_as_float((uint32_t)ent) / mass) / nz[i], (uint32_t)(ent >> 32));
        it = it > wl.tau ? it : 0;
      }
      wl.offer_best(it);
    }
  }
  if (lane == 0) atomicAdd(&sh_kept, kept_w);
  // the streamed noise: a dropped token's key is (+0 / mass) / e = +0 unless e is 0 or NaN
  bool abnormal = false;
#pragma unroll
  for (int c = 0; c < TOPP_NV; ++c) {
    const int i0 = (c * TOPP_THREADS + tid) * 4 - mis;
#pragma unroll
    for (int j = 0; j < 4; ++j) abnormal |= (i0 + j >= 0) & (i0 + j < V) & !(fabsf(en[c][j]) > 0.f);
  }
  if (__ballot(abnormal) != 0 && lane == 0) atomicOr(&sh_bad, 1u);
  if (lane < n) lds_items[wave * n + lane] = wl.item;
  __syncthreads();
  TOPP_STAMP(9);   // keys, noise row
  if (sh_bad) { give_up(6); return; }
  rank_merge_store(lds_items, (TOPP_THREADS / 64) * n, n, a.out_ids + row * n, a.out_keys ? a.out_keys + row * n : nullptr, nullptr);
  const unsigned K = sh_kept;
  if (K < (unsigned)n && wave == 0) {
    // Fewer kept tokens than samples: the rest of the list are dropped tokens, whose keys are all +0 (the noise row holds no 0 /
    // NaN here) - the lowest indices that are not kept, in index order.  They lie below n + K <= 127: lane l tests l and l + 64.
    const int items = (TOPP_THREADS / 64) * n;
    bool kept_lo = false, kept_hi = false;
    for (int i = 0; i < items; ++i) {
      const u64 it = lds_items[i];
      const uint32_t idx = 0xFFFFFFFFu - (uint32_t)it;
      kept_lo |= it != 0 && idx == (uint32_t)lane;
      kept_hi |= it != 0 && idx == (uint32_t)(lane + 64);
    }
    const unsigned long long free_lo = __ballot(!kept_lo && lane < V), free_hi = __ballot(!kept_hi && lane + 64 < V);
    const unsigned below = (unsigned)__popcll(free_lo & ((1ull << lane) - 1ull));
    const unsigned pos_lo = K + below, pos_hi = K + (unsigned)__popcll(free_lo) + (unsigned)__popcll(free_hi & ((1ull << lane) - 1ull));
    if (!kept_lo && lane < V && pos_lo < (unsigned)n) {
      a.out_ids[row * n + pos_lo] = lane;
      if (a.out_keys) a.out_keys[row * n + pos_lo] = 0.f;
    }
    if (!kept_hi && lane + 64 < V && pos_hi < (unsigned)n) {
      a.out_ids[row * n + pos_hi] = lane + 64;
      if (a.out_keys) a.out_keys[row * n + pos_hi] = 0.f;
    }
  }
  TOPP_STAMP(10);  // merge + store
}

}  // namespace bg

using namespace bg;

extern "C" size_t bg_topp_workspace_bytes(int64_t rows, int64_t V, int n) {
  (void)n;
  return (V + 3 <= TOPP_CAP && rows > 0) ? (size_t)(rows + 1) * sizeof(int) : 0;     // the rows left to the first form: count + list
}

extern "C" int bg_topp_select(const float* probs, int64_t rows, int64_t V, int64_t row_stride, const float* noise,
                              int64_t noise_row_stride, const int64_t* blocked, int n_blocked, float top_p, int n,
                              int64_t* out_ids, float* out_keys, void* workspace, size_t workspace_bytes, void* stream) {
  BG_REQUIRE(probs && noise && out_ids, "bg_topp_select: null probs/noise/out_ids");
  BG_REQUIRE(rows > 0 && rows < (1ll << 31) && V > 0 && V < (1ll << 31), "bg_topp_select: rows=%lld V=%lld", (long long)rows, (long long)V);
  BG_REQUIRE(n >= 1 && n <= BG_MAX_TOPN && n <= V, "bg_topp_select: n=%d must be in [1, min(%d, V)]", n, BG_MAX_TOPN);
  BG_REQUIRE(row_stride >= V && noise_row_stride >= V, "bg_topp_select: row stride < V");
  BG_REQUIRE(n_blocked >= 0 && n_blocked <= BG_MAX_BLOCKED && (n_blocked == 0 || blocked), "bg_topp_select: n_blocked=%d", n_blocked);
  BG_REQUIRE(top_p == top_p, "bg_topp_select: top_p is NaN");
  ToppArgs a{probs, noise, blocked, V, row_stride, noise_row_stride, n, n_blocked, top_p, out_ids, out_keys};
  static const bool first_form = [] { const char* e = getenv("BRANCHGAN_TOPP_FIRST_FORM"); return e && e[0] == '1'; }();   // A/B runs
  hipStream_t st = (hipStream_t)stream;
  if (V + 3 > TOPP_CAP) {
    hipLaunchKernelGGL(topp_kernel, dim3((unsigned)rows), dim3(256), 0, st, a);
  } else if (!first_form && rows < (1 << 24) && workspace && workspace_bytes >= (size_t)(rows + 1) * sizeof(int) && (uintptr_t)workspace % 4 == 0 &&
             ((uintptr_t)noise - (uintptr_t)probs) % 16 == 0 && ((noise_row_stride - row_stride) * 4) % 16 == 0) {
    // list form; the rows it cannot do are listed in the workspace and redone by the first form
    hipLaunchKernelGGL(topp_reset_kernel, dim3(1), dim3(1), 0, st, (int*)workspace);
    hipLaunchKernelGGL(topp_list_kernel, dim3((unsigned)rows), dim3(TOPP_THREADS), 0, st, a, (int*)workspace);
    hipLaunchKernelGGL(topp_regs_todo_kernel, dim3((unsigned)(rows < 64 ? rows : 64)), dim3(TOPP_THREADS), 0, st, a, (const int*)workspace);
  } else {
    hipLaunchKernelGGL(topp_regs_kernel, dim3((unsigned)rows), dim3(TOPP_THREADS), 0, st, a);
  }
  BG_CHECK_LAUNCH("bg_topp_select");
  return BG_OK;
}
