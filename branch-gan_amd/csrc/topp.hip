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

}  // namespace bg

using namespace bg;

extern "C" size_t bg_topp_workspace_bytes(int64_t rows, int64_t V, int n) {
  (void)rows; (void)V; (void)n;
  return 0;
}

extern "C" int bg_topp_select(const float* probs, int64_t rows, int64_t V, int64_t row_stride, const float* noise,
                              int64_t noise_row_stride, const int64_t* blocked, int n_blocked, float top_p, int n,
                              int64_t* out_ids, float* out_keys, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  BG_REQUIRE(probs && noise && out_ids, "bg_topp_select: null probs/noise/out_ids");
  BG_REQUIRE(rows > 0 && rows < (1ll << 31) && V > 0 && V < (1ll << 31), "bg_topp_select: rows=%lld V=%lld", (long long)rows, (long long)V);
  BG_REQUIRE(n >= 1 && n <= BG_MAX_TOPN && n <= V, "bg_topp_select: n=%d must be in [1, min(%d, V)]", n, BG_MAX_TOPN);
  BG_REQUIRE(row_stride >= V && noise_row_stride >= V, "bg_topp_select: row stride < V");
  BG_REQUIRE(n_blocked >= 0 && n_blocked <= BG_MAX_BLOCKED && (n_blocked == 0 || blocked), "bg_topp_select: n_blocked=%d", n_blocked);
  BG_REQUIRE(top_p == top_p, "bg_topp_select: top_p is NaN");
  ToppArgs a{probs, noise, blocked, V, row_stride, noise_row_stride, n, n_blocked, top_p, out_ids, out_keys};
  hipLaunchKernelGGL(topp_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a);
  BG_CHECK_LAUNCH("bg_topp_select");
  return BG_OK;
}
