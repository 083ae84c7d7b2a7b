// Fused discriminator and generator-REINFORCE losses: one launch each way instead of ~150 tiny ATen launches.
//
// Replaces Models/Losses/DiscriminatorLoss.py:4-74, AdvantageCalculations.py:5-46, ReinforceLoss.py:21-52 and
// GeneratorLosses.py:84-131 (reference).  See include/branchgan.h: bg_disc_loss_* / bg_rl_loss_*.
// The tensors are KB-sized ((B, N, S) with N <= 64), so a single 1024-thread workgroup with wavefront
// reductions does the whole loss; the figure of merit is latency, not bandwidth.  The reference's zero / 1e-5
// padding of the K real branch columns up to S (GanSetup.py:151-167,202-207, MultiStepPrediction.py:59-65) is
// evaluated arithmetically, never materialised.
#include <algorithm>

#include "common.h"

namespace bg {

constexpr int LOSS_THREADS = 1024;

__device__ __forceinline__ float clog(float x) { return fmaxf(logf(x), -100.f); }  // ATen BCE clamps log at -100
__device__ __forceinline__ float bce(float x, float y) { return -(y * clog(x) + (1.f - y) * clog(1.f - x)); }
// ATen binary_cross_entropy_backward: (x - y) / max((1 - x) * x, 1e-12)
__device__ __forceinline__ float bce_grad(float x, float y) { return (x - y) / fmaxf((1.f - x) * x, 1e-12f); }

template <int NV, int THREADS = LOSS_THREADS>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sh /* [NV][16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) sh[i * 16 + wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float t = 0.f;
    for (int w = 0; w < THREADS / 64; ++w) t += sh[i * 16 + w];
    v[i] = t;
  }
  __syncthreads();
}

struct DiscArgs {
  bg_disc_loss_shape sh;
  const float *in, *alt, *mask, *txt_in, *txt_alt, *alt_alt;
  const int32_t *root, *child, *n_child;
  float* out;
  const float* g;
  float *d_in, *d_alt, *d_txt_in, *d_txt_alt, *d_alt_alt;
};

__device__ __forceinline__ float child_mean(const DiscArgs& a, const int32_t* list, int cnt, int b, int k) {
  // mean over child nodes of the alteration scores at (b, k); k >= K is the reference's zero padding
  if (k >= a.sh.K || cnt == 0) return 0.f;
  float s = 0.f;
  for (int c = 0; c < cnt; ++c) s += a.alt[((int64_t)b * a.sh.N + list[c]) * a.sh.K + k];
  return s / (float)cnt;
}

template <bool BWD>
__global__ __launch_bounds__(LOSS_THREADS) void disc_loss_kernel(DiscArgs a) {
  __shared__ float red[5 * 16];
  __shared__ int n_nonleaf_sh;
  const bg_disc_loss_shape& s = a.sh;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int c = 0;
    for (int i = 0; i < s.N; ++i) c += a.n_child[i] > 0;
    n_nonleaf_sh = c;
  }
  __syncthreads();
  const int n_nonleaf = n_nonleaf_sh;
  const float inv_in = 1.f / ((float)s.B * (s.S - 1));
  const float inv_alt = 1.f / ((float)s.B * s.n_update * s.S);
  const float inv_node = 1.f / ((float)s.B * s.S);
  const float inv_terms = 1.f / (float)(n_nonleaf + 1);
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // input, alt, vh-input, root, sum of node losses
  float g_in = 0.f, g_alt = 0.f, g_vh = 0.f;
  if constexpr (BWD) { g_in = a.g[0]; g_alt = a.g[1]; g_vh = a.g[2]; }

  // text columns: input loss, value-head input loss, root loss
  for (int i = tid; i < s.B * s.S; i += LOSS_THREADS) {
    const int b = i / s.S, t = i % s.S;
    const float x = a.in[i];
    if (t >= 1) {
      if constexpr (BWD) a.d_in[i] = g_in * inv_in * bce_grad(x, 1.f);
      else acc[0] += -clog(x);
    } else if constexpr (BWD) a.d_in[i] = 0.f;
    if (t < s.S - 1) {
      const float y_in = a.in[i + 1];
      const float y_root = child_mean(a, a.root, s.n_root, b, t + 1);  // quirk 10: text column t vs branch column t+1
      if constexpr (BWD) {
        a.d_txt_in[i] = g_vh * 0.5f * inv_in * bce_grad(a.txt_in[i], y_in);
        a.d_txt_alt[i] = g_vh * 0.5f * inv_terms * inv_in * bce_grad(a.txt_alt[i], y_root);
      } else {
        acc[2] += bce(a.txt_in[i], y_in);
        acc[3] += bce(a.txt_alt[i], y_root);
      }
    } else if constexpr (BWD) { a.d_txt_in[i] = 0.f; a.d_txt_alt[i] = 0.f; }
  }
  // branch columns: alteration loss and per-node value-head losses
  for (int i = tid; i < s.B * s.N * s.K; i += LOSS_THREADS) {
    const int k = i % s.K, n = (i / s.K) % s.N, b = i / (s.K * s.N);
    const float m = a.mask[n * s.K + k];
    const float x = a.alt[i];
    if (n < s.n_update) {
      if constexpr (BWD) a.d_alt[i] = g_alt * inv_alt * m * bce_grad(x, 0.f);
      else acc[1] += m * (-clog(1.f - x));
    } else if constexpr (BWD) a.d_alt[i] = 0.f;
    const int cnt = a.n_child[n];
    const int64_t hi = ((int64_t)n * s.B + b) * s.K + k;  // alt_alt is node-major [N][B][K]
    if (cnt > 0) {
      const float y = child_mean(a, a.child + n * s.max_child, cnt, b, k);
      if constexpr (BWD) a.d_alt_alt[hi] = g_vh * 0.5f * inv_terms * inv_node * m * bce_grad(a.alt_alt[hi], y);
      else acc[4] += m * bce(a.alt_alt[hi], y);
    } else if constexpr (BWD) a.d_alt_alt[hi] = 0.f;
  }
  if constexpr (!BWD) {
    block_sum<5>(acc, red);
    if (tid == 0) {
      const float in_loss = acc[0] * inv_in, alt_loss = acc[1] * inv_alt, vh_in = acc[2] * inv_in;
      const float vh_alt = (acc[4] * inv_node + acc[3] * inv_in) * inv_terms;
      a.out[0] = in_loss; a.out[1] = alt_loss; a.out[2] = vh_in; a.out[3] = vh_alt; a.out[4] = 0.5f * (vh_in + vh_alt);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
struct RlArgs {
  bg_rl_loss_shape sh;
  const float *in, *alt, *in_head, *alt_head, *label_lp, *alt_lp, *mask;
  float* out;
  const float* g;
  float *d_label, *d_alt;
  float* diag;  // optional fp32 [BG_RL_DIAG_COUNT]: _calculateDebuggingStats (GeneratorLosses.py:34-81), forward only
};

struct Adv { float adv; bool zone_c; };

__device__ __forceinline__ Adv advantage(const bg_rl_loss_shape& s, float score, float in_h, float alt_h, bool ignore) {
  const bool zone_a = score >= in_h, above = score >= alt_h;
  const bool zone_c = !above, zone_b = zone_a != above;
  const float den = fmaxf(fabsf(in_h - alt_h) + 1e-5f, 1e-5f);
  const float scaled = (score - alt_h) / den;
  float adv = (zone_a ? 1.f : 0.f) * s.wA + scaled * (zone_b ? 1.f : 0.f) * s.wB + (zone_c ? 1.f : 0.f) * s.wC;
  if (ignore && alt_h >= in_h) adv = 0.f;  // IncorrectValueHeadsPolicy.IGNORE_SAMPLE
  return {adv, zone_c};
}

// value and d/d(input) of the mirrored REINFORCE term for one element; `v` is a log-prob or a prob
__device__ __forceinline__ void rl_term(bool probs_input, float v, Adv ad, float& loss, float& dv) {
  const float p = probs_input ? v : __expf(v);
  const float pc = fminf(fmaxf(p, 1e-6f), 1.f - 1e-6f);
  const bool pass = (p >= 1e-6f) && (p <= 1.f - 1e-6f);
  float dp;
  if (ad.zone_c) { loss = -logf(1.f - pc) * ad.adv; dp = ad.adv / (1.f - pc); }
  else { loss = -logf(pc) * ad.adv; dp = -ad.adv / pc; }
  dv = pass ? (probs_input ? dp : dp * p) : 0.f;
}

template <bool BWD>
__global__ __launch_bounds__(LOSS_THREADS) void rl_loss_kernel(RlArgs a) {
  __shared__ float red[3 * 16];
  const bg_rl_loss_shape& s = a.sh;
  const int tid = threadIdx.x;
  float acc[3] = {0.f, 0.f, 0.f};  // input sum, alteration sum, mask sum
  // the alteration normaliser is a plain number, needed by the backward too
  for (int i = tid; i < s.N * s.K; i += LOSS_THREADS) acc[2] += a.mask ? a.mask[i] : 0.f;
  block_sum<3>(acc, red);
  const float mask_sum = acc[2];
  acc[0] = acc[1] = acc[2] = 0.f;
  const float inv_in = 1.f / ((float)s.B * (s.S - 1));
  const float inv_alt = s.use_depth_mask ? 1.f / mask_sum : 1.f / ((float)s.B * s.N * (s.S - 1));
  float g_in = 0.f, g_alt = 0.f;
  if constexpr (BWD) { g_in = a.g[0]; g_alt = a.g[1]; }

  for (int i = tid; i < s.B * (s.S - 1); i += LOSS_THREADS) {
    const int b = i / (s.S - 1), t = i % (s.S - 1);
    const float score = a.in[b * s.S + t + 1];  // input scores shifted by one (AdvantageCalculations.py:16)
    const Adv ad = advantage(s, score, a.in_head[b * s.S + t], a.alt_head[b * s.S + t], s.ignore_input != 0);
    float loss, dv;
    rl_term(s.probs_input != 0, a.label_lp[i], ad, loss, dv);
    if constexpr (BWD) a.d_label[i] = g_in * inv_in * dv;
    else acc[0] += loss;
  }
  const int cols = s.use_depth_mask ? s.K : (s.S - 1);  // without a depth mask the 1e-5-padded columns count too
  for (int i = tid; i < s.B * s.N * cols; i += LOSS_THREADS) {
    const int j = i % cols, n = (i / cols) % s.N, b = i / (cols * s.N);
    const bool real = j < s.K;
    const float score = real ? a.alt[((int64_t)b * s.N + n) * s.K + j] : 0.f;
    // the TEXT value heads at column j are compared with branch column j (reference behaviour)
    const Adv ad = advantage(s, score, a.in_head[b * s.S + j], a.alt_head[b * s.S + j], s.ignore_alt != 0);
    const float w = s.use_depth_mask ? a.mask[n * s.K + j] : 1.f;
    float loss, dv;
    if (real) rl_term(s.probs_input != 0, a.alt_lp[((int64_t)b * s.N + n) * s.K + j], ad, loss, dv);
    else rl_term(true, 1e-5f, ad, loss, dv);
    if constexpr (BWD) { if (real) a.d_alt[((int64_t)b * s.N + n) * s.K + j] = g_alt * inv_alt * w * dv; }
    else acc[1] += w * loss;
  }
  if constexpr (BWD) {
    if (!s.use_depth_mask && s.K > s.S - 1)  // column S-1 is excluded by the reference's [:, :, :-1]
      for (int i = tid; i < s.B * s.N; i += LOSS_THREADS)
        for (int j = s.S - 1; j < s.K; ++j) a.d_alt[(int64_t)i * s.K + j] = 0.f;
  } else {
    block_sum<3>(acc, red);
    if (tid == 0) { a.out[0] = acc[0] * inv_in; a.out[1] = acc[1] * inv_alt; }
  }
}

// _calculateDebuggingStats (GeneratorLosses.py:34-81): the twenty diagnostics the reference's training loop logs beside the
// generator loss, in the reference's dict order.  They are sums over the SAME per-element quantities rl_loss_kernel forms,
// but over the reference's full (B, 1+N, S) extent: the input row keeps its last column (shifted score 0) for the advantage
// means and zone factors and drops it ([:, :, :-1]) for the loss means; the alteration rows keep the S - K padded columns
// (score 0, probability 1e-5) and are NOT weighted by the depth mask, except in the three zone factors.
// A zone whose mask is empty reports 0 (the reference's `if x.sum() > 0 else 0`).
constexpr int RL_DIAG_ACC = 26;
// 512 threads, not the loss kernels' 1024: 26 accumulators beside four elements' operands in flight do not fit the 128 registers
// a lane has at 1024 (310 of them spilled, 1.2 KB of scratch per lane: 82 us at configs[1], four times the loss kernel's time)
constexpr int RL_DIAG_THREADS = 512;

// One workgroup, a few KB of input: the kernel's time is its dependent global round trips, so every loop below takes its
// elements U at a time with all their loads issued before the first use (8 rounds of loads -> 3: 87 -> ~20 us at configs[1]).
// The padded columns j >= K of the alteration rows do not depend on the node: once per (b, j), weighted by N.
__global__ __launch_bounds__(RL_DIAG_THREADS) void rl_diag_kernel(RlArgs a) {
  __shared__ float red[RL_DIAG_ACC * 16];
  const bg_rl_loss_shape& s = a.sh;
  const int tid = threadIdx.x;
  constexpr int U = 4;
  // [z]: input loss sum, input count over S-1, input advantage sum, input count over S  (z = A, B, C)
  // [12 + z]: alteration loss sum, count, advantage sum, count under the depth mask;  [24] heads correct, [25] input correct
  float acc[RL_DIAG_ACC];
#pragma unroll
  for (int i = 0; i < RL_DIAG_ACC; ++i) acc[i] = 0.f;
  auto add_alt = [&](float score, float in_h, float alt_h, bool real, float lp, float m, float weight) {
    const Adv ad = advantage(s, score, in_h, alt_h, s.ignore_alt != 0);
    const bool za = score >= in_h, above = score >= alt_h;
    const bool zone[3] = {za, za != above, !above};
    float loss, dv;
    if (real) rl_term(s.probs_input != 0, lp, ad, loss, dv);
    else rl_term(true, 1e-5f, ad, loss, dv);
#pragma unroll
    for (int z = 0; z < 3; ++z)
      if (zone[z]) { acc[12 + 4 * z] += weight * loss; acc[12 + 4 * z + 1] += weight; acc[12 + 4 * z + 2] += weight * ad.adv; acc[12 + 4 * z + 3] += m; }
  };
  // ---- the input row: (b, t) over B x S -------------------------------------------------------------------------------------
  const int n_in_el = s.B * s.S;
  for (int i0 = tid; i0 < n_in_el; i0 += U * RL_DIAG_THREADS) {
    float in_h[U], alt_h[U], score[U], own[U], lp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u * RL_DIAG_THREADS, n_in_el - 1);
      const int b = i / s.S, t = i % s.S;
      in_h[u] = a.in_head[i]; alt_h[u] = a.alt_head[i]; own[u] = a.in[i];
      score[u] = a.in[min(i + 1, n_in_el - 1)];
      lp[u] = a.label_lp[b * (s.S - 1) + min(t, s.S - 2)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RL_DIAG_THREADS;
      if (i >= n_in_el) break;
      const int t = i % s.S;
      const float sc = t + 1 < s.S ? score[u] : 0.f;
      const Adv ad = advantage(s, sc, in_h[u], alt_h[u], s.ignore_input != 0);
      const bool za = sc >= in_h[u], above = sc >= alt_h[u];
      const bool zone[3] = {za, za != above, !above};
      float loss = 0.f, dv;
      if (t < s.S - 1) rl_term(s.probs_input != 0, lp[u], ad, loss, dv);
#pragma unroll
      for (int z = 0; z < 3; ++z)
        if (zone[z]) {
          if (t < s.S - 1) { acc[4 * z] += loss; acc[4 * z + 1] += 1.f; }
          acc[4 * z + 2] += ad.adv;
          acc[4 * z + 3] += 1.f;
        }
      acc[24] += alt_h[u] >= in_h[u] ? 0.f : 1.f;
      acc[25] += own[u] >= alt_h[u] ? 1.f : 0.f;
    }
  }
  // ---- the alteration rows, real columns: (b, n, j < K) --------------------------------------------------------------------------
  const int n_real = s.B * s.N * s.K;
  for (int i0 = tid; i0 < n_real; i0 += U * RL_DIAG_THREADS) {
    float in_h[U], alt_h[U], score[U], lp[U], m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = min(i0 + u * RL_DIAG_THREADS, n_real - 1);
      const int j = e % s.K, n = (e / s.K) % s.N, b = e / (s.K * s.N);
      score[u] = a.alt[e]; lp[u] = a.alt_lp[e];
      in_h[u] = a.in_head[b * s.S + j]; alt_h[u] = a.alt_head[b * s.S + j];
      m[u] = s.use_depth_mask ? a.mask[n * s.K + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * RL_DIAG_THREADS < n_real) add_alt(score[u], in_h[u], alt_h[u], true, lp[u], m[u], 1.f);
  }
  // ---- their padded columns (score 0, probability 1e-5, outside the depth mask): (b, K <= j < S), the same for every node ----------
  const int pad = s.S - s.K, n_pad = s.B * pad;
  for (int i0 = tid; i0 < n_pad; i0 += U * RL_DIAG_THREADS) {
    float in_h[U], alt_h[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u * RL_DIAG_THREADS, n_pad - 1);
      const int b = i / pad, j = s.K + i % pad;
      in_h[u] = a.in_head[b * s.S + j]; alt_h[u] = a.alt_head[b * s.S + j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * RL_DIAG_THREADS < n_pad) add_alt(0.f, in_h[u], alt_h[u], false, 0.f, 0.f, (float)s.N);
  }
  float mask_sum = 0.f;
  if (s.use_depth_mask)
    for (int i = tid; i < s.N * s.K; i += RL_DIAG_THREADS) mask_sum += a.mask[i];
  block_sum<RL_DIAG_ACC, RL_DIAG_THREADS>(acc, red);
  float one[1] = {mask_sum};
  block_sum<1, RL_DIAG_THREADS>(one, red);
  if (tid == 0) {
    const float n_in = (float)s.B * s.S, n_alt = (float)s.B * s.N * s.S;
    auto ratio = [](float num, float den) { return den > 0.f ? num / den : 0.f; };
    float* d = a.diag;
    for (int z = 0; z < 3; ++z) {
      d[z] = ratio(acc[4 * z], acc[4 * z + 1]);                 // Rl-Loss-{A,B,C}-Input
      d[3 + z] = ratio(acc[12 + 4 * z], acc[12 + 4 * z + 1]);   // Rl-Loss-{A,B,C}-Alt
      d[6 + z] = ratio(acc[4 * z + 2], acc[4 * z + 3]);         // Rl-Advantage-{A,B,C}-Input
      d[9 + z] = ratio(acc[12 + 4 * z + 2], acc[12 + 4 * z + 1]);  // Rl-Advantage-{A,B,C}-Alt
      d[12 + z] = acc[4 * z + 3] / n_in;                        // RL-Zone{A,B,C}-Input-Factor
      d[15 + z] = s.use_depth_mask ? ratio(acc[12 + 4 * z + 3], one[0] * (float)s.B) : acc[12 + 4 * z + 1] / n_alt;
    }
    d[18] = acc[24] / n_in;                                     // RL-CorrectValueHeads-Factor
    d[19] = acc[25] / n_in;                                     // RL-CorrectInput-Factor
  }
}

// ---------------------------------------------------------------------------------------------------
// The element-wise pieces of the generator loss under the reference's own function boundaries (bg_dual_head_advantages,
// bg_reinforce_terms_*): the fused rl_loss_kernel above is what the train step runs; these serve callers of
// AdvantageCalculations.dualHeadAdvantages / ReinforceLoss.reinforceLoss / mirroredNegativeReinforceLoss.
struct AdvArgs {
  const float *in, *alt, *in_head, *alt_head;
  int B, N, S;
  float wA, wB, wC, eps, min_den;
  int ignore_in, ignore_alt;
  float *adv, *za, *zb, *zc, *ok, *in_ok;
};

__global__ __launch_bounds__(256) void dual_head_advantages_kernel(AdvArgs a) {
  const int64_t total = (int64_t)a.B * (1 + a.N) * a.S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % a.S), r = (int)((i / a.S) % (1 + a.N)), b = (int)(i / ((int64_t)a.S * (1 + a.N)));
    const float in_h = a.in_head[b * a.S + t], alt_h = a.alt_head[b * a.S + t];
    // row 0: the input scores shifted left by one with a zero appended (AdvantageCalculations.py:14-17)
    const float score = r == 0 ? (t + 1 < a.S ? a.in[b * a.S + t + 1] : 0.f) : a.alt[((int64_t)b * a.N + (r - 1)) * a.S + t];
    const bool zone_a = score >= in_h, above = score >= alt_h;
    const bool zone_c = !above, zone_b = zone_a != above;
    const float den = fmaxf(fabsf(in_h - alt_h) + a.eps, a.min_den);
    const float scaled = (score - alt_h) / den;
    float adv = (zone_a ? 1.f : 0.f) * a.wA + scaled * (zone_b ? 1.f : 0.f) * a.wB + (zone_c ? 1.f : 0.f) * a.wC;
    const float heads_ok = alt_h >= in_h ? 0.f : 1.f;
    if (r == 0 ? a.ignore_in : a.ignore_alt) adv *= heads_ok;
    a.adv[i] = adv;
    if (a.za) a.za[i] = zone_a ? 1.f : 0.f;
    if (a.zb) a.zb[i] = zone_b ? 1.f : 0.f;
    if (a.zc) a.zc[i] = zone_c ? 1.f : 0.f;
    if (r == 0) {
      if (a.ok) a.ok[b * a.S + t] = heads_ok;
      if (a.in_ok) a.in_ok[b * a.S + t] = a.in[b * a.S + t] >= alt_h ? 1.f : 0.f;   // un-shifted input score (:12)
    }
  }
}

// loss = (1 - m) * normal + m * mirrored, normal = -log(pc) * adv [* limit factor], mirrored = -log(1 - pc) * adv,
// pc = clamp(p, eps, 1 - eps); m = mask_c (NULL: 0, the plain reinforceLoss).  d/dp is zero outside the clamp.
template <bool BWD>
__global__ __launch_bounds__(256) void reinforce_terms_kernel(const float* probs, const float* adv, const float* mask_c, int64_t n,
                                                              int use_min_prob, float neg_push, float eps, const float* g,
                                                              float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float p = probs[i], ad = adv[i], m = mask_c ? mask_c[i] : 0.f;
    const float pc = fminf(fmaxf(p, eps), 1.f - eps);
    const bool pass = (p >= eps) && (p <= 1.f - eps);
    float factor = 1.f;
    if (use_min_prob) {            // calculateMinimumProbLimit (ReinforceLoss.py:11-18): rewardMask + negRewardMask
      const float reward = ad >= 0.f ? 1.f : 0.f;
      const float neg_p = pc * (1.f - reward);
      factor = reward + (neg_p >= neg_push ? 1.f - reward : reward);
    }
    if constexpr (BWD) {
      const float d = (1.f - m) * (-ad / pc) * factor + m * (ad / (1.f - pc));
      out[i] = pass ? g[i] * d : 0.f;
    } else {
      out[i] = (1.f - m) * (-logf(pc) * ad) * factor + m * (-logf(1.f - pc) * ad);
    }
  }
}

static int check_disc(const bg_disc_loss_shape* s, const char* who) {
  BG_REQUIRE(s, "%s: null shape", who);
  BG_REQUIRE(s->B > 0 && s->S > 1 && s->N > 0 && s->K > 0 && s->K <= s->S, "%s: B=%d S=%d N=%d K=%d", who, s->B, s->S, s->N, s->K);
  BG_REQUIRE(s->n_update > 0 && s->n_update <= s->N, "%s: n_update=%d", who, s->n_update);
  BG_REQUIRE(s->n_root > 0 && s->max_child >= 1, "%s: n_root=%d max_child=%d", who, s->n_root, s->max_child);
  BG_REQUIRE((int64_t)s->B * s->N * s->S < (1ll << 30), "%s: too large", who);
  return BG_OK;
}
static int check_rl(const bg_rl_loss_shape* s, const char* who) {
  BG_REQUIRE(s, "%s: null shape", who);
  BG_REQUIRE(s->B > 0 && s->S > 1 && s->N > 0 && s->K > 0 && s->K <= s->S, "%s: B=%d S=%d N=%d K=%d", who, s->B, s->S, s->N, s->K);
  BG_REQUIRE((int64_t)s->B * s->N * s->S < (1ll << 30), "%s: too large", who);
  return BG_OK;
}

}  // namespace bg

using namespace bg;

extern "C" int bg_disc_loss_fwd(const bg_disc_loss_shape* shape, const float* input_scores, const float* alt_scores,
                                const float* depth_mask, const float* txt_in, const float* txt_alt, const float* alt_alt,
                                const int32_t* root_links, const int32_t* child, const int32_t* n_child, float* out_losses,
                                void* stream) {
  if (int e = check_disc(shape, "bg_disc_loss_fwd")) return e;
  BG_REQUIRE(input_scores && alt_scores && depth_mask && txt_in && txt_alt && alt_alt && root_links && child && n_child && out_losses,
             "bg_disc_loss_fwd: null pointer");
  DiscArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape; a.in = input_scores; a.alt = alt_scores; a.mask = depth_mask; a.txt_in = txt_in; a.txt_alt = txt_alt;
  a.alt_alt = alt_alt; a.root = root_links; a.child = child; a.n_child = n_child; a.out = out_losses;
  hipLaunchKernelGGL(disc_loss_kernel<false>, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a);
  BG_CHECK_LAUNCH("bg_disc_loss_fwd");
  return BG_OK;
}

extern "C" int bg_disc_loss_bwd(const bg_disc_loss_shape* shape, const float* input_scores, const float* alt_scores,
                                const float* depth_mask, const float* txt_in, const float* txt_alt, const float* alt_alt,
                                const int32_t* root_links, const int32_t* child, const int32_t* n_child, const float* g_losses,
                                float* d_input_scores, float* d_alt_scores, float* d_txt_in, float* d_txt_alt, float* d_alt_alt,
                                void* stream) {
  if (int e = check_disc(shape, "bg_disc_loss_bwd")) return e;
  BG_REQUIRE(input_scores && alt_scores && depth_mask && txt_in && txt_alt && alt_alt && root_links && child && n_child && g_losses &&
                 d_input_scores && d_alt_scores && d_txt_in && d_txt_alt && d_alt_alt,
             "bg_disc_loss_bwd: null pointer");
  DiscArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape; a.in = input_scores; a.alt = alt_scores; a.mask = depth_mask; a.txt_in = txt_in; a.txt_alt = txt_alt;
  a.alt_alt = alt_alt; a.root = root_links; a.child = child; a.n_child = n_child; a.g = g_losses;
  a.d_in = d_input_scores; a.d_alt = d_alt_scores; a.d_txt_in = d_txt_in; a.d_txt_alt = d_txt_alt; a.d_alt_alt = d_alt_alt;
  hipLaunchKernelGGL(disc_loss_kernel<true>, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a);
  BG_CHECK_LAUNCH("bg_disc_loss_bwd");
  return BG_OK;
}

extern "C" int bg_rl_loss_fwd(const bg_rl_loss_shape* shape, const float* input_scores, const float* alt_scores,
                              const float* in_head, const float* alt_head, const float* label_logp, const float* alt_logp,
                              const float* depth_mask, float* out_losses, float* out_diag, void* stream) {
  if (int e = check_rl(shape, "bg_rl_loss_fwd")) return e;
  BG_REQUIRE(input_scores && alt_scores && in_head && alt_head && label_logp && alt_logp && out_losses && (depth_mask || !shape->use_depth_mask),
             "bg_rl_loss_fwd: null pointer");
  RlArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape; a.in = input_scores; a.alt = alt_scores; a.in_head = in_head; a.alt_head = alt_head; a.label_lp = label_logp;
  a.alt_lp = alt_logp; a.mask = depth_mask; a.out = out_losses; a.diag = out_diag;
  hipLaunchKernelGGL(rl_loss_kernel<false>, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a);
  BG_CHECK_LAUNCH("bg_rl_loss_fwd");
  if (out_diag) {
    hipLaunchKernelGGL(rl_diag_kernel, dim3(1), dim3(RL_DIAG_THREADS), 0, (hipStream_t)stream, a);
    BG_CHECK_LAUNCH("bg_rl_loss_fwd (diagnostics)");
  }
  return BG_OK;
}

extern "C" int bg_rl_loss_bwd(const bg_rl_loss_shape* shape, const float* input_scores, const float* alt_scores,
                              const float* in_head, const float* alt_head, const float* label_logp, const float* alt_logp,
                              const float* depth_mask, const float* g_losses, float* d_label_logp, float* d_alt_logp,
                              void* stream) {
  if (int e = check_rl(shape, "bg_rl_loss_bwd")) return e;
  BG_REQUIRE(input_scores && alt_scores && in_head && alt_head && label_logp && alt_logp && g_losses && d_label_logp && d_alt_logp &&
                 (depth_mask || !shape->use_depth_mask), "bg_rl_loss_bwd: null pointer");
  RlArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape; a.in = input_scores; a.alt = alt_scores; a.in_head = in_head; a.alt_head = alt_head; a.label_lp = label_logp;
  a.alt_lp = alt_logp; a.mask = depth_mask; a.g = g_losses; a.d_label = d_label_logp; a.d_alt = d_alt_logp;
  hipLaunchKernelGGL(rl_loss_kernel<true>, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a);
  BG_CHECK_LAUNCH("bg_rl_loss_bwd");
  return BG_OK;
}

extern "C" int bg_dual_head_advantages(const float* input_scores, const float* alt_scores, const float* in_head, const float* alt_head,
                                       int B, int N, int S, float wA, float wB, float wC, int ignore_input, int ignore_alt,
                                       float epsilon, float min_denominator, float* advantages, float* zone_a, float* zone_b,
                                       float* zone_c, float* heads_correct, float* input_correct, void* stream) {
  BG_REQUIRE(input_scores && alt_scores && in_head && alt_head && advantages, "bg_dual_head_advantages: null pointer");
  BG_REQUIRE(B > 0 && N > 0 && S > 0 && (int64_t)B * (1 + N) * S < (1ll << 31), "bg_dual_head_advantages: B=%d N=%d S=%d", B, N, S);
  AdvArgs a{input_scores, alt_scores, in_head, alt_head, B, N, S, wA, wB, wC, epsilon, min_denominator, ignore_input, ignore_alt,
            advantages, zone_a, zone_b, zone_c, heads_correct, input_correct};
  const int64_t total = (int64_t)B * (1 + N) * S;
  hipLaunchKernelGGL(dual_head_advantages_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 2048)), dim3(256), 0, (hipStream_t)stream, a);
  BG_CHECK_LAUNCH("bg_dual_head_advantages");
  return BG_OK;
}

static int reinforce_terms(const char* who, bool bwd, const float* probs, const float* adv, const float* mask_c, int64_t n,
                           int use_min_prob, float neg_push, float eps, const float* g, float* out, void* stream) {
  BG_REQUIRE(probs && adv && out && (!bwd || g), "%s: null pointer", who);
  BG_REQUIRE(n > 0 && n < (1ll << 40), "%s: n=%lld", who, (long long)n);
  BG_REQUIRE(eps > 0.f && eps < 0.5f, "%s: eps=%g", who, eps);
  const dim3 grid((unsigned)std::min<int64_t>(cdiv(n, 256), 2048));
  if (bwd) hipLaunchKernelGGL(reinforce_terms_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, probs, adv, mask_c, n, use_min_prob, neg_push, eps, g, out);
  else hipLaunchKernelGGL(reinforce_terms_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, probs, adv, mask_c, n, use_min_prob, neg_push, eps, g, out);
  BG_CHECK_LAUNCH(who);
  return BG_OK;
}

extern "C" int bg_reinforce_terms_fwd(const float* probs, const float* advantages, const float* mask_zone_c, int64_t n,
                                      int use_min_prob_limit, float neg_push_prob_limit, float eps, float* loss, void* stream) {
  return reinforce_terms("bg_reinforce_terms_fwd", false, probs, advantages, mask_zone_c, n, use_min_prob_limit, neg_push_prob_limit, eps,
                         nullptr, loss, stream);
}

extern "C" int bg_reinforce_terms_bwd(const float* probs, const float* advantages, const float* mask_zone_c, int64_t n,
                                      int use_min_prob_limit, float neg_push_prob_limit, float eps, const float* g_loss,
                                      float* d_probs, void* stream) {
  return reinforce_terms("bg_reinforce_terms_bwd", true, probs, advantages, mask_zone_c, n, use_min_prob_limit, neg_push_prob_limit, eps,
                         g_loss, d_probs, stream);
}
