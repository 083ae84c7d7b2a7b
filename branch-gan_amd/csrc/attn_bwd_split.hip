// Branch attention backward, work-split form (round 2).
//
// The kernels of attn_bwd.hip give every wave 32 queries (dq) or 32 keys (dkv) and let it walk ALL the tiles on the other
// axis alone: at the BASELINE shapes that is 192-576 workgroups of one wave per SIMD, each a chain of 8-29 dependent tile
// iterations (LDS reads -> MFMA -> exp -> transposed reads -> MFMA, 1.6-2.9 us a tile; profiles/r1_n_attn_scaling.txt).
// Here a workgroup owns ONE 32-row tile and its four waves split the other axis - the key tiles of a query tile (dq), the
// query tiles that reach a key tile (dkv) - each wave staging its own tiles in a private LDS region (no workgroup barrier in
// the loop), with the next tile's global loads in flight behind the current tile's MFMAs.  The backward recomputes P from
// the saved log-sum-exp, so the four partial gradients simply ADD: they meet in LDS once, in a fixed order (deterministic,
// no atomics), and leave as whole 16-byte vectors.  4x the workgroups, a 4x shorter chain per wave.
// Same mathematics and the same per-element operations as attn_bwd.hip; only the summation is regrouped.
#include <stdlib.h>

#include "attn_bwd_slots.h"
#include "attn_dropout.h"

namespace bg {

// BG_ATTN_TRACE (benchmarks/probes/attn_trace.py builds a second library with it): every wave leaves wall-clock stamps
// (100 MHz) at its phase boundaries, with a full s_waitcnt before each so that the stamp closes the phase's loads.
#ifdef BG_ATTN_TRACE
#define BG_TRACE_WAVES 8192
__device__ unsigned long long g_attn_trace[2][BG_TRACE_WAVES][16];
#define BG_TRACE(kind, slot)                                                                       \
  do {                                                                                             \
    __builtin_amdgcn_s_waitcnt(0);                                                                 \
    const unsigned w_ = (unsigned)(bh * (int)n_xtiles + tile_x) * ATT_WAVES + (threadIdx.x >> 6);   \
    if ((threadIdx.x & 63) == 0 && w_ < BG_TRACE_WAVES) g_attn_trace[kind][w_][slot] = wall_clock64(); \
  } while (0)
#define BG_TRACE_VAL(kind, slot, v)                                                                \
  do {                                                                                             \
    const unsigned w_ = (unsigned)(bh * (int)n_xtiles + tile_x) * ATT_WAVES + (threadIdx.x >> 6);   \
    if ((threadIdx.x & 63) == 0 && w_ < BG_TRACE_WAVES) g_attn_trace[kind][w_][slot] = (v);        \
  } while (0)
#else
#define BG_TRACE(kind, slot)
#define BG_TRACE_VAL(kind, slot, v)
#endif

template <typename T, int DH, int DPASS> struct BwdSplitLds {
  using L = Lds<T, DH>;
  static constexpr size_t TILE_BYTES = ((size_t)L::ROW_ELEMS * sizeof(T) + 15) / 16 * 16;
  static constexpr size_t STAT_BYTES = 4 * ATT_TILE * 4;                               // lse, delta, ctx (and dropout row key) of a gathered query tile
  static constexpr size_t WAVE_BYTES = 2 * TILE_BYTES + STAT_BYTES;
  static constexpr size_t MERGE_BYTES = (size_t)ATT_WAVES * DPASS * ATT_TILE * sizeof(float);
  static constexpr size_t STAGE_BYTES = (ATT_WAVES * WAVE_BYTES > MERGE_BYTES ? ATT_WAVES * WAVE_BYTES : MERGE_BYTES);
  // The workgroup's OWN tile (Q and dO of its 32 queries; K and V of its 32 keys) is fetched once, by all 256 threads in
  // whole 16-byte row chunks, and parked behind the staging area: a wave's operand fragments then come from LDS.  A wave
  // that reads its row straight from global memory (RowFrag::load on a global pointer) touches 32 different cache lines per
  // instruction, and four waves do it four times over: measured (benchmarks/probes/attn_trace.py), 5-7 us of a 13 us
  // workgroup life at BASELINE configs[1] went into that prologue.  Where the two extra tiles do not fit (bf16 Dh 256,
  // fp32 Dh 128) the fragments stay on the global path.
  static constexpr size_t OWN_BYTES = 2 * TILE_BYTES + ATT_TILE * 4;
  static constexpr bool OWN = STAGE_BYTES + OWN_BYTES + 20 * 1024 <= 160 * 1024;
  static constexpr size_t BYTES = STAGE_BYTES + (OWN ? OWN_BYTES : 0);
  // bf16 Dh 256 (round 3): no room to KEEP the own tile in LDS, but the same coalesced fetch passes THROUGH the staging area
  // before the loop needs it - every lane then lifts its row's fragments into registers (RowFrag::IN_REGS) and a second
  // barrier hands the area back.  The trace at configs[3] showed 7-10 us of a 25 us workgroup life in the per-lane row reads
  // this replaces (each load instruction touching 32-64 cache lines, four waves reading the same rows).
  static constexpr bool PARK = !OWN && sizeof(T) == 2;
};

// Sum of the four waves' partial [d][row] tiles in `mo` (fp32, [wave][DPASS][32]) for this thread's (row, feature group).
template <int DPASS, int PER>
__device__ __forceinline__ void merge4(const float* mo, int row, int grp, float (&acc)[PER]) {
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) v += mo[(w * DPASS + grp * PER + j) * ATT_TILE + row];
    acc[j] = v;
  }
}

template <typename T, int PER>
struct VecIn {                                            // PER incoming values of a row, requested ahead of their use
  typedef typename std::conditional<sizeof(T) == 4, f32x4, bf16x4>::type V;
  V v[PER / 4];
  bool has;
  __device__ __forceinline__ void load(const T* p, bool valid) {       // p null: nothing comes in
    has = p != nullptr;
#pragma unroll
    for (int c = 0; c < PER / 4; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[c][j] = (T)0.f;
    if (has && valid) {
#pragma unroll
      for (int c = 0; c < PER / 4; ++c) v[c] = *(const V*)(p + 4 * c);
    }
  }
};

template <typename T, int PER>
__device__ __forceinline__ void store_vec(T* p, const float (&acc)[PER], const VecIn<T, PER>& in) {
#pragma unroll
  for (int j0 = 0; j0 < PER; j0 += 4) {
    float x[4] = {acc[j0], acc[j0 + 1], acc[j0 + 2], acc[j0 + 3]};
    if (in.has) {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] += (float)in.v[j0 / 4][j];
    }
    typename VecIn<T, PER>::V o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (T)x[j];
    *(typename VecIn<T, PER>::V*)(p + j0) = o;
  }
}

template <typename T, int PER>
__device__ __forceinline__ void store_vec(T* p, const float (&acc)[PER]) {
  VecIn<T, PER> none;
  none.load(nullptr, false);
  store_vec<T, PER>(p, acc, none);
}

// ---------------------------------------------------------------------------------------------------
// dQ (and delta): one 32-query tile per workgroup, the key tiles dealt round-robin to the four waves.
// DROP (all three tile functions): dropout on the attention probabilities (attn_dropout.h).  With M the keep mask scaled by
// 1 / (1 - p):  O = (P o M) V, so dV = (P o M)^T dO, dP = (dO V^T) o M, dS = P o (dP - delta) with delta = rowsum(dO o O) as before.
template <typename T, int DH, bool DROP>
__device__ __forceinline__ void bwd_dq_tile(const AttnArgs& a, char* smem, int tile_x, int bh) {
  using SL = BwdSplitLds<T, DH, DH>;
  constexpr int NDT = DH / 32;
  const bg_attn_shape& sh = a.sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int R = sh.NQ * sh.K;
  [[maybe_unused]] const int n_xtiles = (R + ATT_TILE - 1) / ATT_TILE;
  BG_TRACE(0, 0);
  BG_TRACE(0, 1);
  T* k_row = (T*)(smem + wave * SL::WAVE_BYTES);
  T* v_row = (T*)(smem + wave * SL::WAVE_BYTES + SL::TILE_BYTES);
  const int b = bh / sh.H, hh = bh % sh.H;
  const int rq = tile_x * ATT_TILE + (lane & 31);
  const bool valid = rq < R;
  const int n = valid ? rq / sh.K : 0, kq = valid ? rq % sh.K : 0;
  const int64_t qoff = (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh + (int64_t)n * sh.q_sn + (int64_t)kq * sh.q_sk;
  const int64_t ooff = (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh + (int64_t)n * sh.o_sn + (int64_t)kq * sh.o_sk;
  const T* q_row = (const T*)a.q + qoff;
  const T* o_row = (const T*)a.out + ooff;
  const T* do_row = (const T*)a.dout + ooff;
  const T* kt = (const T*)a.k_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  const T* vt = (const T*)a.v_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  const int64_t stat = ((int64_t)bh * sh.NQ + n) * sh.K + kq;
  [[maybe_unused]] DropKeys drop;
  [[maybe_unused]] uint32_t row_key = 0;
  if constexpr (DROP) {
    drop = drop_keys(a, bh);
    row_key = drop_row_key(drop, (uint32_t)rq);
  }

  // Everything this wave needs first goes out in ONE round of loads: the context lengths, the workgroup's Q / dO / O tile
  // and - before the number of key tiles is known - key tile `wave` (dropped if the tile turns out to be beyond the context).
  int ctx = valid ? a.ctx_len[kq] : 0;
  const float lse = valid ? a.lse[stat] : 0.f;
  const int plen = (valid && sh.max_path) ? a.path_len[n] : 0;   // path tables may be NULL when max_path == 0
  const int slot_first = wave < sh.max_path ? a.path[n * sh.max_path + wave] : 0;   // this wave's first path entry, in the same round
  TileRegs<T, DH, 64> kreg, vreg;
  typename std::conditional<SL::OWN, LdsRowFrag<T, DH>, RowFrag<T, DH>>::type qf, dof;   // own tile in LDS: read where used
  float delta;
  // this lane's Q / dO rows for the branch keys: in LDS where the own tile stays there
  const T* q_path = q_row;
  const T* do_path = do_row;
  if constexpr (SL::OWN) {
    q_path = (const T*)(smem + SL::STAGE_BYTES) + (lane & 31) * Lds<T, DH>::RSTRIDE;
    do_path = (const T*)(smem + SL::STAGE_BYTES + SL::TILE_BYTES) + (lane & 31) * Lds<T, DH>::RSTRIDE;
  }
  if constexpr (SL::OWN || SL::PARK) {
    constexpr int VEC = 16 / sizeof(T), CPR = DH / VEC;
    // OWN: behind the staging area, for the kernel's life; PARK: in wave 0's staging region (delta in wave 1's) until the loop
    T* q_l = (T*)(smem + (SL::PARK ? 0 : SL::STAGE_BYTES));
    T* do_l = (T*)(smem + (SL::PARK ? 0 : SL::STAGE_BYTES) + SL::TILE_BYTES);
    float* delta_l = (float*)(smem + (SL::PARK ? SL::WAVE_BYTES : SL::STAGE_BYTES + 2 * SL::TILE_BYTES));
    const int row0 = tile_x * ATT_TILE;
    TileRegs<T, DH, 256> gq, gdo, go;
    gq.load_nk((const T*)a.q + (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh, sh.q_sn, sh.q_sk, sh.K, R, row0, tid);
    gdo.load_nk((const T*)a.dout + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh, sh.o_sn, sh.o_sk, sh.K, R, row0, tid);
    go.load_nk((const T*)a.out + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh, sh.o_sn, sh.o_sk, sh.K, R, row0, tid);
    if (wave * ATT_TILE < sh.S_text) {
      kreg.load(kt, sh.t_ss, wave * ATT_TILE, sh.S_text, lane);
      vreg.load(vt, sh.t_ss, wave * ATT_TILE, sh.S_text, lane);
    }
    // delta = rowsum(dO o O): the CPR threads that hold one row add up through the lanes
#pragma unroll
    for (int i = 0; i < TileRegs<T, DH, 256>::PER; ++i) {
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) part = fmaf((float)gdo.v[i][j], (float)go.v[i][j], part);
#pragma unroll
      for (int o = CPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
      const int c = tid + i * 256;
      const int r = c / CPR, rr = row0 + r;
      if (c % CPR == 0 && c < ATT_TILE * CPR) {
        delta_l[r] = part;
        if (rr < R) a.delta[((int64_t)bh * sh.NQ + rr / sh.K) * sh.K + rr % sh.K] = part;
      }
    }
    gq.template store<true, false>(q_l, nullptr, tid);
    gdo.template store<true, false>(do_l, nullptr, tid);
    __syncthreads();
    if constexpr (SL::PARK) {
      qf.load(q_l + (lane & 31) * Lds<T, DH>::RSTRIDE, true, h);
      dof.load(do_l + (lane & 31) * Lds<T, DH>::RSTRIDE, true, h);
      delta = delta_l[lane & 31];
      __syncthreads();                                   // the staging regions are the waves' own again
    } else {
      qf.load(q_l + (lane & 31) * Lds<T, DH>::RSTRIDE);
      dof.load(do_l + (lane & 31) * Lds<T, DH>::RSTRIDE);
      delta = delta_l[lane & 31];
    }
  } else {
    if (wave * ATT_TILE < sh.S_text) {
      kreg.load(kt, sh.t_ss, wave * ATT_TILE, sh.S_text, lane);
      vreg.load(vt, sh.t_ss, wave * ATT_TILE, sh.S_text, lane);
    }
    qf.load(q_row, valid, h);
    dof.load(do_row, valid, h);
    delta = pair_dot<T, DH>(do_row, o_row, h, valid);
    if (wave == 0 && valid && h == 0) a.delta[stat] = delta;
  }
  ctx = min(ctx, sh.S_text);
  int tile_ctx = ctx;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tile_ctx = max(tile_ctx, __shfl_xor(tile_ctx, o));
  const int ntiles = (tile_ctx + ATT_TILE - 1) / ATT_TILE;
  BG_TRACE(0, 2);

  f32x16 dq[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  for (int t = wave; t < ntiles; t += ATT_WAVES) {
    const int kb = t * ATT_TILE;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // this wave's reads of the previous tile are done
    kreg.template store<true, false>(k_row, nullptr, lane);
    vreg.template store<true, false>(v_row, nullptr, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // LDS is in order within a wave
    if (t + ATT_WAVES < ntiles) {
      kreg.load(kt, sh.t_ss, kb + ATT_WAVES * ATT_TILE, sh.S_text, lane);
      vreg.load(vt, sh.t_ss, kb + ATT_WAVES * ATT_TILE, sh.S_text, lane);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    if (t == wave) BG_TRACE(0, 8);
    mma_reduce_d<T, DH>(s, k_row, lane, [&](int tt) { return qf.get(tt, h); });
    mma_reduce_d<T, DH>(dp, v_row, lane, [&](int tt) { return dof.get(tt, h); });
    if (t == wave) BG_TRACE(0, 9);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb + tile_row(r, h);
      const float p = (key < ctx) ? __expf(s[r] * sh.scale - lse) : 0.f;
      float dpr = dp[r];
      if constexpr (DROP) dpr = drop_keep(drop, row_key, (uint32_t)key) ? dpr * drop.inv_keep : 0.f;
      s[r] = p * (dpr - delta) * sh.scale;  // dS^T
    }
    if (t == wave) BG_TRACE(0, 10);
    if constexpr (sizeof(T) == 4) mma_reduce_rows<T, DH, 0, NDT>(dq, s, k_row, lane);
    else mma_reduce_rows_tr<DH, 0, NDT>(dq, s, k_row, lane);
    if (t == wave) BG_TRACE(0, 11);
  }

  BG_TRACE(0, 3);
  // branch keys: path entries wave, wave + 4, ...
  for (int p = wave; p < sh.max_path; p += ATT_WAVES) {
    const int slot = p == wave ? slot_first : a.path[n * sh.max_path + p];   // in bounds for every p < max_path: no wait on path_len first
    const int64_t off = (int64_t)b * sh.s_sb + (int64_t)hh * sh.s_sh + (int64_t)kq * sh.s_sk;
    const T* kr;
    const T* vr;
    slot_rows<T>(a, slot, p < plen, off, kr, vr);
    const bool act = p < plen && kr;
    // the key row (both layouts) and the value row in ONE round of loads where the registers allow: path entry -> slot pointer
    // -> key -> value -> key again was five dependent round trips an entry
    constexpr bool TOGETHER = DH * sizeof(T) <= 256;
    float sc, dpv;
    [[maybe_unused]] BwdPathRow<T, TOGETHER ? DH : 32> kv;
    if constexpr (TOGETHER) {
      kv.load(kr, vr, h, act);
      sc = kv.dot_k(q_path, h, act) * sh.scale;
      dpv = kv.dot_v(do_path, h, act);
    } else {
      sc = pair_dot<T, DH>(q_row, kr, h, act) * sh.scale;
      dpv = pair_dot<T, DH>(do_row, vr, h, act);
    }
    if constexpr (DROP) dpv = drop_keep(drop, row_key, (uint32_t)(sh.S_text + p)) ? dpv * drop.inv_keep : 0.f;
    if (act) {
      const float pw = __expf(sc - lse);
      if constexpr (TOGETHER) kv.axpy_k(dq, pw * (dpv - delta) * sh.scale);
      else axpy_rowvec<T, DH>(dq, pw * (dpv - delta) * sh.scale, kr, h, true);
    }
  }

  // the four partial dQ tiles meet in LDS: mo[wave][d][query]
  BG_TRACE(0, 4);
  __syncthreads();
  BG_TRACE(0, 5);
  float* mo = (float*)smem;
  const int qcol = lane & 31;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) mo[(wave * DH + dt * 32 + tile_row(r, h)) * ATT_TILE + qcol] = dq[dt][r];
  __syncthreads();
  {
    constexpr int PER = DH / 8;                          // thread -> query tid & 31, feature group tid >> 5
    const int qi = tid & 31, grp = tid >> 5;
    const int rq2 = tile_x * ATT_TILE + qi;
    if (rq2 < R) {
      const int n2 = rq2 / sh.K, k2 = rq2 % sh.K;
      float acc[PER];
      merge4<DH, PER>(mo, qi, grp, acc);
      T* dq_row = (T*)a.dq + (int64_t)b * sh.dq_sb + (int64_t)hh * sh.dq_sh + (int64_t)n2 * sh.dq_sn + (int64_t)k2 * sh.dq_sk + grp * PER;
      store_vec<T, PER>(dq_row, acc);
    }
  }
  BG_TRACE(0, 6);
  BG_TRACE_VAL(0, 7, (unsigned long long)(ntiles > wave ? (ntiles - wave + ATT_WAVES - 1) / ATT_WAVES : 0));
}

// ---------------------------------------------------------------------------------------------------
// dK / dV of the text rows: one 32-key tile per workgroup, the query tiles that reach it dealt round-robin to the four
// waves.  [DT0, DT1) is the slice of d (in 32s) this launch accumulates: all of it up to DH = 128, two launches at 256.
// OWN_DELTA: delta = rowsum(dO o O) of a gathered query tile is worked out here, from its O rows, instead of being read from
// a.delta (the one-launch backward, where the dQ workgroups that write a.delta run beside this one, not before it).
template <typename T, int DH, int DT0, int DT1, bool OWN_DELTA, bool DROP = false>
__device__ __forceinline__ void bwd_dkv_tile(const AttnArgs& a, char* smem, int tile_x, int bh) {
  constexpr int DPASS = (DT1 - DT0) * 32;
  // (the full-width pass at Dh 256 holds 256 accumulator registers + 128 of K / V fragments: no room for a tile in flight)
  constexpr bool PREFETCH = DH >= 128 && DPASS < 256;
  using SL = BwdSplitLds<T, DH, DPASS>;
  using L = Lds<T, DH>;
  const bg_attn_shape& sh = a.sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  [[maybe_unused]] const int n_xtiles = (sh.S_text + ATT_TILE - 1) / ATT_TILE;
  BG_TRACE(1, 0);
  T* q_row_l = (T*)(smem + wave * SL::WAVE_BYTES);
  T* do_row_l = (T*)(smem + wave * SL::WAVE_BYTES + SL::TILE_BYTES);
  float* lse_s = (float*)(smem + wave * SL::WAVE_BYTES + 2 * SL::TILE_BYTES);
  float* delta_s = lse_s + ATT_TILE;
  int* ctx_s = (int*)(delta_s + ATT_TILE);
  [[maybe_unused]] uint32_t* rkey_s = (uint32_t*)(ctx_s + ATT_TILE);
  [[maybe_unused]] DropKeys drop;
  [[maybe_unused]] uint32_t g_rkey = 0;
  if constexpr (DROP) drop = drop_keys(a, bh);
  int* reach_s = (int*)(smem + SL::BYTES);               // one int per query tile, behind the staging / merge area

  const int b = bh / sh.H, hh = bh % sh.H;
  const int R = sh.NQ * sh.K;
  const int n_qtiles = (R + ATT_TILE - 1) / ATT_TILE;
  const int key0 = tile_x * ATT_TILE;
  const int key = key0 + (lane & 31);
  [[maybe_unused]] const bool kvalid = key < sh.S_text;
  [[maybe_unused]] const int64_t row_off = (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh + (int64_t)key * sh.t_ss;
  // a wave gathers its query tiles itself: Q and dO rows in whole 16-byte chunks, and the rows' lse / delta / ctx
  constexpr int VEC = 16 / sizeof(T);
  constexpr int CPR = DH / VEC;                          // 16-byte chunks per row
  constexpr int PERW = ATT_TILE * CPR / 64;              // chunks per lane
  // the full-width pass at Dh 256 gathers a tile in two halves of 16 rows (64 registers in flight, not 128: with 256 of
  // accumulators and 128 of K / V fragments the whole tile would spill)
  constexpr int GPARTS = DPASS >= 256 ? 4 : 1;
  constexpr int PERG = PERW / GPARTS;
  typedef typename TileRegs<T, DH, 64>::V Chunk;
  Chunk gq[PERG], gd[PERG];
  [[maybe_unused]] Chunk go[OWN_DELTA ? PERG : 1];
  float g_lse = 0.f, g_delta = 0.f;
  int g_ctx = 0;
  const T* q_head = (const T*)a.q + (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh;
  const T* do_head = (const T*)a.dout + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh;
  [[maybe_unused]] const T* o_head = (const T*)a.out + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh;
  auto gather_part = [&](int tq, auto part) __attribute__((always_inline)) {
    constexpr int PART = decltype(part)::value;
    const int q0 = tq * ATT_TILE;
#pragma unroll
    for (int i = 0; i < PERG; ++i) {
      const int c = lane + (PART * PERG + i) * 64;
      const int r = c / CPR, d0 = (c % CPR) * VEC;
      // rows past the end read the last row instead (no branch around the loads): their ctx is 0 below, which masks them
      const int rr = min(q0 + r, R - 1);
      const int nn = rr / sh.K, kk = rr % sh.K;
      // 32-bit offsets inside one (batch, head) slice - checked on the host (slice_fits_32) - on a wave-uniform base
      const unsigned qoff = (unsigned)nn * (unsigned)sh.q_sn + (unsigned)kk * (unsigned)sh.q_sk + (unsigned)d0;
      const unsigned ooff = (unsigned)nn * (unsigned)sh.o_sn + (unsigned)kk * (unsigned)sh.o_sk + (unsigned)d0;
      gq[i] = *(const Chunk*)(q_head + qoff);
      gd[i] = *(const Chunk*)(do_head + ooff);
      if constexpr (OWN_DELTA) go[i] = *(const Chunk*)(o_head + ooff);
    }
    if constexpr (PART == 0) {
      const int rq = q0 + (lane & 31);
      const bool qv = rq < R;
      const int nn = qv ? rq / sh.K : 0, kk = qv ? rq % sh.K : 0;
      const int64_t stat = ((int64_t)bh * sh.NQ + nn) * sh.K + kk;
      g_ctx = qv ? min(a.ctx_len[kk], sh.S_text) : 0;    // 0 for rows past the end: masks everything
      g_lse = qv ? a.lse[stat] : 0.f;
      if constexpr (!OWN_DELTA) g_delta = qv ? a.delta[stat] : 0.f;
      if constexpr (DROP) g_rkey = drop_row_key(drop, (uint32_t)rq);
    }
  };

  // the gathered tile goes to this wave's staging area (row layout), with its rows' lse / ctx and delta
  auto stage_part = [&](auto part) __attribute__((always_inline)) {
    constexpr int PART = decltype(part)::value;
#pragma unroll
    for (int i = 0; i < PERG; ++i) {
      const int c = lane + (PART * PERG + i) * 64;
      const int r = c / CPR, d0 = (c % CPR) * VEC;
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          q_row_l[r * L::RSTRIDE + d0 + j] = gq[i][j];
          do_row_l[r * L::RSTRIDE + d0 + j] = gd[i][j];
        }
      } else {
        *(bf16x8*)(q_row_l + r * L::RSTRIDE + d0) = gq[i];
        *(bf16x8*)(do_row_l + r * L::RSTRIDE + d0) = gd[i];
      }
      if constexpr (OWN_DELTA) {                         // the CPR lanes that hold one row add up its dO . O
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) part = fmaf((float)gd[i][j], (float)go[i][j], part);
#pragma unroll
        for (int o = CPR / 2; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (c % CPR == 0) delta_s[r] = part;
      }
    }
    if (PART == 0 && lane < ATT_TILE) {
      lse_s[lane] = g_lse;
      if constexpr (!OWN_DELTA) delta_s[lane] = g_delta;
      ctx_s[lane] = g_ctx;
      if constexpr (DROP) rkey_s[lane] = g_rkey;
    }
  };
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, 1> P1;
  auto gather = [&](int tq) __attribute__((always_inline)) { gather_part(tq, P0()); };     // (one-part form: the whole tile)
  auto stage = [&]() __attribute__((always_inline)) { stage_part(P0()); };
  // gather + stage of a whole tile, one part after the other
  auto fetch = [&](int tq) __attribute__((always_inline)) {
    gather_part(tq, P0());
    stage_part(P0());
    if constexpr (GPARTS >= 2) { gather_part(tq, P1()); stage_part(P1()); }
    if constexpr (GPARTS >= 4) {
      gather_part(tq, std::integral_constant<int, 2>()); stage_part(std::integral_constant<int, 2>());
      gather_part(tq, std::integral_constant<int, 3>()); stage_part(std::integral_constant<int, 3>());
    }
  };

  // One round of loads: the workgroup's K / V tile in whole row chunks (parked in LDS, see BwdSplitLds), the context
  // lengths behind the reach of every query tile (largest ctx_len among its 32 queries; 256 queries a step) and - before
  // the reach is known - a first query tile: tiles are dealt from the LAST one down (under a causal mask the late queries
  // are the ones that see a key tile), so wave w bets on tile n_qtiles - 1 - w and stages it at once (a tile held in
  // registers across the barrier and the dealing was what pushed the kernel past three waves per SIMD).
  const int guess = n_qtiles - 1 - wave;
  typename std::conditional<SL::OWN, LdsRowFrag<T, DH>, RowFrag<T, DH>>::type kf, vf;
  if constexpr (SL::OWN) {
    T* k_l = (T*)(smem + SL::STAGE_BYTES);
    T* v_l = (T*)(smem + SL::STAGE_BYTES + SL::TILE_BYTES);
    TileRegs<T, DH, 256> gk, gv;
    const int64_t base = (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
    gk.load((const T*)a.k_text + base, sh.t_ss, key0, sh.S_text, tid);
    gv.load((const T*)a.v_text + base, sh.t_ss, key0, sh.S_text, tid);
    for (int r0 = 0; r0 < n_qtiles * ATT_TILE; r0 += 256) {
      const int rq = r0 + tid;
      int t = rq < R ? min(a.ctx_len[rq % sh.K], sh.S_text) : 0;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) t = max(t, __shfl_xor(t, o));
      if ((tid & 31) == 0 && rq / ATT_TILE < n_qtiles) reach_s[rq / ATT_TILE] = t;
    }
    static_assert(GPARTS == 1, "the one-launch form keeps K / V in LDS: Dh <= 128");
    if (guess >= 0) gather(guess);
    gk.template store<true, false>(k_l, nullptr, tid);
    gv.template store<true, false>(v_l, nullptr, tid);
    if (guess >= 0) stage();
    BG_TRACE(1, 1);
    __syncthreads();
    kf.load(k_l + (lane & 31) * L::RSTRIDE);
    vf.load(v_l + (lane & 31) * L::RSTRIDE);
  } else if constexpr (SL::PARK) {
    // the K / V tile passes through wave 0's staging region on its way into every wave's registers (see BwdSplitLds::PARK);
    // the bet's first gather flies meanwhile, and is staged once the region is free again
    T* k_l = (T*)smem;
    T* v_l = (T*)(smem + SL::TILE_BYTES);
    TileRegs<T, DH, 256> gk, gv;
    const int64_t base = (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
    gk.load((const T*)a.k_text + base, sh.t_ss, key0, sh.S_text, tid);
    gv.load((const T*)a.v_text + base, sh.t_ss, key0, sh.S_text, tid);
    for (int r0 = 0; r0 < n_qtiles * ATT_TILE; r0 += 256) {
      const int rq = r0 + tid;
      int t = rq < R ? min(a.ctx_len[rq % sh.K], sh.S_text) : 0;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) t = max(t, __shfl_xor(t, o));
      if ((tid & 31) == 0 && rq / ATT_TILE < n_qtiles) reach_s[rq / ATT_TILE] = t;
    }
    if (guess >= 0) gather_part(guess, P0());
    gk.template store<true, false>(k_l, nullptr, tid);
    gv.template store<true, false>(v_l, nullptr, tid);
    __syncthreads();
    kf.load(k_l + (lane & 31) * L::RSTRIDE, true, h);
    vf.load(v_l + (lane & 31) * L::RSTRIDE, true, h);
    BG_TRACE(1, 1);
    __syncthreads();
    if (guess >= 0) {
      stage_part(P0());
      if constexpr (GPARTS >= 2) { gather_part(guess, P1()); stage_part(P1()); }
      if constexpr (GPARTS >= 4) {
        gather_part(guess, std::integral_constant<int, 2>()); stage_part(std::integral_constant<int, 2>());
        gather_part(guess, std::integral_constant<int, 3>()); stage_part(std::integral_constant<int, 3>());
      }
    }
  } else {
    kf.load((const T*)a.k_text + row_off, kvalid, h);
    vf.load((const T*)a.v_text + row_off, kvalid, h);
    for (int tq = wave; tq < n_qtiles; tq += ATT_WAVES) {
      const int rq = tq * ATT_TILE + (lane & 31);
      int t = rq < R ? min(a.ctx_len[rq % sh.K], sh.S_text) : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t = max(t, __shfl_xor(t, o));
      if (lane == 0) reach_s[tq] = t;
    }
    if (guess >= 0) fetch(guess);
    BG_TRACE(1, 1);
    __syncthreads();
  }
  BG_TRACE(1, 2);
  // This wave's tiles: every fourth of the tiles that reach key0, counted from the last tile down.  The eligible tiles
  // of 64 at a time are one ballot; walking it is scalar work.
  int chunk = (n_qtiles - 1) / 64, dealt = 0;
  auto eligible = [&](int ch) __attribute__((always_inline)) -> unsigned long long {
    const int idx = ch * 64 + lane;
    return __ballot(idx < n_qtiles && reach_s[idx < n_qtiles ? idx : 0] > key0);
  };
  unsigned long long elig = n_qtiles > 0 ? eligible(chunk) : 0ull;
  auto next_mine = [&]() __attribute__((always_inline)) -> int {
    for (;;) {
      while (elig) {
        const int t = 63 - __builtin_clzll(elig);
        elig &= ~(1ull << t);
        const bool mine = (dealt % ATT_WAVES) == wave;
        ++dealt;
        if (mine) return chunk * 64 + t;
      }
      if (chunk <= 0) return -1;
      --chunk;
      elig = eligible(chunk);
    }
  };

  f32x16 dk[DH / 32], dv[DH / 32];
#pragma unroll
  for (int dt = DT0; dt < DT1; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  int tq = next_mine();
  bool staged = tq == guess;                             // the bet was won: the tile is in the staging area already
  [[maybe_unused]] bool have = false;                    // (prefetch form) the tile is in the gather registers
  BG_TRACE(1, 3);
  [[maybe_unused]] unsigned long long my_tiles = 0;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  while (tq >= 0) {
    ++my_tiles;
    if (!staged) {
      if constexpr (GPARTS == 1) {
        if (!PREFETCH || !have) gather(tq);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // this wave's reads of the previous tile are done
        stage();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        fetch(tq);
      }
    }
    staged = false;
    have = false;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    tq = next_mine();
    // Dh >= 128 runs one wave per SIMD whatever it does: the next tile's loads fly behind this tile's MFMAs.  Below that
    // the 32 registers of a tile in flight are the difference between two and three waves per SIMD, and three waves hide
    // the load better than the prefetch does (most waves have one tile at these sizes): the gather follows the MFMAs.
    if constexpr (PREFETCH) {
      if (tq >= 0) { gather(tq); have = true; }
    }

    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    if (my_tiles == 1) BG_TRACE(1, 8);
    mma_reduce_d<T, DH>(s, q_row_l, lane, [&](int t) { return kf.get(t, h); });      // S[query][key]
    mma_reduce_d<T, DH>(dp, do_row_l, lane, [&](int t) { return vf.get(t, h); });    // dP[query][key]
    if (my_tiles == 1) BG_TRACE(1, 9);
    // the 16 query rows of this lane's registers are four runs of four: rows 8 g + 4 h .. + 3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 lse4 = *(const f32x4*)(lse_s + 8 * g + 4 * h);
      const f32x4 del4 = *(const f32x4*)(delta_s + 8 * g + 4 * h);
      const i32x4 ctx4 = *(const i32x4*)(ctx_s + 8 * g + 4 * h);
      [[maybe_unused]] i32x4 rk4;
      if constexpr (DROP) rk4 = *(const i32x4*)(rkey_s + 8 * g + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j;
        const float p = (key < ctx4[j]) ? __expf(s[r] * sh.scale - lse4[j]) : 0.f;
        float mk = 1.f;
        if constexpr (DROP) mk = drop_keep(drop, (uint32_t)rk4[j], (uint32_t)key) ? drop.inv_keep : 0.f;
        dp[r] = p * (dp[r] * mk - del4[j]) * sh.scale;  // dS
        s[r] = p * mk;                                  // P o M
      }
    }
    if (my_tiles == 1) BG_TRACE(1, 10);
    if constexpr (sizeof(T) == 4) {
      mma_reduce_rows<T, DH, DT0, DT1>(dv, s, do_row_l, lane);     // dV^T += dO^T·P
      mma_reduce_rows<T, DH, DT0, DT1>(dk, dp, q_row_l, lane);     // dK^T += Q^T·dS
    } else {
      mma_reduce_rows_tr<DH, DT0, DT1>(dv, s, do_row_l, lane);
      mma_reduce_rows_tr<DH, DT0, DT1>(dk, dp, q_row_l, lane);
    }
    if (my_tiles == 1) BG_TRACE(1, 11);
  }

  // merge and store, dK then dV through the same LDS area: mo[wave][d - 32 DT0][key]
  BG_TRACE(1, 4);
  float* mo = (float*)smem;
  const int kcol = lane & 31;
  constexpr int PER = DPASS / 8;
  const int ki = tid & 31, grp = tid >> 5;
  const int key2 = key0 + ki;
  const int64_t off2 = (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh + (int64_t)key2 * sh.t_ss + DT0 * 32 + grp * PER;
  // what comes in for this thread's piece of the key row (the tree pass's gradient of the text keys / values) is requested
  // before the merge: read inside the store, each piece was a dependent round trip at the very end of the workgroup's life.
  // Dh 256 holds 512 registers by now: its dV piece is requested once dK's has left.
  VecIn<T, PER> in_k, in_v;
  in_k.load(a.dk_text_in ? (const T*)a.dk_text_in + off2 : nullptr, key2 < sh.S_text);
  if constexpr (DPASS <= 128) in_v.load(a.dv_text_in ? (const T*)a.dv_text_in + off2 : nullptr, key2 < sh.S_text);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    if constexpr (DPASS > 128) {
      if (which == 1) in_v.load(a.dv_text_in ? (const T*)a.dv_text_in + off2 : nullptr, key2 < sh.S_text);
    }
    __syncthreads();                                     // staging (or the previous merge) no longer read
#pragma unroll
    for (int dt = DT0; dt < DT1; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        mo[(wave * DPASS + (dt - DT0) * 32 + tile_row(r, h)) * ATT_TILE + kcol] = which == 0 ? dk[dt][r] : dv[dt][r];
    __syncthreads();
    if (key2 < sh.S_text) {
      float acc[PER];
      merge4<DPASS, PER>(mo, ki, grp, acc);
      T* out = (T*)(which == 0 ? a.dk_text : a.dv_text) + off2;
      store_vec<T, PER>(out, acc, which == 0 ? in_k : in_v);
    }
    if (which == 0) BG_TRACE(1, 5);
  }
  BG_TRACE(1, 6);
  BG_TRACE_VAL(1, 7, my_tiles);
}

// ---------------------------------------------------------------------------------------------------
// The kernels.  Two-launch form (dQ + delta, then dK / dV reading a.delta; the slot rows follow in attn_bwd.hip's kernel)
// and the one-launch form, where every workgroup of the backward - key tiles, query tiles, slot rows - is in ONE grid and
// whoever needs delta works it out from the O rows it gathers anyway.  At the BASELINE shapes a workgroup's life is a few
// dependent memory round trips (3-4 us of loads around 1-3 us of MFMA work, benchmarks/probes/attn_trace.py): three launches
// pay them one after the other, each with its own ramp and drain; one launch overlaps them.
template <typename T, int DH>
__global__ __launch_bounds__(256, (DH <= 64 ? 3 : 1)) void branch_attn_bwd_dq_split_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const BlockXY blk = xcd_block();
  bwd_dq_tile<T, DH, false>(a, smem, blk.x, blk.y);
}

template <typename T, int DH, int DT0, int DT1>
__global__ __launch_bounds__(256, (DH * sizeof(T) <= 128 ? 3 : 1)) void branch_attn_bwd_dkv_split_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const BlockXY blk = xcd_block();
  bwd_dkv_tile<T, DH, DT0, DT1, false>(a, smem, blk.x, blk.y);
}

// grid.x = B*H * (key tiles + query tiles + slots).  The workgroups of one head share its Q, K, V, dO: they are given
// block numbers that are congruent mod 8, i.e. one XCD and one L2 (blocks are dealt round-robin over the 8 XCDs), and
// within a head the order is heaviest first - key tiles from 0 (under a causal mask tile 0 is reached by every query
// tile), query tiles from the last one down, then the slot rows.
template <typename T, int DH, bool DROP = false>
__global__ __launch_bounds__(256, (DH * sizeof(T) <= 128 ? 3 : 1)) void branch_attn_bwd_one_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bg_attn_shape& sh = a.sh;
  const int nq = (sh.NQ * sh.K + ATT_TILE - 1) / ATT_TILE, nk = (sh.S_text + ATT_TILE - 1) / ATT_TILE;
  const int per_head = nk + nq + sh.n_slots, heads = sh.B * sh.H;
  const int id = blockIdx.x;
  int bh, item;
  if (heads % 8 == 0) {
    const int j = id / 8;
    bh = (j / per_head) * 8 + id % 8;
    item = j % per_head;
  } else {
    bh = id / per_head;
    item = id % per_head;
  }
  if (item < nk) bwd_dkv_tile<T, DH, 0, DH / 32, true, DROP>(a, smem, item, bh);
  else if (item < nk + nq) bwd_dq_tile<T, DH, DROP>(a, smem, nq - 1 - (item - nk), bh);
  else bwd_slot_rows<T, DH, true, DROP>(a, (int*)smem, (int*)smem + 256, item - nk - nq, bh);
}

template <typename T, int DH, int DT0, int DT1>
static void launch_dkv_split(const AttnArgs& a, dim3 grid, size_t reach_bytes, hipStream_t st) {
  constexpr size_t lds = BwdSplitLds<T, DH, (DT1 - DT0) * 32>::BYTES;
  auto kernel = branch_attn_bwd_dkv_split_kernel<T, DH, DT0, DT1>;
  set_lds_limit(kernel, lds + reach_bytes);
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds + reach_bytes, st, a);
}

template <typename T, int DH>
static int launch_split(const AttnArgs& a, bool one, hipStream_t st) {
  const bg_attn_shape& sh = a.sh;
  const int64_t R = (int64_t)sh.NQ * sh.K;
  const size_t reach = (size_t)cdiv(R, ATT_TILE) * 4 + 16;
  constexpr size_t lq = BwdSplitLds<T, DH, DH>::BYTES;
  // Dh 256 (bf16) CAN take the one-launch form since the full-width dK / dV pass (BRANCHGAN_ATTN_BWD_ONE256=1), but at one
  // workgroup per CU (137 KB of LDS) the grid runs round after round whichever launch a workgroup belongs to: 155.2 us per
  // call against 68.2 + 65.6 + 18.2 at configs[3] (profiles/r3_s_attn_c4_one256.txt) - off by default.
  static const bool one256 = [] { const char* e = getenv("BRANCHGAN_ATTN_BWD_ONE256"); return e && e[0] == '1'; }();
  if (DH <= 128 || one256) {
    if (one) {
      constexpr size_t lk = BwdSplitLds<T, DH, DH>::BYTES;
      size_t lds = lq > lk + reach ? lq : lk + reach;
      if (lds < (256 + ATT_WAVES) * sizeof(int)) lds = (256 + ATT_WAVES) * sizeof(int);
      const int64_t per_head = cdiv((int64_t)sh.S_text, ATT_TILE) + cdiv(R, ATT_TILE) + sh.n_slots;
      const dim3 grid((unsigned)(per_head * sh.B * sh.H));
      if (sh.p_drop > 0.f) {
        auto kernel = branch_attn_bwd_one_kernel<T, DH, true>;
        set_lds_limit(kernel, lds);
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
        return 2;
      }
      auto kernel = branch_attn_bwd_one_kernel<T, DH>;
      set_lds_limit(kernel, lds);
      hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
      return 2;
    }
  }
  const dim3 grid_q((unsigned)cdiv(R, ATT_TILE), (unsigned)(sh.B * sh.H));
  const dim3 grid_k((unsigned)cdiv((int64_t)sh.S_text, ATT_TILE), (unsigned)(sh.B * sh.H));
  auto dqk = branch_attn_bwd_dq_split_kernel<T, DH>;
  set_lds_limit(dqk, lq);
  hipLaunchKernelGGL(dqk, grid_q, dim3(256), lq, st, a);
  if (sh.S_text > 0) {
    // Dh 256: ONE pass over the whole head dimension (round 3) instead of two half-passes that each recomputed S and dP and
    // gathered Q / dO again; BRANCHGAN_ATTN_DKV256_HALVES=1 brings the two launches back (A/B runs)
    static const bool halves = [] { const char* e = getenv("BRANCHGAN_ATTN_DKV256_HALVES"); return e && e[0] == '1'; }();
    if constexpr (DH <= 128) {
      launch_dkv_split<T, DH, 0, DH / 32>(a, grid_k, reach, st);
    } else if (!halves) {
      launch_dkv_split<T, DH, 0, DH / 32>(a, grid_k, reach, st);
    } else {
      launch_dkv_split<T, DH, 0, DH / 64>(a, grid_k, reach, st);
      launch_dkv_split<T, DH, DH / 64, DH / 32>(a, grid_k, reach, st);
    }
  }
  return 1;
}

// Launches the backward in the split form when the shape is covered.  Returns 0 = not covered (fp32 at Dh = 256: four private
// fp32 tile pairs do not fit the LDS) or switched off (BRANCHGAN_ATTN_BWD_SPLIT=0: A/B runs); 1 = dQ, delta and the text rows'
// dK / dV are launched, the slot rows are the caller's; 2 = everything is launched (the one-launch form: Dh <= 128, unless
// BRANCHGAN_ATTN_BWD_ONE=0).
// The gathers address a (batch, head) slice of q / out / dout with 32-bit element offsets.
static bool slice_fits_32(const bg_attn_shape& sh) {
  const int64_t lim = (int64_t)1 << 31;
  const int64_t q = (int64_t)(sh.NQ - 1) * sh.q_sn + (int64_t)(sh.K - 1) * sh.q_sk + sh.Dh;
  const int64_t o = (int64_t)(sh.NQ - 1) * sh.o_sn + (int64_t)(sh.K - 1) * sh.o_sk + sh.Dh;
  return sh.q_sn >= 0 && sh.q_sk >= 0 && sh.o_sn >= 0 && sh.o_sk >= 0 && q < lim && o < lim;
}

template <typename T>
int launch_bwd_split(const AttnArgs& a, hipStream_t st) {
  static const bool enabled = [] { const char* e = getenv("BRANCHGAN_ATTN_BWD_SPLIT"); return !(e && e[0] == '0'); }();
  static const bool one = [] { const char* e = getenv("BRANCHGAN_ATTN_BWD_ONE"); return !(e && e[0] == '0'); }();
  const bg_attn_shape& sh = a.sh;
  const bool drop = sh.p_drop > 0.f;                     // dropout lives in the one-launch form only (attn_bwd_covers_dropout)
  if (!enabled && !drop) return 0;
  if (sizeof(T) == 4 && sh.Dh == 256) return 0;
  if (!slice_fits_32(sh)) return 0;
  // reach_s lives behind the staging area: one int per query tile
  if ((size_t)cdiv((int64_t)sh.NQ * sh.K, ATT_TILE) * 4 > 16 * 1024) return 0;
  switch (sh.Dh) {
    case 32: return launch_split<T, 32>(a, one || drop, st);
    case 64: return launch_split<T, 64>(a, one || drop, st);
    case 128: return launch_split<T, 128>(a, one || drop, st);
    default:
      if constexpr (sizeof(T) == 2) return launch_split<T, 256>(a, one, st);
      return 0;
  }
}

// The shapes whose backward can apply dropout: the ones the one-launch form takes.
bool attn_bwd_covers_dropout(const bg_attn_shape& sh) {
  return sh.Dh <= 128 && slice_fits_32(sh) && (size_t)cdiv((int64_t)sh.NQ * sh.K, ATT_TILE) * 4 <= 16 * 1024;
}

#ifdef BG_ATTN_TRACE
extern "C" int bg_debug_attn_trace(unsigned long long* host, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_trace), bytes < sizeof(g_attn_trace) ? bytes : sizeof(g_attn_trace));
}
#endif

template int launch_bwd_split<float>(const AttnArgs&, hipStream_t);
template int launch_bwd_split<bf16_t>(const AttnArgs&, hipStream_t);

}  // namespace bg
