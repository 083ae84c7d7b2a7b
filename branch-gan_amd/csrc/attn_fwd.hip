// Branch ("non-residual") attention forward — one MFMA kernel, arithmetic mask, online softmax.
//
// Replaces Models/NeoXBranchAttention/NeoXNonResidualAttention.py:183-232, Models/GPT2BranchAttention/
// NonResidualAttention.py:119-172, the mask tensors of BranchPredictions/MultiStepPrediction.py:92-127 and
// the KV concatenation of :129-152.  See include/branchgan.h: bg_branch_attn_fwd.
//
// Work split: grid = (ceil(NQ*K / 128), B*H); a workgroup = 4 waves, each wave owns 32 query rows and
// keeps them "on the lane": S^T = K·Q^T puts one query per lane column, so the running max/sum and the
// output rescale are per-lane scalars, and P^T feeds O^T = V^T·P^T straight from the accumulators.
// Text K/V tiles (32 keys) are staged once per workgroup in LDS and shared by the 4 waves; tiles beyond
// the workgroup's largest ctx_len are skipped (sparseIndex is sorted, so a tile of consecutive k has a
// tight bound).  Branch keys (ancestors + self: the same branch point k of each slot on the node's path)
// are this lane's own rows, so they are handled as per-lane dot products folded into the same online
// softmax — no 32x wasted diagonal tiles.
#include <stdlib.h>

#include "attn_common.h"
#include "attn_dropout.h"

namespace bg {

template <typename T, int DH>
__global__ __launch_bounds__(256) void branch_attn_fwd_kernel(AttnArgs a) {
  using L = Lds<T, DH>;
  constexpr int NDT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* k_row = (T*)smem;                              // K tile, "row" layout
  T* v_z = k_row + L::ROW_ELEMS;                    // V tile, row layout: read by columns (bf16: transposed LDS reads)
  int* red = (int*)(v_z + L::ROW_ELEMS);

  const bg_attn_shape& sh = a.sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const BlockXY blk = xcd_block();
  const int bh = blk.y, b = bh / sh.H, hh = bh % sh.H;
  const int R = sh.NQ * sh.K;
  const int rq = blk.x * (ATT_WAVES * ATT_TILE) + wave * ATT_TILE + (lane & 31);
  const bool valid = rq < R;
  const int n = valid ? rq / sh.K : 0, kq = valid ? rq % sh.K : 0;
  int ctx = valid ? a.ctx_len[kq] : 0;
  ctx = min(ctx, sh.S_text);

  const T* q_row = (const T*)a.q + (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh + (int64_t)n * sh.q_sn + (int64_t)kq * sh.q_sk;
  const T* kt = (const T*)a.k_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  const T* vt = (const T*)a.v_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;

  RowFrag<T, DH> qf;
  qf.load(q_row, valid, h);

  // workgroup / wave bounds on the text range
  int wave_ctx = ctx;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_ctx = max(wave_ctx, __shfl_xor(wave_ctx, o));
  if (lane == 0) red[wave] = wave_ctx;
  __syncthreads();
  const int block_ctx = max(max(red[0], red[1]), max(red[2], red[3]));

  float m = -INFINITY, l = 0.f;
  f32x16 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

  // software pipeline: the global loads of tile i+1 are issued before tile i is consumed
  TileRegs<T, DH, 256> kreg, vreg;
  if (block_ctx > 0) {
    kreg.load(kt, sh.t_ss, 0, sh.S_text, tid);
    vreg.load(vt, sh.t_ss, 0, sh.S_text, tid);
  }
  for (int kb = 0; kb < block_ctx; kb += ATT_TILE) {
    __syncthreads();  // previous tile fully consumed
    kreg.template store<true, false>(k_row, nullptr, tid);
    vreg.template store<true, false>(v_z, nullptr, tid);
    __syncthreads();
    if (kb + ATT_TILE < block_ctx) {
      kreg.load(kt, sh.t_ss, kb + ATT_TILE, sh.S_text, tid);
      vreg.load(vt, sh.t_ss, kb + ATT_TILE, sh.S_text, tid);
    }
    if (kb >= wave_ctx) continue;  // wave-uniform

    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    mma_reduce_d<T, DH>(s, k_row, lane, [&](int t) { return qf.get(t, h); });

    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb + tile_row(r, h);
      s[r] = (key < ctx) ? s[r] * sh.scale : -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __expf(m - m_use);  // m = -inf -> 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_use);
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    if constexpr (sizeof(T) == 4) mma_reduce_rows<T, DH, 0, NDT>(o, s, v_z, lane);
    else mma_reduce_rows_tr<DH, 0, NDT>(o, s, v_z, lane);
  }

  // branch keys: the same branch point kq of every slot on this node's path (ancestors, then self)
  const int plen = (valid && sh.max_path) ? a.path_len[n] : 0;   // path tables may be NULL when max_path == 0
  for (int p = 0; p < sh.max_path; ++p) {
    const bool act = p < plen;
    int slot = act ? a.path[n * sh.max_path + p] : 0;
    const int64_t off = (int64_t)b * sh.s_sb + (int64_t)hh * sh.s_sh + (int64_t)kq * sh.s_sk;
    // k_slots[] lives in the kernel argument buffer: index it with a wave-uniform value where possible
    const T* kr = nullptr;
    const T* vr = nullptr;
    for (int sidx = 0; sidx < sh.n_slots; ++sidx) {
      if (sidx == slot) { kr = (const T*)a.k_slots[sidx] + off; vr = (const T*)a.v_slots[sidx] + off; }
    }
    const float sc = pair_dot<T, DH>(q_row, kr, h, act) * sh.scale;
    if (act) {
      const float m_new = fmaxf(m, sc);
      const float alpha = __expf(m - m_new);
      const float pw = __expf(sc - m_new);
      l = l * alpha + pw;
      m = m_new;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      axpy_rowvec<T, DH>(o, pw, vr, h, true);
    }
  }

  T* o_row = (T*)a.out + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh + (int64_t)n * sh.o_sn + (int64_t)kq * sh.o_sk;
  const float inv = (l > 0.f) ? 1.f / l : 0.f;
  store_rowvec<T, DH>(o, inv, o_row, h, valid);
  if (valid && h == 0) a.lse[((int64_t)bh * sh.NQ + n) * sh.K + kq] = (l > 0.f) ? m + logf(l) : -INFINITY;
}


// ---- key-split variant (Dh <= 128; 256 in bf16) -------------------------------------------------------------------------------------
// At Branch-GAN shapes a (batch, head) has only 32-256 query rows, so giving every wave its own 32 queries leaves most of
// the chip idle behind one long dependent chain per wave (8 key tiles, one after the other).  Here a workgroup owns ONE
// 32-query tile and its four waves split the KEY range: wave w takes text tiles w, w+4, ... and branch-path entries
// w, w+4, ..., each with a private K/V staging area in LDS (no workgroup barrier inside the loop, next tile's global
// loads in flight behind the current tile's MFMAs), and the four partial (max, sum, O) states are merged through LDS at
// the end.  grid.x = number of 32-query tiles: 4x the workgroups and a 4x shorter chain per wave.
// BG_ATTN_TRACE (the diagnostic library of `make trace`, benchmarks/probes/attn_fwd_trace.py): every wave of the key-split forward
// leaves wall-clock stamps (100 MHz) at its phase boundaries, with a full s_waitcnt before each.
#ifdef BG_ATTN_TRACE
#define BG_FTRACE_WAVES 8192
__device__ unsigned long long g_attn_fwd_trace[BG_FTRACE_WAVES][16];
#define BG_FTRACE(slot)                                                                            \
  do {                                                                                             \
    __builtin_amdgcn_s_waitcnt(0);                                                                 \
    const unsigned w_ = (blockIdx.x + blockIdx.y * gridDim.x) * ATT_WAVES + (threadIdx.x >> 6);    \
    if ((threadIdx.x & 63) == 0 && w_ < BG_FTRACE_WAVES) g_attn_fwd_trace[w_][slot] = wall_clock64(); \
  } while (0)
#else
#define BG_FTRACE(slot)
#endif

template <typename T, int DH> struct SplitLds {
  using L = Lds<T, DH>;
  static constexpr int K_ELEMS = L::ROW_ELEMS;
  static constexpr int V_ELEMS = L::ROW_ELEMS;
  static constexpr size_t WAVE_BYTES = ((size_t)(K_ELEMS + V_ELEMS) * sizeof(T) + 15) / 16 * 16;
  static constexpr size_t MERGE_BYTES = (size_t)ATT_WAVES * (DH + 2) * ATT_TILE * sizeof(float);
  static constexpr size_t STAGE_BYTES = ((ATT_WAVES * WAVE_BYTES > MERGE_BYTES ? ATT_WAVES * WAVE_BYTES : MERGE_BYTES) + 63) / 64 * 64;
  // the workgroup's own Q tile, fetched once by all 256 threads in whole 16-byte row chunks (as in attn_bwd_split.hip: a lane
  // that reads its own row from global memory touches 32 cache lines per load instruction, and four waves did it four times)
  static constexpr size_t Q_BYTES = ((size_t)L::ROW_ELEMS * sizeof(T) + 15) / 16 * 16;
  // the slot pointer tables (kernel arguments) copied once, by lane, in the prologue's round of loads: a path entry's rows are
  // then one LDS read away from their address (read from the argument buffer they were one more round trip behind the entry)
  static constexpr size_t PTR_OFF = STAGE_BYTES + Q_BYTES + 64;
  static constexpr size_t BYTES = PTR_OFF + 2 * BG_MAX_SLOTS * sizeof(void*);
};

//
// KVSPLIT (round 3): a (batch, head) with few query rows and a LONG key range - the generator's branch steps at c4: 64 queries
// against 512 keys, B*H = 32, i.e. 64 workgroups on a 256-CU chip, each pulling 512 KB of K / V through ONE CU (a CU takes in
// ~25 GB/s from HBM and ~70 GB/s from its L2: 8-20 us whatever the other 192 CUs do).  The text tiles of a query tile are
// dealt to a.kv_splits workgroups (grid.x = query tiles x splits; split s takes tiles [s * per, (s + 1) * per), the branch keys
// ride with the last split); each leaves its merged state - running max, sum and UNNORMALISED output, fp32 - in a.part, and
// branch_attn_fwd_merge_kernel combines the splits of a tile (a second launch: 1.5 us of boundary against the 3.5 us + per
// workgroup that a release / acquire hand-off inside one launch costs on this part, MI355X_MICROARCH.md).
// Layout of a.part per (bh, query tile, split): [DH][32] outputs (feature-major, query on the lane) then [2][32] = max, sum.
template <int DH> constexpr size_t fwd_part_floats() { return (size_t)(DH + 2) * ATT_TILE; }

// DROP: dropout on the attention probabilities (attn_dropout.h) - the sum l runs over the undropped weights, the products with V
// over the kept ones, the output is scaled by 1 / (1 - p) at the end.
// PF: the key / value rows of a wave's first two path entries are requested BEFORE the text tiles and arrive behind them (64
// more registers at Dh 64: two waves per SIMD, not three - taken for launches of at most one workgroup per CU, where the others
// would not exist anyway).  Without PF they are requested together behind the tiles.
template <typename T, int DH, bool KVSPLIT = false, bool DROP = false, bool PF = false>
__global__ __launch_bounds__(256, (PF ? (DH * sizeof(T) <= 128 ? 2 : 1) : (DH * sizeof(T) <= 128 ? 3 : 1))) void branch_attn_fwd_split_kernel(AttnArgs a) {
  using SL = SplitLds<T, DH>;
  constexpr int NDT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bg_attn_shape& sh = a.sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  BG_FTRACE(0);
  T* k_row = (T*)(smem + wave * SL::WAVE_BYTES);
  T* v_z = k_row + SL::K_ELEMS;
  BlockXY blk = xcd_block();
  const int n_splits = KVSPLIT ? a.kv_splits : 1;
  const int split = KVSPLIT ? blk.x % n_splits : 0;
  if constexpr (KVSPLIT) blk.x /= n_splits;
  // text tiles of this workgroup: [t_lo, t_lo + t_per), cut by the tile's own context below
  const int t_per = KVSPLIT ? ((sh.S_text + ATT_TILE - 1) / ATT_TILE + n_splits - 1) / n_splits : (1 << 28);
  const int t_lo = split * t_per;
  const int bh = blk.y, b = bh / sh.H, hh = bh % sh.H;
  const int R = sh.NQ * sh.K;
  const int rq = blk.x * ATT_TILE + (lane & 31);
  const bool valid = rq < R;
  const int n = valid ? rq / sh.K : 0, kq = valid ? rq % sh.K : 0;
  const T* kt = (const T*)a.k_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  const T* vt = (const T*)a.v_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  [[maybe_unused]] DropKeys drop;
  [[maybe_unused]] uint32_t row_key = 0;
  if constexpr (DROP) {
    drop = drop_keys(a, bh);
    row_key = drop_row_key(drop, (uint32_t)rq);
  }
  // ONE round of loads: context length, path, the Q tile and - before the number of key tiles is known - key tile `wave`
  int ctx = valid ? a.ctx_len[kq] : 0;
  const int plen = (valid && sh.max_path) ? a.path_len[n] : 0;   // path tables may be NULL when max_path == 0
  // this wave's first two path entries (p = wave, wave + 4), in the same round
  const int slot_first = wave < sh.max_path ? a.path[n * sh.max_path + wave] : 0;
  const int slot_second = wave + ATT_WAVES < sh.max_path ? a.path[n * sh.max_path + wave + ATT_WAVES] : 0;
  T* q_l = (T*)(smem + SL::STAGE_BYTES);
  TileRegs<T, DH, 256> gq;
  gq.load_nk((const T*)a.q + (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh, sh.q_sn, sh.q_sk, sh.K, R, blk.x * ATT_TILE, tid);
  TileRegs<T, DH, 64> kreg, vreg;
  if ((t_lo + wave) * ATT_TILE < sh.S_text) {
    kreg.load(kt, sh.t_ss, (t_lo + wave) * ATT_TILE, sh.S_text, lane);
    vreg.load(vt, sh.t_ss, (t_lo + wave) * ATT_TILE, sh.S_text, lane);
  }
  BG_FTRACE(1);   // the prologue's loads are in
  const void** ptr_l = (const void**)(smem + SL::PTR_OFF);
  if (tid < sh.n_slots) {
    ptr_l[tid] = a.k_slots[tid];
    ptr_l[BG_MAX_SLOTS + tid] = a.v_slots[tid];
  }
  gq.template store<true, false>(q_l, nullptr, tid);
  __syncthreads();
  const T* q_row = q_l + (lane & 31) * Lds<T, DH>::RSTRIDE;       // this lane's query row, in LDS (rows past the end are zero)
  LdsRowFrag<T, DH> qf;
  qf.load(q_row);
  ctx = min(ctx, sh.S_text);

  int tile_ctx = ctx;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tile_ctx = max(tile_ctx, __shfl_xor(tile_ctx, o));
  const int ntiles = min((tile_ctx + ATT_TILE - 1) / ATT_TILE, KVSPLIT ? t_lo + t_per : (1 << 28));

  BG_FTRACE(2);   // Q staged, barrier, reach
  // key and value rows of an entry in ONE round of loads where the registers allow (path -> pointer -> key -> value was four
  // round trips an entry), the first two entries of the wave in the SAME round (path tables up to 8 entries: all of them)
  // (Dh 256 bf16 with ONE entry's rows together was tried: the 128 registers push the full-range kernel back into spilling,
  // c4 root causal 29.7 -> 35.1 us, tree pass 43.7 -> 50.7 us; in the key-split kernel alone, whose branch steps spend 4.5 us on
  // every entry - benchmarks/probes/attn_fwd_trace.py c4 - it fits (450 registers) but the c4 step did not get faster, 73.4 ms,
  // and the bench's side replay of the recorded forward launches did not finish within its time limit with it: not understood,
  // not shipped)
  constexpr bool TOGETHER = DH * sizeof(T) <= 256;
  constexpr int NPRE = TOGETHER ? 2 : 0;
  const bool path_here = !KVSPLIT || split == n_splits - 1;          // key-range split: the branch keys ride with the last split
  const int64_t slot_off = (int64_t)b * sh.s_sb + (int64_t)hh * sh.s_sh + (int64_t)kq * sh.s_sk;
  [[maybe_unused]] PathRow<T, TOGETHER ? DH : 32> kvp[2];
  [[maybe_unused]] bool actp[2] = {false, false};
  auto rows_of = [&](int slot, bool act, const T*& kr, const T*& vr) __attribute__((always_inline)) {
    kr = nullptr;
    vr = nullptr;
    if (act && slot >= 0 && slot < sh.n_slots) {
      kr = (const T*)ptr_l[slot] + slot_off;
      vr = (const T*)ptr_l[BG_MAX_SLOTS + slot] + slot_off;
    }
  };
  auto preload = [&]() __attribute__((always_inline)) {
    if constexpr (NPRE > 0) {
#pragma unroll
      for (int e = 0; e < NPRE; ++e) {
        const int p = wave + e * ATT_WAVES;
        const T* kr;
        const T* vr;
        rows_of(e == 0 ? slot_first : slot_second, path_here && p < plen, kr, vr);
        actp[e] = path_here && p < plen && kr;
        kvp[e].load(kr, vr, h, actp[e]);
      }
    }
  };
  if constexpr (PF) preload();

  float m = -INFINITY, l = 0.f;
  f32x16 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

  for (int t = t_lo + wave; t < ntiles; t += ATT_WAVES) {
    const int kb = t * ATT_TILE;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // this wave's reads of the previous tile are done
    kreg.template store<true, false>(k_row, nullptr, lane);
    vreg.template store<true, false>(v_z, nullptr, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // LDS is in order within a wave: stores land before the reads below
    if (t + ATT_WAVES < ntiles) {
      kreg.load(kt, sh.t_ss, kb + ATT_WAVES * ATT_TILE, sh.S_text, lane);
      vreg.load(vt, sh.t_ss, kb + ATT_WAVES * ATT_TILE, sh.S_text, lane);
    }
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    mma_reduce_d<T, DH>(s, k_row, lane, [&](int tt) { return qf.get(tt, h); });
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb + tile_row(r, h);
      s[r] = (key < ctx) ? s[r] * sh.scale : -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __expf(m - m_use);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_use);
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32);
    l = l * alpha + psum;
    m = m_new;
    if constexpr (DROP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = drop_keep(drop, row_key, (uint32_t)(kb + tile_row(r, h))) ? s[r] : 0.f;
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    if constexpr (sizeof(T) == 4) mma_reduce_rows<T, DH, 0, NDT>(o, s, v_z, lane);
    else mma_reduce_rows_tr<DH, 0, NDT>(o, s, v_z, lane);
  }

  BG_FTRACE(3);   // text tiles
  // branch keys, path entries wave, wave + 4, ...  (key-range split: the last split's)
  auto entry = [&](float sc, bool act, int p, auto&& add_value) __attribute__((always_inline)) {
    if (act) {
      const float m_new = fmaxf(m, sc);
      const float alpha = __expf(m - m_new);
      const float pw = __expf(sc - m_new);
      l = l * alpha + pw;
      m = m_new;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      bool kept = true;
      if constexpr (DROP) kept = drop_keep(drop, row_key, (uint32_t)(sh.S_text + p));
      if (kept) add_value(pw);
    }
  };
  if constexpr (!PF) preload();
  if constexpr (NPRE > 0) {
#pragma unroll
    for (int e = 0; e < NPRE; ++e) {
      const int p = wave + e * ATT_WAVES;
      if (p < sh.max_path && path_here)                    // wave-uniform
        entry(kvp[e].dot(q_row, h, actp[e]) * sh.scale, actp[e], p, [&](float pw) { kvp[e].axpy(o, pw); });
    }
  }
  for (int p = path_here ? wave + NPRE * ATT_WAVES : sh.max_path; p < sh.max_path; p += ATT_WAVES) {
    const int slot = a.path[n * sh.max_path + p];          // in bounds for every p < max_path: no wait on path_len first
    const T* kr;
    const T* vr;
    rows_of(slot, p < plen, kr, vr);
    const bool act = p < plen && kr;
    if constexpr (TOGETHER) {
      PathRow<T, TOGETHER ? DH : 32> kv;
      kv.load(kr, vr, h, act);
      entry(kv.dot(q_row, h, act) * sh.scale, act, p, [&](float pw) { kv.axpy(o, pw); });
    } else {
      entry(pair_dot<T, DH>(q_row, kr, h, act) * sh.scale, act, p, [&](float pw) { axpy_rowvec<T, DH>(o, pw, vr, h, true); });
    }
  }

  BG_FTRACE(4);   // branch keys
  // merge the four waves' states:  mo[wave][d][query], ml[wave][{m, l}][query]
  __syncthreads();
  float* mo = (float*)smem;
  float* ml = mo + ATT_WAVES * DH * ATT_TILE;
  const int qcol = lane & 31;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) mo[(wave * DH + dt * 32 + tile_row(r, h)) * ATT_TILE + qcol] = o[dt][r];
  if (h == 0) {
    ml[(wave * 2 + 0) * ATT_TILE + qcol] = m;
    ml[(wave * 2 + 1) * ATT_TILE + qcol] = l;
  }
  __syncthreads();
  BG_FTRACE(5);   // states in LDS, barrier
  // thread -> query tid & 31, feature group tid >> 5 (8 groups of DH / 8 consecutive features)
  {
    constexpr int PER = DH / 8;
    const int qi = tid & 31, grp = tid >> 5;
    const int rq2 = blk.x * ATT_TILE + qi;
    if (rq2 < R) {
      [[maybe_unused]] const int n2 = rq2 / sh.K, k2 = rq2 % sh.K;
      float mw[ATT_WAVES], sw[ATT_WAVES], mstar = -INFINITY, lstar = 0.f;
#pragma unroll
      for (int w = 0; w < ATT_WAVES; ++w) {
        mw[w] = ml[(w * 2 + 0) * ATT_TILE + qi];
        mstar = fmaxf(mstar, mw[w]);
      }
#pragma unroll
      for (int w = 0; w < ATT_WAVES; ++w) {
        sw[w] = (mw[w] == -INFINITY) ? 0.f : __expf(mw[w] - mstar);
        lstar += ml[(w * 2 + 1) * ATT_TILE + qi] * sw[w];
      }
      if constexpr (KVSPLIT) {
        // this workgroup's state, unnormalised: the merge kernel scales it by exp(mstar - max over the splits)
        const int n_qtiles = (R + ATT_TILE - 1) / ATT_TILE;
        float* part = a.part + (((int64_t)bh * n_qtiles + blk.x) * n_splits + split) * fwd_part_floats<DH>();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < ATT_WAVES; ++w) v += mo[(w * DH + grp * PER + j) * ATT_TILE + qi] * sw[w];
          part[(grp * PER + j) * ATT_TILE + qi] = v;
        }
        if (grp == 0) {
          part[(DH + 0) * ATT_TILE + qi] = mstar;
          part[(DH + 1) * ATT_TILE + qi] = lstar;
        }
        return;
      }
      float inv = (lstar > 0.f) ? 1.f / lstar : 0.f;
      if constexpr (DROP) inv *= drop.inv_keep;
      T* o_row = (T*)a.out + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh + (int64_t)n2 * sh.o_sn + (int64_t)k2 * sh.o_sk + grp * PER;
      float acc[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) v += mo[(w * DH + grp * PER + j) * ATT_TILE + qi] * sw[w];
        acc[j] = v * inv;
      }
#pragma unroll
      for (int j0 = 0; j0 < PER; j0 += 4) {
        if constexpr (sizeof(T) == 4) {
          f32x4 ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) ov[j] = acc[j0 + j];
          *(f32x4*)(o_row + j0) = ov;
        } else {
          bf16x4 ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)acc[j0 + j];
          *(bf16x4*)(o_row + j0) = ov;
        }
      }
      if (grp == 0) a.lse[((int64_t)bh * sh.NQ + n2) * sh.K + k2] = (lstar > 0.f) ? mstar + logf(lstar) : -INFINITY;
    }
  }
  BG_FTRACE(6);   // stored
}

// The splits of a query tile combined: thread -> query tid & 31, feature group tid >> 5, as in the epilogue above.
template <typename T, int DH>
__global__ __launch_bounds__(256) void branch_attn_fwd_merge_kernel(AttnArgs a) {
  const bg_attn_shape& sh = a.sh;
  const int tid = threadIdx.x, qi = tid & 31, grp = tid >> 5;
  constexpr int PER = DH / 8;
  const int bh = blockIdx.y, b = bh / sh.H, hh = bh % sh.H;
  const int R = sh.NQ * sh.K, n_qtiles = (R + ATT_TILE - 1) / ATT_TILE, ns = a.kv_splits;
  const int rq = blockIdx.x * ATT_TILE + qi;
  if (rq >= R) return;
  const float* part = a.part + ((int64_t)bh * n_qtiles + blockIdx.x) * ns * fwd_part_floats<DH>();
  float mstar = -INFINITY, lstar = 0.f;
  for (int s = 0; s < ns; ++s) mstar = fmaxf(mstar, part[s * fwd_part_floats<DH>() + (DH + 0) * ATT_TILE + qi]);
  float acc[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) acc[j] = 0.f;
  for (int s = 0; s < ns; ++s) {
    const float* ps = part + s * fwd_part_floats<DH>();
    const float ms = ps[(DH + 0) * ATT_TILE + qi];
    const float w = (ms == -INFINITY) ? 0.f : __expf(ms - mstar);
    lstar += ps[(DH + 1) * ATT_TILE + qi] * w;
#pragma unroll
    for (int j = 0; j < PER; ++j) acc[j] = fmaf(ps[(grp * PER + j) * ATT_TILE + qi], w, acc[j]);
  }
  const float inv = (lstar > 0.f) ? 1.f / lstar : 0.f;
  const int n = rq / sh.K, k = rq % sh.K;
  T* o_row = (T*)a.out + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh + (int64_t)n * sh.o_sn + (int64_t)k * sh.o_sk + grp * PER;
#pragma unroll
  for (int j0 = 0; j0 < PER; j0 += 4) {
    if constexpr (sizeof(T) == 4) {
      f32x4 ov;
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[j] = acc[j0 + j] * inv;
      *(f32x4*)(o_row + j0) = ov;
    } else {
      bf16x4 ov;
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(acc[j0 + j] * inv);
      *(bf16x4*)(o_row + j0) = ov;
    }
  }
  if (grp == 0) a.lse[((int64_t)bh * sh.NQ + n) * sh.K + k] = (lstar > 0.f) ? mstar + logf(lstar) : -INFINITY;
}

// How many workgroups the key range of a query tile is dealt to (1 = no split, no workspace): few workgroups, many key tiles.
// BRANCHGAN_ATTN_FWD_KVSPLIT=0 switches it off, =N forces N splits wherever the shape allows (A/B runs).
int attn_fwd_kv_splits(const bg_attn_shape& sh) {
  static const int forced = [] { const char* e = getenv("BRANCHGAN_ATTN_FWD_KVSPLIT"); return e ? atoi(e) : -1; }();
  const bool fits = sh.Dh <= 128 || (sh.Dh == 256 && sh.dtype == BG_BF16);
  const int64_t wgs = cdiv((int64_t)sh.NQ * sh.K, ATT_TILE) * sh.B * sh.H;
  const int nk = (int)cdiv((int64_t)sh.S_text, ATT_TILE);
  if (!fits || forced == 0 || nk < 2 * ATT_WAVES || sh.p_drop > 0.f) return 1;
  int ns = forced > 0 ? forced : (int)cdiv(256, wgs);
  if (forced < 0 && (wgs > 128 || nk < 3 * ATT_WAVES)) return 1;
  ns = ns < nk / ATT_WAVES ? ns : nk / ATT_WAVES;          // every split has a tile for each of its four waves
  return ns < 1 ? 1 : (ns > 8 ? 8 : ns);
}

size_t attn_fwd_workspace_bytes(const bg_attn_shape& sh) {
  const int ns = attn_fwd_kv_splits(sh);
  if (ns <= 1) return 0;
  return (size_t)sh.B * sh.H * cdiv((int64_t)sh.NQ * sh.K, ATT_TILE) * ns * (sh.Dh + 2) * ATT_TILE * sizeof(float);
}

size_t attn_fwd_lds_bytes(int dtype, int Dh) {
  if (dtype == BG_F32) return (size_t)2 * ATT_TILE * (Dh + 1) * 4 + 64;
  return (size_t)(2 * ATT_TILE * (Dh + 8)) * 2 + 64;
}

int validate_attn_shape(const bg_attn_shape* s, const char* who) {
  BG_REQUIRE(s, "%s: null shape", who);
  BG_REQUIRE(s->dtype == BG_F32 || s->dtype == BG_BF16, "%s: dtype %d", who, s->dtype);
  BG_REQUIRE(s->Dh == 32 || s->Dh == 64 || s->Dh == 128 || s->Dh == 256, "%s: Dh=%d not in {32,64,128,256}", who, s->Dh);
  BG_REQUIRE(s->B > 0 && s->H > 0 && s->NQ > 0 && s->K > 0 && s->S_text >= 0, "%s: B=%d H=%d NQ=%d K=%d S=%d", who, s->B, s->H, s->NQ, s->K, s->S_text);
  BG_REQUIRE((int64_t)s->B * s->H < 65536, "%s: B*H=%lld exceeds grid.y", who, (long long)s->B * s->H);
  BG_REQUIRE(s->n_slots >= 0 && s->n_slots <= BG_MAX_SLOTS, "%s: n_slots=%d (max %d)", who, s->n_slots, BG_MAX_SLOTS);
  BG_REQUIRE(s->max_path >= 0 && s->max_path <= BG_MAX_SLOTS, "%s: max_path=%d", who, s->max_path);
  BG_REQUIRE(s->max_path == 0 || s->n_slots > 0, "%s: path entries but no slots", who);
  const int64_t al = (s->dtype == BG_F32) ? 4 : 8;  // elements per 16 bytes
  const int64_t st[] = {s->q_sb, s->q_sh, s->q_sn, s->q_sk, s->o_sb, s->o_sh, s->o_sn, s->o_sk, s->t_sb, s->t_sh, s->t_ss, s->s_sb, s->s_sh, s->s_sk};
  for (int64_t v : st) BG_REQUIRE(v >= 0 && v % al == 0, "%s: stride %lld is not a multiple of %lld elements (16 bytes)", who, (long long)v, (long long)al);
  BG_REQUIRE(s->p_drop >= 0.f && s->p_drop < 1.f, "%s: p_drop=%g is not in [0, 1)", who, (double)s->p_drop);
  if (s->p_drop > 0.f) {
    BG_REQUIRE(s->drop_state, "%s: p_drop > 0 needs drop_state (device: seed, step counter)", who);
    BG_REQUIRE(s->Dh <= 128, "%s: attention dropout is built for Dh <= 128 (Dh=%d)", who, s->Dh);
  }
  return BG_OK;
}

int check_aligned16(const void* p, const char* who, const char* name) {
  BG_REQUIRE(p && (((uintptr_t)p) & 15) == 0, "%s: %s is null or not 16-byte aligned", who, name);
  return BG_OK;
}

template <typename T, int DH>
static void launch_fwd_split(const AttnArgs& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds = SplitLds<T, DH>::BYTES;
  if (a.sh.p_drop > 0.f) {                               // validated by the entry point: Dh <= 128, no key-range split
    if constexpr (DH <= 128) {
      auto kernel = branch_attn_fwd_split_kernel<T, DH, false, true>;
      set_lds_limit(kernel, lds);
      hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
    }
    return;
  }
  if (a.kv_splits > 1) {
    auto kernel = branch_attn_fwd_split_kernel<T, DH, true>;
    set_lds_limit(kernel, lds);
    hipLaunchKernelGGL(kernel, dim3(grid.x * a.kv_splits, grid.y), dim3(256), lds, st, a);
    hipLaunchKernelGGL((branch_attn_fwd_merge_kernel<T, DH>), grid, dim3(256), 0, st, a);
    return;
  }
  // at most one workgroup per CU and branch keys to fetch: their rows are requested before the text tiles (PF)
  static const bool pf_on = [] { const char* e = getenv("BRANCHGAN_ATTN_FWD_PF"); return !(e && e[0] == '0'); }();   // A/B runs
  if constexpr (sizeof(T) == 2 && DH <= 128) {
    if (pf_on && a.sh.max_path > 0 && (int64_t)grid.x * grid.y <= 256) {   // (257-512 workgroups, c2's tree passes: 11.9 vs 11.7 us)
      auto kernel = branch_attn_fwd_split_kernel<T, DH, false, false, true>;
      set_lds_limit(kernel, lds);
      hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
      return;
    }
  }
  auto kernel = branch_attn_fwd_split_kernel<T, DH, false>;
  set_lds_limit(kernel, lds);
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
}

template <typename T>
static int launch_fwd(const AttnArgs& a, hipStream_t st) {
  const bg_attn_shape& sh = a.sh;
  dim3 grid((unsigned)cdiv((int64_t)sh.NQ * sh.K, ATT_WAVES * ATT_TILE), (unsigned)(sh.B * sh.H)), block(256);
  const size_t lds = attn_fwd_lds_bytes(sh.dtype, sh.Dh);
  dim3 grid_split((unsigned)cdiv((int64_t)sh.NQ * sh.K, ATT_TILE), (unsigned)(sh.B * sh.H));
  // few query rows per (batch, head): split the keys over the waves (4x the workgroups, 4x shorter chains); many rows:
  // the query-split kernel stages each K/V tile once for four waves (measured crossover at 128 rows, MI355X, Dh 64)
  const int64_t R = (int64_t)sh.NQ * sh.K;
  static const bool all_split = [] { const char* e = getenv("BRANCHGAN_ATTN_FWD_SPLIT_ALL"); return !(e && e[0] == '0'); }();   // A/B runs
  // Dh 256: the four waves' K / V tiles + the Q tile fit the 160 KB of LDS in bf16 only (152 KB; fp32 would need 280 KB)
  static const bool split_256 = [] { const char* e = getenv("BRANCHGAN_ATTN_FWD_SPLIT_256"); return !(e && e[0] == '0'); }();
  const bool fits = sh.Dh <= 128 || (sh.Dh == 256 && sizeof(T) == 2 && split_256);
  const bool split = fits && (all_split || a.kv_splits > 1 || sh.p_drop > 0.f || (R <= 128 && cdiv(R, ATT_WAVES * ATT_TILE) * sh.B * sh.H <= 512));
  if (split) {
    switch (sh.Dh) {
      case 32: launch_fwd_split<T, 32>(a, grid_split, st); break;
      case 64: launch_fwd_split<T, 64>(a, grid_split, st); break;
      case 128: launch_fwd_split<T, 128>(a, grid_split, st); break;
      default:
        if constexpr (sizeof(T) == 2) launch_fwd_split<T, 256>(a, grid_split, st);
        break;
    }
    return BG_OK;
  }
  switch (sh.Dh) {
    case 32: hipLaunchKernelGGL((branch_attn_fwd_kernel<T, 32>), grid, block, lds, st, a); break;
    case 64: hipLaunchKernelGGL((branch_attn_fwd_kernel<T, 64>), grid, block, lds, st, a); break;
    case 128: hipLaunchKernelGGL((branch_attn_fwd_kernel<T, 128>), grid, block, lds, st, a); break;
    default:
      set_lds_limit(branch_attn_fwd_kernel<T, 256>, lds);
      hipLaunchKernelGGL((branch_attn_fwd_kernel<T, 256>), grid, block, lds, st, a);
      break;
  }
  return BG_OK;
}

}  // namespace bg

using namespace bg;

#ifdef BG_ATTN_TRACE
extern "C" int bg_debug_attn_fwd_trace(unsigned long long* host, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bg::g_attn_fwd_trace), bytes < sizeof(bg::g_attn_fwd_trace) ? bytes : sizeof(bg::g_attn_fwd_trace));
}
#endif

extern "C" size_t bg_branch_attn_fwd_workspace_bytes(const bg_attn_shape* shape) {
  if (!shape || validate_attn_shape(shape, "bg_branch_attn_fwd_workspace_bytes")) return 0;
  return attn_fwd_workspace_bytes(*shape);
}

extern "C" int bg_branch_attn_fwd(const bg_attn_shape* shape, const void* q, const void* k_text, const void* v_text,
                                  const void* const* k_slots, const void* const* v_slots, const int32_t* ctx_len,
                                  const int32_t* path, const int32_t* path_len, void* out, float* lse, void* stream) {
  return bg_branch_attn_fwd_ws(shape, q, k_text, v_text, k_slots, v_slots, ctx_len, path, path_len, out, lse, nullptr, 0, stream);
}

extern "C" int bg_branch_attn_fwd_ws(const bg_attn_shape* shape, const void* q, const void* k_text, const void* v_text,
                                     const void* const* k_slots, const void* const* v_slots, const int32_t* ctx_len,
                                     const int32_t* path, const int32_t* path_len, void* out, float* lse, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  const char* who = "bg_branch_attn_fwd";
  if (int e = validate_attn_shape(shape, who)) return e;
  if (int e = check_aligned16(q, who, "q")) return e;
  if (int e = check_aligned16(out, who, "out")) return e;
  BG_REQUIRE(lse && ctx_len, "%s: null lse/ctx_len", who);
  if (shape->S_text > 0) {
    if (int e = check_aligned16(k_text, who, "k_text")) return e;
    if (int e = check_aligned16(v_text, who, "v_text")) return e;
  }
  BG_REQUIRE(shape->max_path == 0 || (path && path_len), "%s: null path/path_len", who);
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape;
  a.q = q; a.k_text = k_text; a.v_text = v_text; a.ctx_len = ctx_len; a.path = path; a.path_len = path_len;
  a.out = out; a.lse = lse;
  a.kv_splits = 1;
  if (workspace && shape->p_drop == 0.f) {               // enough workspace: the key range of a query tile may be split
    const size_t need = attn_fwd_workspace_bytes(*shape);
    if (need > 0 && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0) {
      a.kv_splits = attn_fwd_kv_splits(*shape);
      a.part = (float*)workspace;
    }
  }
  for (int i = 0; i < shape->n_slots; ++i) {
    if (int e = check_aligned16(k_slots[i], who, "k_slots[i]")) return e;
    if (int e = check_aligned16(v_slots[i], who, "v_slots[i]")) return e;
    a.k_slots[i] = k_slots[i];
    a.v_slots[i] = v_slots[i];
  }
  hipStream_t st = (hipStream_t)stream;
  if (shape->dtype == BG_F32) launch_fwd<float>(a, st);
  else launch_fwd<bf16_t>(a, st);
  BG_CHECK_LAUNCH(who);
  return BG_OK;
}
