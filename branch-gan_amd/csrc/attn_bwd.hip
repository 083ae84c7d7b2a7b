// Branch attention backward: three deterministic kernels (no atomics), recomputing P from the saved LSE.
//
//   1. dq kernel      (query on the lane, same shape as the forward): delta = rowsum(dO∘O), dQ.
//   2. dkv kernel     (key on the lane): each wave owns 32 text keys and sweeps the query tiles that can see
//                     them; dK/dV accumulate in registers and are stored once (every text row is written).
//   3. slots kernel   per-row VALU: dK/dV of each branch slot = sum over the query nodes whose path holds it.
// Gradients of the reference ops cited in attn_fwd.hip (autograd of NeoXNonResidualAttention.py:183-232 /
// NonResidualAttention.py:119-172).  See include/branchgan.h: bg_branch_attn_bwd.
#include "attn_bwd_slots.h"

namespace bg {

size_t attn_fwd_lds_bytes(int dtype, int Dh);
int validate_attn_shape(const bg_attn_shape* s, const char* who);
int check_aligned16(const void* p, const char* who, const char* name);

// ---------------------------------------------------------------------------------------------------
template <typename T, int DH>
__global__ __launch_bounds__(256) void branch_attn_bwd_dq_kernel(AttnArgs a) {
  using L = Lds<T, DH>;
  constexpr int NDT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* k_row = (T*)smem;
  T* v_row = k_row + L::ROW_ELEMS;
  int* red = (int*)(v_row + L::ROW_ELEMS);

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

  const int64_t qoff = (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh + (int64_t)n * sh.q_sn + (int64_t)kq * sh.q_sk;
  const int64_t ooff = (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh + (int64_t)n * sh.o_sn + (int64_t)kq * sh.o_sk;
  const T* q_row = (const T*)a.q + qoff;
  const T* o_row = (const T*)a.out + ooff;
  const T* do_row = (const T*)a.dout + ooff;
  const T* kt = (const T*)a.k_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  const T* vt = (const T*)a.v_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  const int64_t stat = ((int64_t)bh * sh.NQ + n) * sh.K + kq;

  RowFrag<T, DH> qf, dof;
  qf.load(q_row, valid, h);
  dof.load(do_row, valid, h);
  const float delta = pair_dot<T, DH>(do_row, o_row, h, valid);
  const float lse = valid ? a.lse[stat] : 0.f;
  if (valid && h == 0) a.delta[stat] = delta;

  int wave_ctx = ctx;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_ctx = max(wave_ctx, __shfl_xor(wave_ctx, o));
  if (lane == 0) red[wave] = wave_ctx;
  __syncthreads();
  const int block_ctx = max(max(red[0], red[1]), max(red[2], red[3]));

  f32x16 dq[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  // software pipeline as in the forward: the global loads of tile i+1 are in flight while tile i is consumed
  TileRegs<T, DH, 256> kreg, vreg;
  if (block_ctx > 0) {
    kreg.load(kt, sh.t_ss, 0, sh.S_text, tid);
    vreg.load(vt, sh.t_ss, 0, sh.S_text, tid);
  }
  for (int kb = 0; kb < block_ctx; kb += ATT_TILE) {
    __syncthreads();  // previous tile fully consumed
    kreg.template store<true, false>(k_row, nullptr, tid);
    vreg.template store<true, false>(v_row, nullptr, tid);
    __syncthreads();
    if (kb + ATT_TILE < block_ctx) {
      kreg.load(kt, sh.t_ss, kb + ATT_TILE, sh.S_text, tid);
      vreg.load(vt, sh.t_ss, kb + ATT_TILE, sh.S_text, tid);
    }
    if (kb >= wave_ctx) continue;

    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    mma_reduce_d<T, DH>(s, k_row, lane, [&](int t) { return qf.get(t, h); });
    mma_reduce_d<T, DH>(dp, v_row, lane, [&](int t) { return dof.get(t, h); });
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb + tile_row(r, h);
      const float p = (key < ctx) ? __expf(s[r] * sh.scale - lse) : 0.f;
      s[r] = p * (dp[r] - delta) * sh.scale;  // dS^T
    }
    if constexpr (sizeof(T) == 4) mma_reduce_rows<T, DH, 0, NDT>(dq, s, k_row, lane);
    else mma_reduce_rows_tr<DH, 0, NDT>(dq, s, k_row, lane);
  }

  const int plen = (valid && sh.max_path) ? a.path_len[n] : 0;   // path tables may be NULL when max_path == 0
  for (int p = 0; p < sh.max_path; ++p) {
    const bool act = p < plen;
    const int slot = act ? a.path[n * sh.max_path + p] : 0;
    const int64_t off = (int64_t)b * sh.s_sb + (int64_t)hh * sh.s_sh + (int64_t)kq * sh.s_sk;
    const T* kr = nullptr;
    const T* vr = nullptr;
    for (int sidx = 0; sidx < sh.n_slots; ++sidx)
      if (sidx == slot) { kr = (const T*)a.k_slots[sidx] + off; vr = (const T*)a.v_slots[sidx] + off; }
    const float sc = pair_dot<T, DH>(q_row, kr, h, act) * sh.scale;
    const float dpv = pair_dot<T, DH>(do_row, vr, h, act);
    if (act) {
      const float pw = __expf(sc - lse);
      axpy_rowvec<T, DH>(dq, pw * (dpv - delta) * sh.scale, kr, h, true);
    }
  }
  const int64_t dqoff = (int64_t)b * sh.dq_sb + (int64_t)hh * sh.dq_sh + (int64_t)n * sh.dq_sn + (int64_t)kq * sh.dq_sk;
  store_rowvec<T, DH>(dq, 1.f, (T*)a.dq + dqoff, h, valid);
}

static size_t attn_dq_lds_bytes(int dtype, int Dh) {
  if (dtype == BG_F32) return (size_t)2 * ATT_TILE * (Dh + 1) * 4 + 64;
  return (size_t)(2 * ATT_TILE * (Dh + 8)) * 2 + 64;
}

// ---------------------------------------------------------------------------------------------------
// PASS splits the d range of the accumulators for DH = 256 (two launches, 128 d each).
template <typename T, int DH, int DT0, int DT1>
__global__ __launch_bounds__(256) void branch_attn_bwd_dkv_kernel(AttnArgs a) {
  using L = Lds<T, DH>;
  constexpr int NDT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* q_row_l = (T*)smem;
  T* do_row_l = q_row_l + L::ROW_ELEMS;
  float* lse_s = (float*)(do_row_l + L::ROW_ELEMS);
  float* delta_s = lse_s + ATT_TILE;
  int* ctx_s = (int*)(delta_s + ATT_TILE);

  const bg_attn_shape& sh = a.sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const BlockXY blk = xcd_block();
  const int bh = blk.y, b = bh / sh.H, hh = bh % sh.H;
  const int R = sh.NQ * sh.K;
  const int key0_block = blk.x * (ATT_WAVES * ATT_TILE);
  const int key0_wave = key0_block + wave * ATT_TILE;
  const int key = key0_wave + (lane & 31);
  const bool kvalid = key < sh.S_text;
  const T* k_rowp = (const T*)a.k_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh + (int64_t)key * sh.t_ss;
  const T* v_rowp = (const T*)a.v_text + (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh + (int64_t)key * sh.t_ss;
  RowFrag<T, DH> kf, vf;
  kf.load(k_rowp, kvalid, h);
  vf.load(v_rowp, kvalid, h);

  f32x16 dk[NDT], dv[NDT];
#pragma unroll
  for (int dt = DT0; dt < DT1; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  // Query tiles that reach this workgroup's keys, with the gather of tile i+1 (Q, dO rows at arbitrary strides: 16-byte chunks,
  // plus lse / delta / ctx of its queries) in flight while tile i is consumed.
  constexpr int VEC = 16 / sizeof(T);
  constexpr int CPR = DH / VEC;                          // 16-byte chunks per row
  constexpr int PER = (ATT_TILE * CPR + 255) / 256;      // chunks per thread
  typedef typename TileRegs<T, DH, 256>::V Chunk;
  struct Stage {                                         // one gathered query tile on its way to LDS
    Chunk q[PER], d[PER];
    float lse = 0.f, delta = 0.f;
    int ctx = 0;
  };
  Stage st_a, st_b;                                      // two tiles in flight: a gather has two iterations to land
  // max ctx over the queries of tile q0 (uniform over the workgroup); also leaves this lane's query in (n, kq, ctx)
  auto tile_reach = [&](int q0, int& n, int& kq, int& ctx) {
    const int rq = q0 + (lane & 31);
    const bool qv = rq < R;
    n = qv ? rq / sh.K : 0;
    kq = qv ? rq % sh.K : 0;
    ctx = qv ? min(a.ctx_len[kq], sh.S_text) : 0;
    int t = ctx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t = max(t, __shfl_xor(t, o));
    return t;
  };
  // The reach of every query tile is worked out ONCE, up front (the tiles spread over the waves, their ctx_len loads in
  // flight together): inside the sweep it decides which tile comes next, and a dependent global load there would put a full
  // memory latency on every iteration's critical path.
  int* reach_s = ctx_s + ATT_TILE;
  for (int tq = wave; tq * ATT_TILE < R; tq += ATT_WAVES) {
    int n, kq, ctx;
    const int reach = tile_reach(tq * ATT_TILE, n, kq, ctx);
    if (lane == 0) reach_s[tq] = reach;
  }
  __syncthreads();
  auto next_tile = [&](int q0, int& reach) {             // first tile at or after q0 that reaches the keys, or >= R
    for (; q0 < R; q0 += ATT_TILE) {
      reach = reach_s[q0 / ATT_TILE];
      if (reach > key0_block) break;
    }
    return q0;
  };
  auto gather = [&](int q0, Stage& st) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * 256;
      const int r = c / CPR, d0 = (c % CPR) * VEC;
      const int rr = q0 + r;
      const bool ok = c < ATT_TILE * CPR && rr < R;
      const int nn = ok ? rr / sh.K : 0, kk = ok ? rr % sh.K : 0;
      const int64_t qoff = (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh + (int64_t)nn * sh.q_sn + (int64_t)kk * sh.q_sk + d0;
      const int64_t ooff = (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh + (int64_t)nn * sh.o_sn + (int64_t)kk * sh.o_sk + d0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) { st.q[i][j] = (T)0.0f; st.d[i][j] = (T)0.0f; }
      if (ok) {
        st.q[i] = *(const Chunk*)((const T*)a.q + qoff);
        st.d[i] = *(const Chunk*)((const T*)a.dout + ooff);
      }
    }
    if (tid < ATT_TILE) {
      int n, kq;
      tile_reach(q0, n, kq, st.ctx);                     // wave 0 only: its own shuffle, same values
      const bool qv = q0 + tid < R;
      const int64_t stat = ((int64_t)bh * sh.NQ + n) * sh.K + kq;
      st.lse = qv ? a.lse[stat] : 0.f;
      st.delta = qv ? a.delta[stat] : 0.f;
    }
  };
  int tile_ctx = 0, next_ctx = 0, after_ctx = 0;
  int q0 = next_tile(0, tile_ctx);
  if (q0 < R) gather(q0, st_a);
  int q1 = next_tile(q0 + ATT_TILE, next_ctx);
  if (q1 < R) gather(q1, st_b);
  // one iteration: the tile held in `st` goes to LDS, `st` is refilled with the tile after next, the tile is consumed
  auto sweep = [&](Stage& st) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * 256;
      if (c >= ATT_TILE * CPR) continue;
      const int r = c / CPR, d0 = (c % CPR) * VEC;
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          q_row_l[r * L::RSTRIDE + d0 + j] = st.q[i][j];
          do_row_l[r * L::RSTRIDE + d0 + j] = st.d[i][j];
        }
      } else {
        *(bf16x8*)(q_row_l + r * L::RSTRIDE + d0) = st.q[i];       // one image: read by rows for S / dP, by columns
        *(bf16x8*)(do_row_l + r * L::RSTRIDE + d0) = st.d[i];      // (transposed LDS reads) for dV / dK
      }
    }
    if (tid < ATT_TILE) {
      lse_s[tid] = st.lse;
      delta_s[tid] = st.delta;
      ctx_s[tid] = st.ctx;  // 0 for invalid queries: masks everything
    }
    __syncthreads();
    const int q2 = next_tile(q1 + ATT_TILE, after_ctx);
    if (q2 < R) gather(q2, st);
    const int this_ctx = tile_ctx;
    q0 = q1;
    q1 = q2;
    tile_ctx = next_ctx;
    next_ctx = after_ctx;
    if (this_ctx <= key0_wave) return;  // wave-uniform: nothing in this tile sees this wave's keys

    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    // S[query][key] = Q·K^T, dP[query][key] = dO·V^T  (tile rows = queries, column = this lane's key)
    mma_reduce_d<T, DH>(s, q_row_l, lane, [&](int t) { return kf.get(t, h); });
    mma_reduce_d<T, DH>(dp, do_row_l, lane, [&](int t) { return vf.get(t, h); });
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile_row(r, h);
      const float p = (key < ctx_s[row]) ? __expf(s[r] * sh.scale - lse_s[row]) : 0.f;
      dp[r] = p * (dp[r] - delta_s[row]) * sh.scale;  // dS
      s[r] = p;                                       // P
    }
    if constexpr (sizeof(T) == 4) {
      mma_reduce_rows<T, DH, DT0, DT1>(dv, s, do_row_l, lane);     // dV^T += dO^T·P
      mma_reduce_rows<T, DH, DT0, DT1>(dk, dp, q_row_l, lane);     // dK^T += Q^T·dS
    } else {
      mma_reduce_rows_tr<DH, DT0, DT1>(dv, s, do_row_l, lane);
      mma_reduce_rows_tr<DH, DT0, DT1>(dk, dp, q_row_l, lane);
    }
  };
  while (q0 < R) {
    sweep(st_a);
    if (q0 >= R) break;
    sweep(st_b);
  }

  if (kvalid) {
    const int64_t row_off = (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh + (int64_t)key * sh.t_ss;
    T* dk_row = (T*)a.dk_text + row_off;
    T* dv_row = (T*)a.dv_text + row_off;
    // gradients other consumers of these keys / values already produced (same layout; may be the output rows themselves):
    // added in fp32 before the one rounding, instead of a separate add kernel over the stored result
    const T* dk_in = a.dk_text_in ? (const T*)a.dk_text_in + row_off : nullptr;
    const T* dv_in = a.dv_text_in ? (const T*)a.dv_text_in + row_off : nullptr;
#pragma unroll
    for (int dt = DT0; dt < DT1; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * h;
        float ik[4] = {0.f, 0.f, 0.f, 0.f}, iv[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (sizeof(T) == 4) {
          if (dk_in) { const f32x4 t = *(const f32x4*)(dk_in + d); for (int j = 0; j < 4; ++j) ik[j] = t[j]; }
          if (dv_in) { const f32x4 t = *(const f32x4*)(dv_in + d); for (int j = 0; j < 4; ++j) iv[j] = t[j]; }
          f32x4 ok, ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) { ok[j] = dk[dt][4 * g + j] + ik[j]; ov[j] = dv[dt][4 * g + j] + iv[j]; }
          *(f32x4*)(dk_row + d) = ok;
          *(f32x4*)(dv_row + d) = ov;
        } else {
          if (dk_in) { const bf16x4 t = *(const bf16x4*)(dk_in + d); for (int j = 0; j < 4; ++j) ik[j] = (float)t[j]; }
          if (dv_in) { const bf16x4 t = *(const bf16x4*)(dv_in + d); for (int j = 0; j < 4; ++j) iv[j] = (float)t[j]; }
          bf16x4 ok, ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) { ok[j] = (bf16_t)(dk[dt][4 * g + j] + ik[j]); ov[j] = (bf16_t)(dv[dt][4 * g + j] + iv[j]); }
          *(bf16x4*)(dk_row + d) = ok;
          *(bf16x4*)(dv_row + d) = ov;
        }
      }
    }
  }
}

static size_t attn_dkv_lds_bytes(int dtype, int Dh, int64_t n_queries) {
  const size_t reach = (size_t)cdiv(n_queries, ATT_TILE) * 4;          // reach_s: one int per query tile
  if (dtype == BG_F32) return (size_t)2 * ATT_TILE * (Dh + 1) * 4 + 3 * ATT_TILE * 4 + reach + 64;
  return (size_t)(2 * ATT_TILE * (Dh + 8)) * 2 + 3 * ATT_TILE * 4 + reach + 64;
}

// ---------------------------------------------------------------------------------------------------
// grid (n_slots, B*H), 256 threads: the slot rows' dK / dV (attn_bwd_slots.h)
template <typename T, int DH>
__global__ __launch_bounds__(256) void branch_attn_bwd_slots_kernel(AttnArgs a) {
  __shared__ int s_members[256];
  __shared__ int s_count[ATT_WAVES];
  // the slots of one head read the same member rows (q, dO of the nodes on whose path they lie): one XCD, one L2 fetch
  // (c4's tree pass: 88.7 MB of HBM traffic per dispatch with the slots dealt over the eight XCDs, profiles/r3_pmc_attn_bwd_c4.json)
  const BlockXY blk = xcd_block();
  bwd_slot_rows<T, DH, false>(a, s_members, s_count, blk.x, blk.y);
}

template <typename T> int launch_bwd_split(const AttnArgs& a, hipStream_t st);      // attn_bwd_split.hip: 0 no, 1 slots left, 2 all
bool attn_bwd_covers_dropout(const bg_attn_shape& sh);                                // attn_bwd_split.hip
bool launch_bwd_head(const AttnArgs& a, hipStream_t st);                              // attn_bwd_head.hip: short sequences, a head per workgroup

template <typename T>
static void launch_slots(const AttnArgs& a, hipStream_t st) {
  const bg_attn_shape& sh = a.sh;
  if (sh.n_slots == 0) return;
  const dim3 grid_s((unsigned)sh.n_slots, (unsigned)(sh.B * sh.H)), block(256);
  switch (sh.Dh) {
    case 32: hipLaunchKernelGGL((branch_attn_bwd_slots_kernel<T, 32>), grid_s, block, 0, st, a); break;
    case 64: hipLaunchKernelGGL((branch_attn_bwd_slots_kernel<T, 64>), grid_s, block, 0, st, a); break;
    case 128: hipLaunchKernelGGL((branch_attn_bwd_slots_kernel<T, 128>), grid_s, block, 0, st, a); break;
    default: hipLaunchKernelGGL((branch_attn_bwd_slots_kernel<T, 256>), grid_s, block, 0, st, a); break;
  }
}

template <typename T>
static void launch_bwd(const AttnArgs& a, hipStream_t st) {
  const bg_attn_shape& sh = a.sh;
  if (launch_bwd_head(a, st)) return;                  // the whole head in one CU's LDS (bf16, Dh <= 64, short sequences)
  if (const int done = launch_bwd_split<T>(a, st)) {   // the other axis split over the waves; one launch where it fits
    if (done == 1) launch_slots<T>(a, st);
    return;
  }
  const dim3 block(256);
  const dim3 grid_q((unsigned)cdiv((int64_t)sh.NQ * sh.K, ATT_WAVES * ATT_TILE), (unsigned)(sh.B * sh.H));
  const dim3 grid_k((unsigned)cdiv((int64_t)sh.S_text, ATT_WAVES * ATT_TILE), (unsigned)(sh.B * sh.H));
  const dim3 grid_s((unsigned)sh.n_slots, (unsigned)(sh.B * sh.H));
  const size_t lq = attn_dq_lds_bytes(sh.dtype, sh.Dh), lk = attn_dkv_lds_bytes(sh.dtype, sh.Dh, (int64_t)sh.NQ * sh.K);
#define BG_BWD_CASE(DH)                                                                                             \
  case DH: {                                                                                                        \
    set_lds_limit(branch_attn_bwd_dq_kernel<T, DH>, lq);                                                            \
    hipLaunchKernelGGL((branch_attn_bwd_dq_kernel<T, DH>), grid_q, block, lq, st, a);                               \
    if (sh.S_text > 0) {                                                                                            \
      if constexpr (DH <= 128) {                                                                                    \
        set_lds_limit(branch_attn_bwd_dkv_kernel<T, DH, 0, DH / 32>, lk);                                           \
        hipLaunchKernelGGL((branch_attn_bwd_dkv_kernel<T, DH, 0, DH / 32>), grid_k, block, lk, st, a);              \
      } else {                                                                                                      \
        set_lds_limit(branch_attn_bwd_dkv_kernel<T, DH, 0, DH / 64>, lk);                                           \
        set_lds_limit(branch_attn_bwd_dkv_kernel<T, DH, DH / 64, DH / 32>, lk);                                     \
        hipLaunchKernelGGL((branch_attn_bwd_dkv_kernel<T, DH, 0, DH / 64>), grid_k, block, lk, st, a);              \
        hipLaunchKernelGGL((branch_attn_bwd_dkv_kernel<T, DH, DH / 64, DH / 32>), grid_k, block, lk, st, a);        \
      }                                                                                                             \
    }                                                                                                               \
    if (sh.n_slots > 0) hipLaunchKernelGGL((branch_attn_bwd_slots_kernel<T, DH>), grid_s, block, 0, st, a);         \
  } break;
  switch (sh.Dh) {
    BG_BWD_CASE(32)
    BG_BWD_CASE(64)
    BG_BWD_CASE(128)
    BG_BWD_CASE(256)
  }
#undef BG_BWD_CASE
}

}  // namespace bg

using namespace bg;

extern "C" int bg_branch_attn_bwd(const bg_attn_shape* shape, const void* q, const void* k_text, const void* v_text,
                                  const void* const* k_slots, const void* const* v_slots, const int32_t* ctx_len,
                                  const int32_t* path, const int32_t* path_len, const void* out, const float* lse,
                                  const void* dout, void* dq, void* dk_text, void* dv_text, const void* dk_text_in,
                                  const void* dv_text_in, void* const* dk_slots, void* const* dv_slots, float* delta,
                                  void* stream) {
  const char* who = "bg_branch_attn_bwd";
  if (int e = validate_attn_shape(shape, who)) return e;
  if (int e = check_aligned16(q, who, "q")) return e;
  if (int e = check_aligned16(out, who, "out")) return e;
  if (int e = check_aligned16(dout, who, "dout")) return e;
  if (int e = check_aligned16(dq, who, "dq")) return e;
  BG_REQUIRE(lse && ctx_len && delta, "%s: null lse/ctx_len/delta", who);
  if (shape->S_text > 0) {
    if (int e = check_aligned16(k_text, who, "k_text")) return e;
    if (int e = check_aligned16(v_text, who, "v_text")) return e;
    if (int e = check_aligned16(dk_text, who, "dk_text")) return e;
    if (int e = check_aligned16(dv_text, who, "dv_text")) return e;
    if (dk_text_in) if (int e = check_aligned16(dk_text_in, who, "dk_text_in")) return e;
    if (dv_text_in) if (int e = check_aligned16(dv_text_in, who, "dv_text_in")) return e;
  }
  BG_REQUIRE(shape->max_path == 0 || (path && path_len), "%s: null path/path_len", who);
  BG_REQUIRE(shape->p_drop == 0.f || (attn_bwd_covers_dropout(*shape) && shape->NQ < (1 << 20)),
             "%s: attention dropout needs the one-launch backward (Dh <= 128, a (batch, head) slice within 2^31 elements, "
             "NQ*K <= 131072 rows, NQ < 2^20)", who);
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape;
  if (!(a.sh.dq_sb | a.sh.dq_sh | a.sh.dq_sn | a.sh.dq_sk)) {
    a.sh.dq_sb = a.sh.o_sb; a.sh.dq_sh = a.sh.o_sh; a.sh.dq_sn = a.sh.o_sn; a.sh.dq_sk = a.sh.o_sk;
  }
  {
    const int64_t al = (a.sh.dtype == BG_F32) ? 4 : 8;
    const int64_t dqs[] = {a.sh.dq_sb, a.sh.dq_sh, a.sh.dq_sn, a.sh.dq_sk};
    for (int64_t v : dqs) BG_REQUIRE(v >= 0 && v % al == 0, "%s: dq stride %lld is not a multiple of %lld elements", who, (long long)v, (long long)al);
  }
  a.q = q; a.k_text = k_text; a.v_text = v_text; a.ctx_len = ctx_len; a.path = path; a.path_len = path_len;
  a.out = const_cast<void*>(out); a.lse = const_cast<float*>(lse); a.dout = dout;
  a.dq = dq; a.dk_text = dk_text; a.dv_text = dv_text; a.delta = delta;
  a.dk_text_in = dk_text_in; a.dv_text_in = dv_text_in;
  for (int i = 0; i < shape->n_slots; ++i) {
    if (int e = check_aligned16(k_slots[i], who, "k_slots[i]")) return e;
    if (int e = check_aligned16(v_slots[i], who, "v_slots[i]")) return e;
    if (int e = check_aligned16(dk_slots[i], who, "dk_slots[i]")) return e;
    if (int e = check_aligned16(dv_slots[i], who, "dv_slots[i]")) return e;
    a.k_slots[i] = k_slots[i]; a.v_slots[i] = v_slots[i];
    a.dk_slots[i] = dk_slots[i]; a.dv_slots[i] = dv_slots[i];
  }
  hipStream_t st = (hipStream_t)stream;
  if (shape->dtype == BG_F32) launch_bwd<float>(a, st);
  else launch_bwd<bf16_t>(a, st);
  BG_CHECK_LAUNCH(who);
  return BG_OK;
}
