// Branch attention backward, head-resident form (round 3): short sequences, bf16, Dh <= 64.
//
// At the BASELINE shapes a (batch, head) is small - configs[1]: 256 text rows and 96-256 query rows of 64 features - and the
// one-launch backward (attn_bwd_split.hip) spends its time on per-workgroup overheads, not on bytes or MFMAs: 1536 workgroups
// of a few microseconds each, in two residency rounds, every one of them fetching its own tile and gathering the others
// (profiles/r2_b_attn_bwd_split.txt).  Here Q, dO, K and V of one head are loaded into LDS ONCE (4 x 256 x 72 x 2 B = 147 KB of
// the CU's 160 KB), delta = rowsum(dO o O) is formed on the way, and after one barrier every wave works alone on whole output
// tiles straight out of LDS: a dQ tile walks the key tiles it can see, a dK / dV tile walks the query tiles that reach it, a
// slot's rows walk the nodes whose path holds it.  No staging copies, no gathers, no cross-wave merge (a wave owns its output
// tile, sums in tile order: deterministic), no second barrier.  `parts` workgroups per head share the tiles of a head between
// them (each loads the whole head - from L2 after the first) so that B*H = 96 heads still put 192 workgroups on 256 CUs.
// Same mathematics and per-element operations as attn_bwd_split.hip; only the grouping of the sums differs.
#include <stdlib.h>

#include "attn_common.h"

namespace bg {

constexpr int HEAD_THREADS = 512;
constexpr int HEAD_WAVES = HEAD_THREADS / 64;

struct HeadLds {
  int rpad, spad;                 // query rows / text rows, padded to whole tiles
  size_t q, dout, k, v;           // byte offsets of the four row-major images ([rows][DH + 8] bf16)
  size_t lse, delta, ctx, reach;  // fp32 [rpad], fp32 [rpad], int [rpad], int [rpad / 32]
  size_t bytes;
};

static inline HeadLds head_lds(int R, int S, int Dh) {
  HeadLds L;
  L.rpad = (int)cdiv(R, ATT_TILE) * ATT_TILE;
  L.spad = (int)cdiv(S, ATT_TILE) * ATT_TILE;
  const size_t row = (size_t)(Dh + 8) * 2;
  L.q = 0;
  L.dout = L.q + (size_t)L.rpad * row;
  L.k = L.dout + (size_t)L.rpad * row;
  L.v = L.k + (size_t)L.spad * row;
  L.lse = L.v + (size_t)L.spad * row;
  L.delta = L.lse + (size_t)L.rpad * 4;
  L.ctx = L.delta + (size_t)L.rpad * 4;
  L.reach = L.ctx + (size_t)L.rpad * 4;
  L.bytes = L.reach + (size_t)(L.rpad / ATT_TILE) * 4 + 16;
  return L;
}

struct HeadArgs {
  AttnArgs a;
  HeadLds L;
  int parts;
};

// The incoming gradient of this lane's key row (what the tree pass sent back to the text keys / values), requested BEFORE the
// tile loop: read inside the store, each 8-byte piece was a load waited for on the spot - sixteen dependent round trips at the
// end of every key tile.
template <int DH>
struct RowIn {
  bf16x4 v[DH / 8];
  bool has;
  __device__ __forceinline__ void load(const bf16_t* add, int h, bool valid) {
    has = add != nullptr;
#pragma unroll
    for (int i = 0; i < DH / 8; ++i) v[i] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (has && valid) {
#pragma unroll
      for (int i = 0; i < DH / 8; ++i) v[i] = *(const bf16x4*)(add + (i >> 2) * 32 + 8 * (i & 3) + 4 * h);
    }
  }
};

// dK or dV of this lane's key row: accumulators (rows = d) + optional incoming gradient -> global
template <int DH>
__device__ __forceinline__ void store_rowvec_add(const f32x16 (&acc)[DH / 32], bf16_t* row, const RowIn<DH>& in, int h, bool valid) {
  if (!valid) return;
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int off = dt * 32 + 8 * g + 4 * h;
      float x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = acc[dt][4 * g + j];
      if (in.has) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] += (float)in.v[dt * 4 + g][j];
      }
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16_t)x[j];
      *(bf16x4*)(row + off) = o;
    }
  }
}

template <int DH>
__global__ __launch_bounds__(HEAD_THREADS) void branch_attn_bwd_head_kernel(HeadArgs ha) {
  using T = bf16_t;
  constexpr int NDT = DH / 32, RS = DH + 8, CPR = DH / 8;       // LDS row stride (elements), 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = ha.a;
  const bg_attn_shape& sh = a.sh;
  const HeadLds& L = ha.L;
  T* q_l = (T*)(smem + L.q);
  T* do_l = (T*)(smem + L.dout);
  T* k_l = (T*)(smem + L.k);
  T* v_l = (T*)(smem + L.v);
  float* lse_l = (float*)(smem + L.lse);
  float* delta_l = (float*)(smem + L.delta);
  int* ctx_l = (int*)(smem + L.ctx);
  int* reach_l = (int*)(smem + L.reach);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  // the parts of one head on ONE XCD (blocks are dealt round-robin over the 8 XCDs, each with an L2 of its own): the second part
  // then finds the head in L2: 45.3 -> 28.8 MB of HBM traffic per launch at configs[1] (25.2 MB algorithmic; rocprofv3 --pmc
  // FETCH_SIZE / WRITE_SIZE, benchmarks/probes/pmc_attention.sh), 17.6 -> 17.0 us
  const BlockXY blk = xcd_block();
  const int part = blk.x, bh = blk.y, b = bh / sh.H, hh = bh % sh.H;
  const int R = sh.NQ * sh.K, S = sh.S_text;
  const int nq = L.rpad / ATT_TILE, nk = L.spad / ATT_TILE;
  const T* q_head = (const T*)a.q + (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh;
  const T* o_head = (const T*)a.out + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh;
  const T* do_head = (const T*)a.dout + (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh;
  const int64_t t_off = (int64_t)b * sh.t_sb + (int64_t)hh * sh.t_sh;
  const T* kt = (const T*)a.k_text + t_off;
  const T* vt = (const T*)a.v_text + t_off;

  // ---- the head, once, in ONE round of loads per thread (a round trip to HBM is ~2 us: three rounds - Q / dO / O, then K / V,
  // then the rows' lse / ctx - were a third of the kernel's time): jobs [0, rpad * CPR) are (Q, dO, O) row chunks, with delta
  // formed on the way, jobs behind them (K, V) row chunks; a wave's 64 jobs are of one kind (rpad * CPR is a multiple of 256).
  constexpr int UNR = 8;
  const int q_jobs = L.rpad * CPR, n_jobs = q_jobs + L.spad * CPR;
  float r_lse[2];
  int r_ctx[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {                                    // rows tid, tid + 512 (rpad <= 1024: checked on the host)
    const int rr = tid + u * HEAD_THREADS;
    const bool ok = rr < R;
    const int n = ok ? rr / sh.K : 0, k = ok ? rr % sh.K : 0;
    r_ctx[u] = ok ? min(a.ctx_len[k], S) : 0;                      // 0 for rows past the end: masks everything
    r_lse[u] = ok ? a.lse[((int64_t)bh * sh.NQ + n) * sh.K + k] : 0.f;
  }
  for (int c0 = tid; c0 < n_jobs; c0 += HEAD_THREADS * UNR) {
    bf16x8 g0[UNR], g1[UNR], g2[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int c = c0 + u * HEAD_THREADS;
#pragma unroll
      for (int j = 0; j < 8; ++j) { g0[u][j] = (T)0.f; g1[u][j] = (T)0.f; g2[u][j] = (T)0.f; }
      if (c < q_jobs) {
        const int rr = c / CPR, d0 = (c % CPR) * 8;
        if (rr < R) {
          const int n = rr / sh.K, k = rr % sh.K;
          const int64_t qo = (int64_t)n * sh.q_sn + (int64_t)k * sh.q_sk + d0, oo = (int64_t)n * sh.o_sn + (int64_t)k * sh.o_sk + d0;
          g0[u] = *(const bf16x8*)(q_head + qo);
          g1[u] = *(const bf16x8*)(do_head + oo);
          g2[u] = *(const bf16x8*)(o_head + oo);
        }
      } else if (c < n_jobs) {
        const int cc = c - q_jobs, rr = cc / CPR, d0 = (cc % CPR) * 8;
        if (rr < S) {
          g0[u] = *(const bf16x8*)(kt + (int64_t)rr * sh.t_ss + d0);
          g1[u] = *(const bf16x8*)(vt + (int64_t)rr * sh.t_ss + d0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int c = c0 + u * HEAD_THREADS;
      if (c < q_jobs) {
        const int rr = c / CPR, d0 = (c % CPR) * 8;
        *(bf16x8*)(q_l + rr * RS + d0) = g0[u];
        *(bf16x8*)(do_l + rr * RS + d0) = g1[u];
        float partial = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) partial = fmaf((float)g1[u][j], (float)g2[u][j], partial);
#pragma unroll
        for (int o = CPR / 2; o > 0; o >>= 1) partial += __shfl_xor(partial, o);   // the CPR lanes of one row sit side by side
        if (c % CPR == 0) delta_l[rr] = partial;
      } else if (c < n_jobs) {
        const int cc = c - q_jobs, rr = cc / CPR, d0 = (cc % CPR) * 8;
        *(bf16x8*)(k_l + rr * RS + d0) = g0[u];
        *(bf16x8*)(v_l + rr * RS + d0) = g1[u];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rr = tid + u * HEAD_THREADS;
    if (rr < L.rpad) { ctx_l[rr] = r_ctx[u]; lse_l[rr] = r_lse[u]; }
  }
  __syncthreads();
  for (int tq = wave; tq < nq; tq += HEAD_WAVES) {                  // reach of a query tile: the largest context among its rows
    int t = lane < ATT_TILE ? ctx_l[tq * ATT_TILE + lane] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t = max(t, __shfl_xor(t, o));
    if (lane == 0) reach_l[tq] = t;
  }
  __syncthreads();

  // ---- work items of this head, heaviest first under a causal mask: key tile 0, query tile nq-1, key tile 1, ...; then the slots.
  // Dealing: consecutive items (similar weight) go to DIFFERENT workgroups of the head, and inside a workgroup the k-th heaviest
  // item goes to wave k for k < 4 and to wave 11 - k after that, so that the two waves of a SIMD (w and w + 4 - a kernel that is
  // bound by instruction issue per SIMD, not by latency: two tiles per iteration with accumulators of their own changed
  // nothing) hold a heavy and a light item: root causal at S 256, 2 parts: 9 tile iterations on every SIMD instead of 14 / 12 / 10 / 6.
  // Later rounds (more items than seats) run back to front.
  const int n_tiles = nq + nk, n_items = n_tiles + sh.n_slots;
  const int seats = ha.parts * HEAD_WAVES;
  const int local_rank = wave < 4 ? wave : 11 - wave;
  for (int round = 0; round * seats < n_items; ++round) {
    const int j = local_rank * ha.parts + part;
    const int i = round * seats + ((round & 1) ? seats - 1 - j : j);
    if (i >= n_items) continue;
    int kind, idx;                                                   // 0 = dK / dV tile, 1 = dQ tile, 2 = slot rows
    if (i >= n_tiles) { kind = 2; idx = i - n_tiles; }
    else {
      const int pair = i >> 1, both = min(nq, nk);
      if (pair < both) { kind = i & 1; idx = kind == 0 ? pair : nq - 1 - pair; }
      else if (nk > nq) { kind = 0; idx = i - both; }                // the longer list's tail
      else { kind = 1; idx = nq - 1 - (i - both); }
    }

    if (kind == 1) {
      // ---- dQ of query tile idx: this lane's query row against the key tiles it can see ------------------------------------------
      const int rq = idx * ATT_TILE + (lane & 31);
      const bool valid = rq < R;
      const int n = valid ? rq / sh.K : 0, kq = valid ? rq % sh.K : 0;
      const T* q_row = q_l + rq * RS;
      const T* do_row = do_l + rq * RS;
      LdsRowFrag<T, DH> qf, dof;
      qf.load(q_row);
      dof.load(do_row);
      const int ctx = ctx_l[rq];
      const float lse = lse_l[rq], delta = delta_l[rq];
      const int ntiles = (reach_l[idx] + ATT_TILE - 1) / ATT_TILE;
      f32x16 dq[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
      for (int t = 0; t < ntiles; ++t) {
        const T* k_tile = k_l + t * ATT_TILE * RS;
        const T* v_tile = v_l + t * ATT_TILE * RS;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mma_reduce_d<T, DH>(s, k_tile, lane, [&](int tt) { return qf.get(tt, h); });
        mma_reduce_d<T, DH>(dp, v_tile, lane, [&](int tt) { return dof.get(tt, h); });
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * ATT_TILE + tile_row(r, h);
          const float p = (key < ctx) ? __expf(s[r] * sh.scale - lse) : 0.f;
          s[r] = p * (dp[r] - delta) * sh.scale;                     // dS^T
        }
        mma_reduce_rows_tr<DH, 0, NDT>(dq, s, k_tile, lane);
      }
      // branch keys: the same branch point of every slot on this node's path
      const int plen = (valid && sh.max_path) ? a.path_len[n] : 0;
      for (int p = 0; p < sh.max_path; ++p) {
        const int slot = a.path[n * sh.max_path + p];
        const int64_t off = (int64_t)b * sh.s_sb + (int64_t)hh * sh.s_sh + (int64_t)kq * sh.s_sk;
        const T* kr;
        const T* vr;
        slot_rows<T>(a, slot, p < plen, off, kr, vr);
        const bool act = p < plen && kr;
        BwdPathRow<T, DH> kv;                                        // key (both layouts) and value row in one round of loads
        kv.load(kr, vr, h, act);
        const float sc = kv.dot_k(q_row, h, act) * sh.scale;
        const float dpv = kv.dot_v(do_row, h, act);
        if (act) {
          const float pw = __expf(sc - lse);
          kv.axpy_k(dq, pw * (dpv - delta) * sh.scale);
        }
      }
      T* dq_row = (T*)a.dq + (int64_t)b * sh.dq_sb + (int64_t)hh * sh.dq_sh + (int64_t)n * sh.dq_sn + (int64_t)kq * sh.dq_sk;
      store_rowvec<T, DH>(dq, 1.f, dq_row, h, valid);
    } else if (kind == 0) {
      // ---- dK / dV of key tile idx: this lane's key row against the query tiles that reach it -------------------------------------
      const int key0 = idx * ATT_TILE, key = key0 + (lane & 31);
      const bool kvalid = key < S;
      const int64_t off = t_off + (int64_t)key * sh.t_ss;
      RowIn<DH> in_k, in_v;
      in_k.load(a.dk_text_in ? (const T*)a.dk_text_in + off : nullptr, h, kvalid);
      in_v.load(a.dv_text_in ? (const T*)a.dv_text_in + off : nullptr, h, kvalid);
      LdsRowFrag<T, DH> kf, vf;
      kf.load(k_l + key * RS);
      vf.load(v_l + key * RS);
      f32x16 dk[NDT], dv[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      for (int tq = nq - 1; tq >= 0; --tq) {
        if (reach_l[tq] <= key0) continue;                           // wave-uniform
        const T* q_tile = q_l + tq * ATT_TILE * RS;
        const T* do_tile = do_l + tq * ATT_TILE * RS;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        mma_reduce_d<T, DH>(s, q_tile, lane, [&](int t) { return kf.get(t, h); });       // S[query][key]
        mma_reduce_d<T, DH>(dp, do_tile, lane, [&](int t) { return vf.get(t, h); });     // dP[query][key]
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                // this lane's 16 query rows: four runs of four
          const int r0 = tq * ATT_TILE + 8 * g + 4 * h;
          const f32x4 lse4 = *(const f32x4*)(lse_l + r0);
          const f32x4 del4 = *(const f32x4*)(delta_l + r0);
          const i32x4 ctx4 = *(const i32x4*)(ctx_l + r0);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            const float p = (key < ctx4[j]) ? __expf(s[r] * sh.scale - lse4[j]) : 0.f;
            dp[r] = p * (dp[r] - del4[j]) * sh.scale;                // dS
            s[r] = p;                                                // P
          }
        }
        mma_reduce_rows_tr<DH, 0, NDT>(dv, s, do_tile, lane);        // dV^T += dO^T P
        mma_reduce_rows_tr<DH, 0, NDT>(dk, dp, q_tile, lane);        // dK^T += Q^T dS
      }
      store_rowvec_add<DH>(dk, (T*)a.dk_text + off, in_k, h, kvalid);
      store_rowvec_add<DH>(dv, (T*)a.dv_text + off, in_v, h, kvalid);
    } else {
      // ---- dK / dV of slot idx: eight branch points at a time (eight lanes each, DH / 8 features a lane), over the member nodes ----
      constexpr int PER = DH / 8;
      const int sub = lane & 7;
      const int su = __builtin_amdgcn_readfirstlane(idx);           // the wave's slot: scalar loads of its four pointers
      const T* kslot = (const T*)a.k_slots[su];
      const T* vslot = (const T*)a.v_slots[su];
      T* dkslot = (T*)a.dk_slots[su];
      T* dvslot = (T*)a.dv_slots[su];
      for (int k0 = 0; k0 < sh.K; k0 += 8) {
        const int kq = k0 + (lane >> 3);
        const bool valid = kq < sh.K;
        const int64_t soff = (int64_t)b * sh.s_sb + (int64_t)hh * sh.s_sh + (int64_t)(valid ? kq : 0) * sh.s_sk + sub * PER;
        float kk[PER], vv[PER], dk[PER], dv[PER];
#pragma unroll
        for (int c = 0; c < PER / 4; ++c) {
          bf16x4 tk = {(T)0.f, (T)0.f, (T)0.f, (T)0.f}, tv = tk;
          if (valid) { tk = *(const bf16x4*)(kslot + soff + 4 * c); tv = *(const bf16x4*)(vslot + soff + 4 * c); }
#pragma unroll
          for (int j = 0; j < 4; ++j) { kk[4 * c + j] = (float)tk[j]; vv[4 * c + j] = (float)tv[j]; }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) { dk[j] = 0.f; dv[j] = 0.f; }
        for (int n = 0; n < sh.NQ; ++n) {                            // wave-uniform membership test, nodes in ascending order
          const int plen = a.path_len[n];
          bool member = false;
          for (int p = 0; p < sh.max_path; ++p) member |= (a.path[n * sh.max_path + p] == idx) & (p < plen);
          if (!member) continue;
          const int rr = n * sh.K + (valid ? kq : 0);
          float sdot = 0.f, pdot = 0.f, qq[PER], dd[PER];
#pragma unroll
          for (int c = 0; c < PER / 4; ++c) {
            const bf16x4 tq4 = *(const bf16x4*)(q_l + rr * RS + sub * PER + 4 * c);
            const bf16x4 td4 = *(const bf16x4*)(do_l + rr * RS + sub * PER + 4 * c);
#pragma unroll
            for (int j = 0; j < 4; ++j) { qq[4 * c + j] = (float)tq4[j]; dd[4 * c + j] = (float)td4[j]; }
          }
#pragma unroll
          for (int j = 0; j < PER; ++j) { sdot = fmaf(qq[j], kk[j], sdot); pdot = fmaf(dd[j], vv[j], pdot); }
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) { sdot += __shfl_xor(sdot, o); pdot += __shfl_xor(pdot, o); }
          const float pw = valid ? __expf(sdot * sh.scale - lse_l[rr]) : 0.f;
          const float ds = valid ? pw * (pdot - delta_l[rr]) * sh.scale : 0.f;
#pragma unroll
          for (int j = 0; j < PER; ++j) { dk[j] = fmaf(ds, qq[j], dk[j]); dv[j] = fmaf(pw, dd[j], dv[j]); }
        }
        if (valid) {
#pragma unroll
          for (int c = 0; c < PER / 4; ++c) {
            bf16x4 ok, ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ok[j] = (T)dk[4 * c + j]; ov[j] = (T)dv[4 * c + j]; }
            *(bf16x4*)(dkslot + soff + 4 * c) = ok;
            *(bf16x4*)(dvslot + soff + 4 * c) = ov;
          }
        }
      }
    }
  }
}

// Takes the backward where it measured faster than the one-launch kernel (profiles/r3_l_attn_bwd_head.txt): root passes (no
// slots) of at least 64 heads whose four images fit the LDS - bf16, Dh 32 / 64, no dropout.  configs[1] (96 heads, S 256):
// 26.4 -> 24.0 us in the step's mix, 22.8 -> 17.6 us replayed back to back; c3 (128 heads, S 128): 14.1 -> 12.1 us.  NOT taken:
// tree passes (c2 25.6 vs 26.5 us, c3 36 vs 55 us: a query tile's chain of key tiles plus its path entries on ONE wave) and
// launches of few heads (c1, 32 heads: 7.7 vs 10.9 us).  What bounds it: a wave's tile is ~180 instructions with a dozen
// LDS-read -> MFMA waits, ~1 us, and the longest item of a head is 8 tiles; at 196 registers there is no room for the four
// waves per SIMD that would hide those waits (two tiles per iteration with accumulators of their own: no gain).
// BRANCHGAN_ATTN_BWD_HEAD=0 switches it off, =2 takes it for every shape that fits (A/B runs, tests).  Returns true when launched.
bool launch_bwd_head(const AttnArgs& a, hipStream_t st) {
  static const int mode = [] { const char* e = getenv("BRANCHGAN_ATTN_BWD_HEAD"); return e ? atoi(e) : 1; }();
  static const int forced_parts = [] { const char* e = getenv("BRANCHGAN_ATTN_BWD_HEAD_PARTS"); return e ? atoi(e) : 0; }();
  const bg_attn_shape& sh = a.sh;
  if (mode == 0 || sh.dtype != BG_BF16 || (sh.Dh != 32 && sh.Dh != 64) || sh.p_drop > 0.f || sh.S_text <= 0) return false;
  const int64_t R = (int64_t)sh.NQ * sh.K;
  if (R > 2 * HEAD_THREADS) return false;                       // the prologue keeps the lse / ctx of rows tid, tid + 512 in registers
  HeadArgs ha;
  ha.a = a;
  ha.L = head_lds((int)R, sh.S_text, sh.Dh);
  if (ha.L.bytes > 158 * 1024) return false;
  const int heads = sh.B * sh.H;
  if (mode == 1 && (sh.n_slots != 0 || heads < 64)) return false;
  const int items = ha.L.rpad / ATT_TILE + ha.L.spad / ATT_TILE + sh.n_slots;
  // Every part loads the whole head, so more parts only pay while the chip has CUs left for them: one workgroup per CU (LDS),
  // parts * heads <= 256.  (Measured at configs[1], 96 heads: 28.6 / 22.9 / 32.7 / 32.9 us for 1 / 2 / 3 / 4 parts.)
  int parts = forced_parts > 0 ? forced_parts : 256 / heads;
  parts = parts > 4 ? 4 : parts;
  if (parts * HEAD_WAVES > items) parts = (int)cdiv(items, HEAD_WAVES);
  if (parts < 1) parts = 1;
  ha.parts = parts;
  const dim3 grid((unsigned)parts, (unsigned)heads);
  if (sh.Dh == 32) {
    set_lds_limit(branch_attn_bwd_head_kernel<32>, ha.L.bytes);
    hipLaunchKernelGGL(branch_attn_bwd_head_kernel<32>, grid, dim3(HEAD_THREADS), ha.L.bytes, st, ha);
  } else {
    set_lds_limit(branch_attn_bwd_head_kernel<64>, ha.L.bytes);
    hipLaunchKernelGGL(branch_attn_bwd_head_kernel<64>, grid, dim3(HEAD_THREADS), ha.L.bytes, st, ha);
  }
  return true;
}

}  // namespace bg
