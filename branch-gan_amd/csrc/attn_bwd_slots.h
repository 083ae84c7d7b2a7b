// dK / dV of the branch-slot rows (attention backward): shared by the stand-alone slots kernel (attn_bwd.hip) and the
// one-launch backward (attn_bwd_split.hip).
#pragma once
#include "attn_common.h"
#include "attn_dropout.h"

namespace bg {

// ---------------------------------------------------------------------------------------------------
// grid (n_slots, B*H), 256 threads: 8 lanes per branch point k, each lane owns DH/8 contiguous features.
// The query nodes whose path holds this slot are listed in LDS first (all threads test one node each: ballot + popcount keep
// them in ascending order, so the sums below are taken in a fixed order); the list is then walked four members at a time
// with all their loads issued together - a walk over every node with a dependent load per test was 17 us of latency.
// (every vector of the row is requested before the first is converted: with the bf16 -> float conversion inside the `valid`
// branch the compiler had waited for each 8-byte load before it asked for the next - round 3)
template <typename T, int N>
__device__ __forceinline__ void slot_load(float (&out)[N], const T* p, bool valid) {
  typedef typename std::conditional<sizeof(T) == 4, f32x4, bf16x4>::type V;
  V raw[N / 4];
#pragma unroll
  for (int c = 0; c < N / 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[c][j] = (T)0.0f;
  if (valid) {
#pragma unroll
    for (int c = 0; c < N / 4; ++c) raw[c] = *(const V*)(p + 4 * c);
  }
#pragma unroll
  for (int c = 0; c < N / 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * c + j] = (float)raw[c][j];
}

template <typename T, int N>
__device__ __forceinline__ void slot_store(T* p, const float (&v)[N]) {
#pragma unroll
  for (int c = 0; c < N / 4; ++c) {
    if constexpr (sizeof(T) == 4) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = v[4 * c + j];
      *(f32x4*)(p + 4 * c) = o;
    } else {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[4 * c + j];
      *(bf16x4*)(p + 4 * c) = o;
    }
  }
}

// OWN_DELTA: delta = rowsum(dO o O) is worked out here from the member's O row instead of being read from a.delta - the
// form the one-launch backward uses, where no earlier kernel has written a.delta.
// DROP: attention dropout (attn_dropout.h); a member is listed with its path position (n | p << 20), the key column of this
// slot in the node's row being S_text + p.
template <typename T, int DH, bool OWN_DELTA, bool DROP = false>
__device__ __forceinline__ void bwd_slot_rows(const AttnArgs& a, int* s_members /* [256] */, int* s_count /* [ATT_WAVES] */, int slot, int bh) {
  constexpr int LPR = 8, PER = DH / LPR;
  constexpr int U = (PER <= 8 ? 4 : (PER <= 16 ? 2 : 1)) / (OWN_DELTA && PER <= 16 ? 2 : 1);   // U: members whose loads are in flight together
  static_assert(PER % 4 == 0, "slot kernel: four features per vector");
  const bg_attn_shape& sh = a.sh;
  const int b = bh / sh.H, hh = bh % sh.H;
  const int tid = threadIdx.x, sub = tid % LPR, lane = tid & 63, wave = tid >> 6;
  // the slot is the workgroup's: its four pointers are scalar loads from the argument buffer (a loop over the table that
  // compared every index with `slot` became a vector load per pointer, one round trip before the first row could be asked for)
  const int su = __builtin_amdgcn_readfirstlane(slot);
  const T* kslot = (const T*)a.k_slots[su];
  const T* vslot = (const T*)a.v_slots[su];
  T* dkslot = (T*)a.dk_slots[su];
  T* dvslot = (T*)a.dv_slots[su];

  [[maybe_unused]] DropKeys drop;
  if constexpr (DROP) drop = drop_keys(a, bh);

  for (int k0 = 0; k0 < sh.K; k0 += 256 / LPR) {
    const int kq = k0 + tid / LPR;
    const bool valid = kq < sh.K;
    const int64_t soff = (int64_t)b * sh.s_sb + (int64_t)hh * sh.s_sh + (int64_t)(valid ? kq : 0) * sh.s_sk + sub * PER;
    float kk[PER], vv[PER], dk[PER], dv[PER];
    slot_load<T, PER>(kk, kslot + soff, valid);
    slot_load<T, PER>(vv, vslot + soff, valid);
#pragma unroll
    for (int j = 0; j < PER; ++j) { dk[j] = 0.f; dv[j] = 0.f; }

    for (int n0 = 0; n0 < sh.NQ; n0 += 256) {
      const int nt = n0 + tid;
      bool member = false;
      [[maybe_unused]] int member_p = 0;
      if (nt < sh.NQ) {                                 // every entry of the row is loaded at once (no wait on path_len first)
        const int plen = sh.max_path ? a.path_len[nt] : 0;
        for (int p = 0; p < sh.max_path; ++p) {
          const bool hit = (a.path[nt * sh.max_path + p] == slot) & (p < plen);
          member |= hit;
          if constexpr (DROP) member_p = hit ? p : member_p;
        }
      }
      const unsigned long long bal = __ballot(member);
      __syncthreads();  // the previous chunk's list is no longer read
      if (lane == 0) s_count[wave] = __popcll(bal);
      __syncthreads();
      int base = 0, total = 0;
#pragma unroll
      for (int w = 0; w < ATT_WAVES; ++w) { const int c = s_count[w]; base += (w < wave) ? c : 0; total += c; }
      if (member) s_members[base + __popcll(bal & ((1ull << lane) - 1ull))] = DROP ? (nt | (member_p << 20)) : nt;
      __syncthreads();

      for (int m0 = 0; m0 < total; m0 += U) {
        float qq[U][PER], dd[U][PER], ls[U], de[U];
        [[maybe_unused]] float mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool on = valid && (m0 + u < total);
          int n = s_members[min(m0 + u, total - 1)];
          if constexpr (DROP) {
            const int p = n >> 20;
            n &= (1 << 20) - 1;
            mk[u] = drop_keep(drop, drop_row_key(drop, (uint32_t)(n * sh.K + kq)), (uint32_t)(sh.S_text + p)) ? drop.inv_keep : 0.f;
          }
          const int64_t qoff = (int64_t)b * sh.q_sb + (int64_t)hh * sh.q_sh + (int64_t)n * sh.q_sn + (int64_t)(valid ? kq : 0) * sh.q_sk + sub * PER;
          const int64_t ooff = (int64_t)b * sh.o_sb + (int64_t)hh * sh.o_sh + (int64_t)n * sh.o_sn + (int64_t)(valid ? kq : 0) * sh.o_sk + sub * PER;
          const int64_t stat = ((int64_t)bh * sh.NQ + n) * sh.K + (valid ? kq : 0);
          slot_load<T, PER>(qq[u], (const T*)a.q + qoff, on);
          slot_load<T, PER>(dd[u], (const T*)a.dout + ooff, on);
          ls[u] = on ? a.lse[stat] : 0.f;
          if constexpr (OWN_DELTA) {
            float oo[PER];
            slot_load<T, PER>(oo, (const T*)a.out + ooff, on);
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < PER; ++j) part = fmaf(dd[u][j], oo[j], part);
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) part += __shfl_xor(part, o);
            de[u] = part;
          } else {
            de[u] = on ? a.delta[stat] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (m0 + u >= total) break;  // workgroup-uniform
          float sdot = 0.f, pdot = 0.f;
#pragma unroll
          for (int j = 0; j < PER; ++j) {
            sdot = fmaf(qq[u][j], kk[j], sdot);
            pdot = fmaf(dd[u][j], vv[j], pdot);
          }
#pragma unroll
          for (int o = 1; o < LPR; o <<= 1) { sdot += __shfl_xor(sdot, o); pdot += __shfl_xor(pdot, o); }
          float pw = valid ? __expf(sdot * sh.scale - ls[u]) : 0.f;
          if constexpr (DROP) pdot *= mk[u];
          const float ds = valid ? pw * (pdot - de[u]) * sh.scale : 0.f;
          if constexpr (DROP) pw *= mk[u];
#pragma unroll
          for (int j = 0; j < PER; ++j) {
            dk[j] = fmaf(ds, qq[u][j], dk[j]);
            dv[j] = fmaf(pw, dd[u][j], dv[j]);
          }
        }
      }
    }
    if (valid) {
      slot_store<T, PER>(dkslot + soff, dk);
      slot_store<T, PER>(dvslot + soff, dv);
    }
  }
}

}  // namespace bg
