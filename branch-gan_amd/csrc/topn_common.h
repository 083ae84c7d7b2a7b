// Shared device helpers of the top-n / top-p sampling kernels.
#pragma once
#include "common.h"

namespace bg {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t order_bits(float x) {
  if (x != x) return 0xFFFFFFFFu;  // every NaN sorts first (torch: NaN is the largest)
  x = x + 0.0f;                    // -0.0 -> +0.0 (they compare equal in torch.sort)
  uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_bits_inv(uint32_t o) {
  if (o == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ u64 make_item(float key, uint32_t idx) {
  return ((u64)order_bits(key) << 32) | (u64)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ u64 readlane64(u64 v, int l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}


// Inserts candidate c (known to beat the current n-th best) into the wave's sorted list (lane i = i-th best).
__device__ __forceinline__ void wave_insert(u64& item, u64 c, int n, int lane) {
  const int pos = __popcll(__ballot(item > c));  // sorted list: lanes 0..pos-1 beat c, pos < n
  const u64 up = __shfl_up(item, 1);
  if (lane == pos) item = c;
  else if (lane > pos) item = up;
  if (lane >= n) item = 0;
}

// The wave's running state: sorted list (lane i = i-th best), tau = its n-th item (0 while the list is not full) and
// tau_key = the float key of tau (NaN while not full, so that "key < tau_key" is false for every key).
struct WaveList {
  u64 item, tau;
  float tau_key;
  int n, lane;
  __device__ __forceinline__ void init(int n_, int lane_) { item = 0; tau = 0; tau_key = order_bits_inv(0); n = n_; lane = lane_; }
  __device__ __forceinline__ void offer(u64 c) {     // wave-uniform c
    if (c > tau) {
      wave_insert(item, c, n, lane);
      tau = readlane64(item, n - 1);
      tau_key = tau == 0 ? order_bits_inv(0) : order_bits_inv((uint32_t)(tau >> 32));
    }
  }
  // every lane may bring one item (0 = none): only the n best of them can enter the list, so they are picked by n rounds of a
  // wave-wide maximum (items are unique: the index is part of them) instead of offering all 64 one after the other
  __device__ __forceinline__ void offer_best(u64 mine) {
    for (int r = 0; r < n; ++r) {
      u64 mx = mine;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const u64 other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
      }
      if (mx <= tau) break;                              // wave-uniform (0 = nobody has anything left)
      offer(mx);
      if (mine == mx) mine = 0;
    }
  }
  // every lane may bring one item (0 = none); offered in lane order
  __device__ __forceinline__ void offer_lanes(u64 mine) {
    u64 bal = __ballot(mine != 0);
    while (bal) {
      const int L = __ffsll((unsigned long long)bal) - 1;
      bal &= bal - 1;
      offer(readlane64(mine, L));
    }
  }
};

// Rank-merge `count` items held in LDS into the best n; each thread owns items t, t+blockDim, ...
__device__ __forceinline__ void rank_merge_store(const u64* items, int count, int n, int64_t* out_ids, float* out_keys,
                                                 u64* out_partial) {
  for (int t = threadIdx.x; t < count; t += blockDim.x) {
    u64 mine = items[t];
    int rank = 0;
    for (int j = 0; j < count; ++j) rank += (items[j] > mine) ? 1 : 0;
    if (rank < n && mine != 0) {
      if (out_partial) {
        out_partial[rank] = mine;
      } else {
        out_ids[rank] = (int64_t)(0xFFFFFFFFu - (uint32_t)mine);
        if (out_keys) out_keys[rank] = order_bits_inv((uint32_t)(mine >> 32));
      }
    }
  }
}


}  // namespace bg
