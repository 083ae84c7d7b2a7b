// Dropout on the attention probabilities, inside the attention kernels (forward mask regenerated in backward: nothing stored).
//
// Replaces the always-on F.dropout(attn_weights, dropoutRate or attn_pdrop) of Models/GPT2BranchAttention/
// NonResidualAttention.py:107-109,157-159 and nn.Dropout(config.attention_dropout) of Models/NeoXBranchAttention/
// NeoXNonResidualAttention.py:224,287: the softmax is normalised over ALL visible keys, then each weight is kept with
// probability 1 - p and scaled by 1 / (1 - p).
//
// The keep decision of (batch*head bh, query row, key column) is a pure function of
//   (seed, step counter)  - 2 x u64 in device memory (bg_attn_shape.drop_state), read by the kernel when it RUNS, so a captured
//                           graph draws a fresh mask on every replay once the counter is advanced (bg_dropout_advance);
//   drop_call             - which attention call of the step this is (layer, pass): the same number in forward and backward;
//   bh, row, column       - row = drop_row0 + n * K + k of the query, column = text key index, or S_text + p for entry p of the
//                           node's path.
// Philox4x32-10 (Salmon et al. 2011) is run ONCE per wave on (drop_call, bh, counter) under the seed and yields two 32-bit keys;
// an element's 32 random bits are then mix_b(mix_a(row ^ k0) + k1 + column * odd) with two bijective 32-bit finalisers
// (murmur3's and "lowbias32"): ~10 integer operations per score instead of a Philox block per four, and the same value whichever
// kernel (forward: query on the lane; dK / dV: key on the lane; slot rows) asks for it.  Within a row distinct columns get
// distinct 32-bit values (every step is a bijection), kept iff value >= p * 2^32.
#pragma once
#include "attn_common.h"

namespace bg {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ uint32_t mix_a(uint32_t x) {      // murmur3 fmix32
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t mix_b(uint32_t x) {      // lowbias32
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

struct DropKeys {
  uint32_t k0, k1, thresh, row0;
  float inv_keep;
};

__device__ __forceinline__ DropKeys drop_keys_of(const uint64_t* state, uint32_t site, uint32_t stream_id, float p, uint32_t row0) {
  DropKeys d;
  const uint64_t seed = state[0], step = state[1];
  uint32_t c[4] = {site, stream_id, (uint32_t)step, (uint32_t)(step >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  d.k0 = c[0]; d.k1 = c[1];
  // p in (0, 1): keep iff bits >= p * 2^32 (p * 2^32 < 2^32 fits; computed in double on the host side of the same formula)
  d.thresh = (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f);
  d.inv_keep = 1.0f / (1.0f - p);
  d.row0 = row0;
  return d;
}

// Wave-uniform: every lane computes the same keys (the loads are scalar loads of kernel-argument pointers).
__device__ __forceinline__ DropKeys drop_keys(const AttnArgs& a, int bh) {
  return drop_keys_of(a.sh.drop_state, a.sh.drop_call, (uint32_t)bh, a.sh.p_drop, a.sh.drop_row0);
}

// row: the call's own row number n * K + k; the call's drop_row0 moves it to the logical row the mask is a function of
__device__ __forceinline__ uint32_t drop_row_key(const DropKeys& d, uint32_t row) { return mix_a((row + d.row0) ^ d.k0) + d.k1; }

__device__ __forceinline__ bool drop_keep(const DropKeys& d, uint32_t row_key, uint32_t column) {
  return mix_b(row_key + column * 0x9E3779B1u) >= d.thresh;
}

}  // namespace bg
