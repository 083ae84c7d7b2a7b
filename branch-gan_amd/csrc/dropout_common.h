// The counter-based dropout generator shared by the attention kernels (attn_dropout.h), bg_dropout_rows and the LayerNorm
// kernels' dropout variants (layer.hip): Philox4x32-10 once per wave -> two 32-bit keys, then two bijective 32-bit finalisers per
// element.  See attn_dropout.h for the design.
#pragma once
#include "common.h"

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

// row: the call's own row number n * K + k; the call's drop_row0 moves it to the logical row the mask is a function of
__device__ __forceinline__ uint32_t drop_row_key(const DropKeys& d, uint32_t row) { return mix_a((row + d.row0) ^ d.k0) + d.k1; }

__device__ __forceinline__ bool drop_keep(const DropKeys& d, uint32_t row_key, uint32_t column) {
  return mix_b(row_key + column * 0x9E3779B1u) >= d.thresh;
}

// The keys of row `r` of a flattened [batch][rows_per_batch][cols] activation (bg_row_dropout: text rows keep their numbers,
// tree rows count from row0): Philox on (site, batch), then the row finaliser.  -> d (threshold, scale) and the row key.
struct RowDropArgs {
  const uint64_t* state;      // nullptr: no dropout
  float p;
  uint32_t site, row0;
  int text_rows, rows_per_batch;
};
__device__ __forceinline__ uint32_t row_drop_key(const RowDropArgs& a, int64_t r, DropKeys& d) {
  const int b = (int)(r / a.rows_per_batch), t = (int)(r % a.rows_per_batch);
  d = drop_keys_of(a.state, a.site, 0x80000000u | (uint32_t)b, a.p, 0);
  return drop_row_key(d, t < a.text_rows ? (uint32_t)t : a.row0 + (uint32_t)(t - a.text_rows));
}

}  // namespace bg
