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
#include "dropout_common.h"

namespace bg {

// Wave-uniform: every lane computes the same keys (the loads are scalar loads of kernel-argument pointers).
__device__ __forceinline__ DropKeys drop_keys(const AttnArgs& a, int bh) {
  return drop_keys_of(a.sh.drop_state, a.sh.drop_call, (uint32_t)bh, a.sh.p_drop, a.sh.drop_row0);
}

}  // namespace bg
