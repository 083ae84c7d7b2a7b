// Attention weights on request (include/branchgan.h: bg_branch_attn_weights).
//
// The reference returns the softmax weights next to the attention output (NeoXNonResidualAttention.py:224-232,
// NonResidualAttention.py:160-172) and its modules hand them out when output_attentions=True.  The fused kernels never form
// them; this kernel rebuilds them from q, the keys and the forward's log-sum-exp, in the reference's column order
// [text 0..S-1 | ancestor depth 1 (K columns) | ... | self].  A diagnostic path: one workgroup per query row, plain dot
// products, coalesced stores of the row; no attempt at MFMA.
#include "attn_common.h"
#include "attn_dropout.h"

namespace bg {

int validate_attn_shape(const bg_attn_shape* s, const char* who);     // attn_fwd.hip

template <typename T>
__global__ __launch_bounds__(256) void branch_attn_weights_kernel(AttnArgs a, float* weights, int W) {
  __shared__ float qs[256];
  const bg_attn_shape& sh = a.sh;
  const int tid = threadIdx.x;
  int64_t row = blockIdx.x;                               // (b, h, n, k), k fastest
  const int k = (int)(row % sh.K); row /= sh.K;
  const int n = (int)(row % sh.NQ); row /= sh.NQ;
  const int h = (int)(row % sh.H);
  const int b = (int)(row / sh.H);
  const T* q = (const T*)a.q + b * sh.q_sb + h * sh.q_sh + n * sh.q_sn + k * sh.q_sk;
  if (tid < sh.Dh) qs[tid] = to_f32(q[tid]);
  __syncthreads();
  const float lse = a.lse[(((int64_t)b * sh.H + h) * sh.NQ + n) * sh.K + k];
  // with dropout the reference hands out the weights AFTER the dropout (kept ones scaled by 1 / (1 - p)): the same mask here
  const bool dropping = sh.p_drop > 0.f;
  DropKeys drop = {0u, 0u, 0u, 0u, 1.f};
  uint32_t row_key = 0;
  if (dropping) {
    drop = drop_keys(a, b * sh.H + h);
    row_key = drop_row_key(drop, (uint32_t)(n * sh.K + k));
  }
  auto mask_of = [&](int column) -> float { return !dropping ? 1.f : (drop_keep(drop, row_key, (uint32_t)column) ? drop.inv_keep : 0.f); };
  float* w = weights + (int64_t)blockIdx.x * W;
  auto weight_of = [&](const T* key) -> float {
    float dot = 0.f;
    for (int d = 0; d < sh.Dh; ++d) dot += qs[d] * to_f32(key[d]);
    return __expf(dot * sh.scale - lse);
  };
  const int ctx = min(a.ctx_len[k], sh.S_text);
  const T* kt = (const T*)a.k_text + b * sh.t_sb + h * sh.t_sh;
  for (int c = tid; c < sh.S_text; c += 256) w[c] = c < ctx ? weight_of(kt + (int64_t)c * sh.t_ss) * mask_of(c) : 0.f;
  for (int c = sh.S_text + tid; c < W; c += 256) w[c] = 0.f;
  __syncthreads();
  // the node's own path: ancestors on the diagonal of their K-column block, the last entry (itself) in the last column
  const int plen = sh.max_path ? a.path_len[n] : 0;
  if (tid < plen) {
    const int slot = a.path[n * sh.max_path + tid];
    const T* key = (const T*)a.k_slots[slot] + b * sh.s_sb + h * sh.s_sh + k * sh.s_sk;
    const int col = tid == plen - 1 ? W - 1 : sh.S_text + tid * sh.K + k;
    w[col] = weight_of(key) * mask_of(sh.S_text + tid);
  }
}

}  // namespace bg

using namespace bg;

extern "C" int bg_branch_attn_weights(const bg_attn_shape* shape, const void* q, const void* k_text, const void* const* k_slots,
                                      const int32_t* ctx_len, const int32_t* path, const int32_t* path_len, const float* lse,
                                      float* weights, int W, void* stream) {
  const char* who = "bg_branch_attn_weights";
  if (int e = validate_attn_shape(shape, who)) return e;
  BG_REQUIRE(q && lse && ctx_len && weights, "%s: null q/lse/ctx_len/weights", who);
  BG_REQUIRE(shape->S_text == 0 || k_text, "%s: null k_text", who);
  BG_REQUIRE(shape->max_path == 0 || (path && path_len), "%s: null path/path_len", who);
  const int need = shape->S_text + (shape->max_path > 0 ? (shape->max_path - 1) * shape->K + 1 : 0);
  BG_REQUIRE(W >= need && W >= 1, "%s: W=%d < S_text + (max_path - 1) * K + 1 = %d", who, W, need);
  BG_REQUIRE(shape->max_path <= 256, "%s: max_path=%d", who, shape->max_path);
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape;
  a.q = q; a.k_text = k_text; a.ctx_len = ctx_len; a.path = path; a.path_len = path_len; a.lse = const_cast<float*>(lse);
  for (int i = 0; i < shape->n_slots; ++i) a.k_slots[i] = k_slots[i];
  const int64_t rows = (int64_t)shape->B * shape->H * shape->NQ * shape->K;
  BG_REQUIRE(rows < (1ll << 31), "%s: too many query rows", who);
  hipStream_t st = (hipStream_t)stream;
  if (shape->dtype == BG_F32) hipLaunchKernelGGL(branch_attn_weights_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, a, weights, W);
  else hipLaunchKernelGGL(branch_attn_weights_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, st, a, weights, W);
  BG_CHECK_LAUNCH(who);
  return BG_OK;
}
