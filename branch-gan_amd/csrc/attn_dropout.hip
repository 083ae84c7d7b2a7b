// The keep mask of the attention dropout written out, and the step counter of its state (include/branchgan.h:
// bg_attn_dropout_mask, bg_dropout_advance).  The attention kernels regenerate the mask where they need it (attn_dropout.h);
// this file only exposes the same function to tests and keeps the counter.
#include "attn_dropout.h"

namespace bg {

int validate_attn_shape(const bg_attn_shape* s, const char* who);     // attn_fwd.hip

// grid (ceil(W / 256), R, B*H): one thread per (query row, key column)
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(AttnArgs a, uint8_t* keep, int W) {
  const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, bh = blockIdx.z;
  const DropKeys d = drop_keys(a, bh);
  if (col >= W) return;
  const int64_t R = (int64_t)a.sh.NQ * a.sh.K;
  keep[((int64_t)bh * R + row) * W + col] = drop_keep(d, drop_row_key(d, (uint32_t)row), (uint32_t)col) ? 1 : 0;
}

__global__ void dropout_advance_kernel(uint64_t* state) { state[1] += 1; }

}  // namespace bg

using namespace bg;

extern "C" int bg_attn_dropout_mask(const bg_attn_shape* shape, uint8_t* keep, void* stream) {
  const char* who = "bg_attn_dropout_mask";
  if (int e = validate_attn_shape(shape, who)) return e;
  BG_REQUIRE(shape->p_drop > 0.f && keep, "%s: p_drop must be > 0 and keep non-null", who);
  const int64_t R = (int64_t)shape->NQ * shape->K;
  const int W = shape->S_text + shape->max_path;
  BG_REQUIRE(R < 65536 && W > 0, "%s: NQ*K=%lld rows (max 65535), %d columns", who, (long long)R, W);
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.sh = *shape;
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)cdiv(W, 256), (unsigned)R, (unsigned)(shape->B * shape->H)), dim3(256), 0,
                     (hipStream_t)stream, a, keep, W);
  BG_CHECK_LAUNCH(who);
  return BG_OK;
}

extern "C" int bg_dropout_advance(uint64_t* drop_state, void* stream) {
  BG_REQUIRE(drop_state, "bg_dropout_advance: null state");
  hipLaunchKernelGGL(dropout_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, drop_state);
  BG_CHECK_LAUNCH("bg_dropout_advance");
  return BG_OK;
}
