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

// Dropout over the rows of a [batch][rows][cols] activation with the same generator (bg_dropout_rows): one wave per row, the
// row's key worked out once (Philox per wave on (site, batch), then the row finaliser), an element's decision ~10 integer
// operations; whole 16-byte vectors.  VEC = 1: the unaligned / odd-width fallback.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dropout_rows_kernel(const T* in, T* out, int batch, int rows, int cols, int64_t isr, int64_t osr,
                                                           int text_rows, uint32_t row0, float p, uint32_t site, const uint64_t* state) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;                 // row of the flattened [batch * rows]
  if (r >= (int64_t)batch * rows) return;
  const int b = (int)(r / rows), t = (int)(r % rows);
  const DropKeys d = drop_keys_of(state, site, 0x80000000u | (uint32_t)b, p, 0);
  const uint32_t key = drop_row_key(d, t < text_rows ? (uint32_t)t : row0 + (uint32_t)(t - text_rows));
  const T* src = in + r * isr;
  T* dst = out + r * osr;
  typedef T V __attribute__((ext_vector_type(VEC)));
  for (int c = lane * VEC; c < cols; c += 64 * VEC) {
    if constexpr (VEC == 1) {
      dst[c] = drop_keep(d, key, (uint32_t)c) ? (T)((float)src[c] * d.inv_keep) : (T)0.0f;
    } else {
      V v = *(const V*)(src + c);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = drop_keep(d, key, (uint32_t)(c + j)) ? (T)((float)v[j] * d.inv_keep) : (T)0.0f;
      *(V*)(dst + c) = v;
    }
  }
}

template <typename T, int VEC>
static void launch_dropout_rows(const void* in, void* out, int batch, int rows, int cols, int64_t isr, int64_t osr, int text_rows,
                                uint32_t row0, float p, uint32_t site, const uint64_t* state, hipStream_t st) {
  const int64_t total = (int64_t)batch * rows;
  hipLaunchKernelGGL((dropout_rows_kernel<T, VEC>), dim3((unsigned)cdiv(total, 4)), dim3(256), 0, st, (const T*)in, (T*)out, batch, rows,
                     cols, isr, osr, text_rows, row0, p, site, state);
}

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

extern "C" int bg_dropout_rows(const void* in, void* out, int dtype, int batch, int rows, int cols, int64_t in_row_stride,
                               int64_t out_row_stride, int text_rows, uint32_t row0, float p, uint32_t site, const uint64_t* drop_state,
                               void* stream) {
  const char* who = "bg_dropout_rows";
  BG_REQUIRE(in && out && drop_state, "%s: null pointer", who);
  BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16, "%s: dtype %d (BG_F32 / BG_BF16)", who, dtype);
  BG_REQUIRE(batch > 0 && batch < 4096 && rows > 0 && cols > 0 && text_rows >= 0, "%s: batch %d rows %d cols %d text_rows %d", who, batch,
             rows, cols, text_rows);
  BG_REQUIRE(p > 0.f && p < 1.f, "%s: p = %g is not in (0, 1)", who, (double)p);
  BG_REQUIRE(in_row_stride >= cols && out_row_stride >= cols, "%s: row strides %lld / %lld below cols %d", who, (long long)in_row_stride,
             (long long)out_row_stride, cols);
  const int64_t tree_rows = rows > text_rows ? rows - text_rows : 0;
  BG_REQUIRE((text_rows < rows ? text_rows : rows) <= (1 << 20) && (int64_t)row0 + tree_rows <= ((int64_t)1 << 31), "%s: logical rows out of range", who);
  BG_REQUIRE((int64_t)batch * rows / 4 + 1 < ((int64_t)1 << 31), "%s: too many rows", who);
  const hipStream_t st = (hipStream_t)stream;
  const size_t es = dtype == BG_F32 ? 4 : 2;
  const int vec = (int)(16 / es);
  const bool aligned = cols % vec == 0 && in_row_stride % vec == 0 && out_row_stride % vec == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0;
  if (dtype == BG_F32) {
    if (aligned) launch_dropout_rows<float, 4>(in, out, batch, rows, cols, in_row_stride, out_row_stride, text_rows, row0, p, site, drop_state, st);
    else launch_dropout_rows<float, 1>(in, out, batch, rows, cols, in_row_stride, out_row_stride, text_rows, row0, p, site, drop_state, st);
  } else {
    if (aligned) launch_dropout_rows<bf16_t, 8>(in, out, batch, rows, cols, in_row_stride, out_row_stride, text_rows, row0, p, site, drop_state, st);
    else launch_dropout_rows<bf16_t, 1>(in, out, batch, rows, cols, in_row_stride, out_row_stride, text_rows, row0, p, site, drop_state, st);
  }
  BG_CHECK_LAUNCH(who);
  return BG_OK;
}
