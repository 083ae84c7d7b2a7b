// NeoX attention prologue: split the fused QKV projection, bring heads forward, apply partial rotary.
//
// Replaces Models/NeoXBranchAttention/NeoXNonResidualAttention.py:94-115 (view / slices / permute / cat) and
// :418-432 (rotate_half + apply_rotary_pos_emb with its per-layer gather of (B,1,T,rot) index tensors).
// See include/branchgan.h: bg_neox_qkv_rope_fwd / _bwd.  Elementwise, HBM-bound; one thread per 16 bytes of output.
#include "common.h"

namespace bg {

struct RopeArgs {
  const void* qkv;  // fwd: [B][T][H][3*Dh]; bwd: output dqkv
  void *q, *k, *v;  // fwd: outputs [B][H][T][Dh]; bwd: inputs dq, dk, dv
  const float *cos_tab, *sin_tab;
  const int32_t* pos;
  int B, T, H, Dh, rot, max_pos;
};

// x' = x*cos + rotate_half(x)*sin on d < rot.  The backward is the transpose of that rotation:
// dx[d] = dy[d]*cos[d] + (d < rot/2 ? dy[d+rot/2]*sin[d+rot/2] : -dy[d-rot/2]*sin[d-rot/2]).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void neox_qkv_rope_kernel(RopeArgs a) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = a.Dh / VEC;  // chunks per (b,t,h) row of one of q/k/v
  const int64_t total = (int64_t)a.B * a.T * a.H * 3 * cpr;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % cpr);
    const int which = (int)((idx / cpr) % 3);
    const int h = (int)((idx / (3 * cpr)) % a.H);
    const int t = (int)((idx / ((int64_t)3 * cpr * a.H)) % a.T);
    const int b = (int)(idx / ((int64_t)3 * cpr * a.H * a.T));
    const int d0 = c * VEC;
    T* fused = (T*)a.qkv + (((int64_t)b * a.T + t) * a.H + h) * 3 * a.Dh + which * a.Dh;
    T* split = (T*)(which == 0 ? a.q : which == 1 ? a.k : a.v) + (((int64_t)b * a.H + h) * a.T + t) * a.Dh;
    const T* src = BWD ? split : fused;
    T* dst = BWD ? fused : split;
    float x[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) x[j] = to_f32(src[d0 + j]);
    if (which < 2 && d0 < a.rot) {
      int p = a.pos[b * a.T + t];
      p = min(max(p, 0), a.max_pos - 1);
      const float* cs = a.cos_tab + (int64_t)p * a.rot;
      const float* sn = a.sin_tab + (int64_t)p * a.rot;
      const int half = a.rot / 2;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int d = d0 + j;
        if (d < a.rot) {
          const int dp = d < half ? d + half : d - half;
          const float other = to_f32(src[dp]);
          float r;
          if constexpr (!BWD) r = (d < half ? -other : other) * sn[d];
          else r = (d < half ? other : -other) * sn[dp];
          x[j] = x[j] * cs[d] + r;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) dst[d0 + j] = from_f32<T>(x[j]);
  }
}

static int check_rope(const char* who, int dtype, int B, int T, int H, int Dh, int rot, int max_pos) {
  BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16, "%s: dtype %d", who, dtype);
  BG_REQUIRE(B > 0 && T > 0 && H > 0 && Dh > 0 && Dh % 8 == 0, "%s: B=%d T=%d H=%d Dh=%d (Dh must be a multiple of 8)", who, B, T, H, Dh);
  BG_REQUIRE(rot >= 0 && rot <= Dh && rot % 2 == 0 && max_pos > 0, "%s: rot=%d max_pos=%d", who, rot, max_pos);
  return BG_OK;
}

template <bool BWD>
static void launch_rope(const RopeArgs& a, int dtype, hipStream_t st) {
  const int vec = dtype == BG_F32 ? 4 : 8;
  const int64_t total = (int64_t)a.B * a.T * a.H * 3 * (a.Dh / vec);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 8192);
  if (dtype == BG_F32) hipLaunchKernelGGL((neox_qkv_rope_kernel<float, BWD>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((neox_qkv_rope_kernel<bf16_t, BWD>), dim3(grid), dim3(256), 0, st, a);
}

}  // namespace bg

using namespace bg;

extern "C" int bg_neox_qkv_rope_fwd(const void* qkv, int dtype, int B, int T, int H, int Dh, int rot, const float* cos_tab,
                                    const float* sin_tab, int max_pos, const int32_t* position_ids, void* q, void* k, void* v,
                                    void* stream) {
  if (int e = check_rope("bg_neox_qkv_rope_fwd", dtype, B, T, H, Dh, rot, max_pos)) return e;
  BG_REQUIRE(qkv && q && k && v && position_ids && (rot == 0 || (cos_tab && sin_tab)), "bg_neox_qkv_rope_fwd: null pointer");
  RopeArgs a{qkv, q, k, v, cos_tab, sin_tab, position_ids, B, T, H, Dh, rot, max_pos};
  launch_rope<false>(a, dtype, (hipStream_t)stream);
  BG_CHECK_LAUNCH("bg_neox_qkv_rope_fwd");
  return BG_OK;
}

extern "C" int bg_neox_qkv_rope_bwd(const void* dq, const void* dk, const void* dv, int dtype, int B, int T, int H, int Dh,
                                    int rot, const float* cos_tab, const float* sin_tab, int max_pos,
                                    const int32_t* position_ids, void* dqkv, void* stream) {
  if (int e = check_rope("bg_neox_qkv_rope_bwd", dtype, B, T, H, Dh, rot, max_pos)) return e;
  BG_REQUIRE(dqkv && dq && dk && dv && position_ids && (rot == 0 || (cos_tab && sin_tab)), "bg_neox_qkv_rope_bwd: null pointer");
  RopeArgs a{dqkv, const_cast<void*>(dq), const_cast<void*>(dk), const_cast<void*>(dv), cos_tab, sin_tab, position_ids, B, T, H, Dh, rot, max_pos};
  launch_rope<true>(a, dtype, (hipStream_t)stream);
  BG_CHECK_LAUNCH("bg_neox_qkv_rope_bwd");
  return BG_OK;
}
