// Row-block gather: copies any number of 2-D sources (rows x row_bytes, their own row stride) into contiguous destinations in
// ONE launch.  Used to stack the saved activations and output gradients of the layers of one weight kind - separate allocations,
// some with padded rows - into the [layers][rows][features] operands of a strided-batched weight-gradient GEMM
// (branchgan/fused.py: _DeferredWeightGrads).  ATen's cat kernel moved those operands at ~1 TB/s (47 us a launch, 1.2 ms a
// step); this is a plain 16-byte-per-lane streaming copy.
#include <vector>

#include "common.h"

namespace bg {

constexpr int STACK_BATCH = 48;
constexpr int STACK_VEC_PER_BLOCK = 256 * 8;      // 32 KB per workgroup
struct StackItem {
  const char* src;
  char* dst;
  int64_t vecs;             // rows * vecs_per_row
  int64_t src_row_stride;   // bytes
  int vecs_per_row, first_block;
};
struct StackBatch {
  StackItem d[STACK_BATCH];
  int count;
};

__global__ __launch_bounds__(256) void stack_rows_kernel(StackBatch b) {
  int lo = 0, hi = b.count - 1;          // the last item whose first_block <= blockIdx.x (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.d[mid].first_block <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const StackItem& it = b.d[lo];
  const int64_t v0 = (int64_t)((int)blockIdx.x - it.first_block) * STACK_VEC_PER_BLOCK + threadIdx.x;
  f32x4 r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t v = v0 + k * 256;
    if (v < it.vecs) {
      const int64_t row = v / it.vecs_per_row, col = v % it.vecs_per_row;
      r[k] = __builtin_nontemporal_load((const f32x4*)(it.src + row * it.src_row_stride + col * 16));
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t v = v0 + k * 256;
    if (v < it.vecs) *(f32x4*)(it.dst + v * 16) = r[k];
  }
}

}  // namespace bg

using namespace bg;

extern "C" int bg_stack_rows(const bg_stack_desc* descs, int count, void* stream) {
  BG_REQUIRE(count >= 0 && (descs || count == 0), "bg_stack_rows: count=%d", count);
  for (int i = 0; i < count; ++i) {
    const bg_stack_desc& d = descs[i];
    BG_REQUIRE(d.src && d.dst && d.rows >= 0 && d.row_bytes > 0 && d.src_row_stride >= d.row_bytes, "bg_stack_rows: descriptor %d: rows=%lld row_bytes=%lld stride=%lld",
               i, (long long)d.rows, (long long)d.row_bytes, (long long)d.src_row_stride);
    BG_REQUIRE(d.row_bytes % 16 == 0 && d.src_row_stride % 16 == 0 && ((uintptr_t)d.src | (uintptr_t)d.dst) % 16 == 0,
               "bg_stack_rows: descriptor %d: rows, strides and pointers must be multiples of 16 bytes", i);
    BG_REQUIRE(d.row_bytes / 16 < (1ll << 31), "bg_stack_rows: descriptor %d: row too long", i);
  }
  for (int base = 0; base < count; base += STACK_BATCH) {
    StackBatch b;
    b.count = 0;
    int64_t blocks = 0;
    for (int i = base; i < count && i < base + STACK_BATCH; ++i) {
      const bg_stack_desc& d = descs[i];
      if (d.rows == 0) continue;
      StackItem& it = b.d[b.count++];
      it.src = (const char*)d.src, it.dst = (char*)d.dst;
      it.vecs_per_row = (int)(d.row_bytes / 16);
      it.vecs = d.rows * it.vecs_per_row;
      it.src_row_stride = d.src_row_stride;
      it.first_block = (int)blocks;
      blocks += cdiv(it.vecs, (int64_t)STACK_VEC_PER_BLOCK);
      BG_REQUIRE(blocks < (1ll << 31), "bg_stack_rows: too many blocks");
    }
    if (b.count == 0) continue;
    hipLaunchKernelGGL(stack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
  }
  BG_CHECK_LAUNCH("bg_stack_rows");
  return BG_OK;
}
