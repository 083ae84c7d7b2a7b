// Token + position embedding of the GPT-2 family: gather-and-add forward, deterministic scatter-add backward.
//
// The reference's model computes `wte(input_ids) + wpe(position_ids)` (Models/GPT2BranchAttention/NonResidualDiscriminatorBase.py:
// 300-330, GPT2Model.forward) through two ATen gathers and an add, and autograd answers with two embedding_dense_backward calls
// (~120 us each at 3072 rows x 768 on MI355X) plus the AccumulateGrad adds.  Here:
//   forward   one launch: out[r] = wte[ids[r]] + wpe[pos[r % pos_rows]]                       (fp32, HBM-bound)
//   backward  two launches per table: (1) `first[r]` = the lowest row with the same index (an LDS scan of the index list);
//             (2) the workgroups of every lowest row (one per 256 columns) gather all rows that share its index, in ascending
//             row order split over their four waves and combined in a fixed order, and add the sum into the fp32 gradient of
//             that table row.
//             No atomics: the result does not depend on scheduling.  Rows are processed in chunks of EMBED_CHUNK.
// See include/branchgan.h: bg_embed_fwd, bg_embed_bwd.
#include "common.h"

namespace bg {

#define EMBED_CHUNK 8192      // rows per backward pass: the index list of one chunk lives in LDS (32 KB)
#define EMBED_THREADS 256

__global__ __launch_bounds__(EMBED_THREADS) void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos,
                                                                   int64_t pos_rows, const float* __restrict__ wte, int64_t vocab,
                                                                   const float* __restrict__ wpe, int64_t n_pos, int C,
                                                                   float* __restrict__ out) {
  const int64_t r = blockIdx.x;
  const int c = 4 * (blockIdx.y * EMBED_THREADS + threadIdx.x);
  if (c >= C) return;
  const int64_t t = ids[r];
  const int64_t p = pos ? pos[r % pos_rows] : 0;
  f32x4 v;
  if (t < 0 || t >= vocab || (pos && (p < 0 || p >= n_pos))) {
    const float nan = __builtin_nanf("");       // the reference raises an IndexError; a kernel cannot: poison the row
    v = f32x4{nan, nan, nan, nan};
  } else {
    v = *(const f32x4*)(wte + t * C + c);
    if (pos) v += *(const f32x4*)(wpe + p * C + c);
  }
  __builtin_nontemporal_store(v, (f32x4*)(out + r * C + c));
}

// first[r] = min { j <= r : idx[j] == idx[r] } within the chunk [0, n); a row whose index is outside the table matches nothing.
// One workgroup per 64 rows (row on the lane); its four waves scan a quarter of [0, r] each - every lane of a wave reads the
// same LDS words (broadcast), four indices per read - and the quarters' minima are combined through LDS.
#define EMBED_FIRST_ROWS 64
__global__ __launch_bounds__(EMBED_THREADS) void embed_first_kernel(const int64_t* __restrict__ idx, int64_t base, int64_t idx_rows, int n,
                                                                     int64_t table_rows, int* __restrict__ first) {
  __shared__ __align__(16) int s_idx[EMBED_CHUNK];
  __shared__ int s_min[4][EMBED_FIRST_ROWS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * EMBED_FIRST_ROWS;
  const int r_last = min(n - 1, r0 + EMBED_FIRST_ROWS - 1);          // only indices up to the last row of this workgroup matter
  // four index loads in flight per thread (guarded one by one, each was a round trip of its own: twelve in a row at 3072 rows);
  // the row of the index list is a 32-bit remainder when the list is short enough for that (it is for position ids)
  const int lim = (r_last + 4) & ~3;
  const bool narrow = idx_rows < (1ll << 31) - EMBED_CHUNK;
  const unsigned base_mod = narrow ? (unsigned)(base % idx_rows) : 0u;
  auto index_row = [&](int j) -> int64_t {
    if (narrow) return (int64_t)((base_mod + (unsigned)j) % (unsigned)idx_rows);
    return (base + j) % idx_rows;
  };
  for (int j0 = threadIdx.x; j0 < lim; j0 += 4 * EMBED_THREADS) {
    int64_t t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = idx[index_row(min(j0 + u * EMBED_THREADS, r_last))];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * EMBED_THREADS;
      if (j < lim) s_idx[j] = j > r_last ? -1 : (t[u] >= 0 && t[u] < table_rows) ? (int)t[u] : -(j + 3);
    }
  }
  __syncthreads();
  const int r = r0 + lane;
  const int mine = r < n ? s_idx[r] : -2;
  const int quarter = (((r_last + 4) >> 2) + 3) >> 2 << 2;            // int4 reads per wave, rounded up: 4 quarters cover [0, r_last]
  const int j_begin = wave * quarter, j_end = min(j_begin + quarter, (r_last + 4) & ~3);
  int f = r;
  auto fold = [&](const int4 v, int j) __attribute__((always_inline)) {
    if (v.w == mine && j + 3 < f) f = j + 3;
    if (v.z == mine && j + 2 < f) f = j + 2;
    if (v.y == mine && j + 1 < f) f = j + 1;
    if (v.x == mine && j < f) f = j;
  };
  // four LDS reads in flight per lane (one read per iteration left the scan waiting out the LDS latency 192 times at 3072 rows);
  // f is a minimum over matches, so the order of the folds does not matter
  int j = j_begin;
  for (; j + 16 <= j_end; j += 16) {
    const int4 v0 = *(const int4*)&s_idx[j], v1 = *(const int4*)&s_idx[j + 4], v2 = *(const int4*)&s_idx[j + 8], v3 = *(const int4*)&s_idx[j + 12];
    fold(v0, j);
    fold(v1, j + 4);
    fold(v2, j + 8);
    fold(v3, j + 12);
  }
  for (; j < j_end; j += 4) fold(*(const int4*)&s_idx[j], j);
  s_min[wave][lane] = f;
  __syncthreads();
  if (wave == 0 && r < n) first[r] = min(min(s_min[0][lane], s_min[1][lane]), min(s_min[2][lane], s_min[3][lane]));
}

#define EMBED_U 16            // rows in flight per wave
#define EMBED_SLOTS (EMBED_CHUNK / EMBED_THREADS)

// Workgroup (r, y): row r of the chunk, columns [256 y, 256 y + 256); only the lowest row of each index works.
__global__ __launch_bounds__(EMBED_THREADS) void embed_accum_kernel(const float* __restrict__ dy, const int64_t* __restrict__ idx, int64_t base,
                                                                     int64_t idx_rows, const int* __restrict__ first, int n, int C,
                                                                     int64_t table_rows, float* __restrict__ dW) {
  const int r = blockIdx.x;
  if (first[r] != r) return;
  const int64_t t = idx[(base + r) % idx_rows];
  if (t < 0 || t >= table_rows) return;                 // the forward poisoned these rows; there is nothing to add them to
  __shared__ int s_list[EMBED_CHUNK];
  __shared__ int s_total;
  __shared__ __align__(16) float s_part[3][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // which of the rows j >= r share the index: all loads in flight at once, one bit per slot
  const int j_begin = r & ~(EMBED_THREADS - 1);
  unsigned hits = 0;
  {
    int f[EMBED_SLOTS];                                  // (the address is clamped, not the load guarded: a guarded load is waited for
#pragma unroll                                           // on the spot, and there are 32 of them)
    for (int k = 0; k < EMBED_SLOTS; ++k) f[k] = first[min(max(j_begin + k * EMBED_THREADS + (int)threadIdx.x, r), n - 1)];
#pragma unroll
    for (int k = 0; k < EMBED_SLOTS; ++k) {
      const int j = j_begin + k * EMBED_THREADS + threadIdx.x;
      if (j >= r && j < n && f[k] == r) hits |= 1u << k;
    }
  }
  const int c = 256 * blockIdx.y + 4 * lane;
  const bool live = c < C;
  const float* src = dy + base * (int64_t)C + c;
  // most indices occur once (the tail of a token distribution): nothing to gather
  if (__syncthreads_count(hits != 0) == 1 && __syncthreads_count(__popc(hits) > 1) == 0) {
    if (wave == 0 && live) {
      f32x4* dst = (f32x4*)(dW + t * C + c);
      *dst += __builtin_nontemporal_load((const f32x4*)(src + (int64_t)r * C));
    }
    return;
  }
  // ordered compaction into s_list: the hits of slot k, wave w are counted by a ballot (no barrier), the counts meet in LDS,
  // and every wave adds up what lies before each of its slots - two barriers, where a running total per slot took three each
  __shared__ int s_count[EMBED_SLOTS][4];
  static_assert(EMBED_SLOTS <= 64, "one lane per slot");
  {
    int count_k = 0;                                     // lane k keeps this wave's count of slot k (0 for slots past the chunk's end)
#pragma unroll
    for (int k = 0; k < EMBED_SLOTS; ++k) {
      const int c = __popcll(__ballot((hits >> k) & 1u));
      if (lane == k) count_k = c;
    }
    if (lane < EMBED_SLOTS) s_count[lane][wave] = count_k;
  }
  __syncthreads();
  {
    // lane k of every wave: entries before (slot k, this wave) = all waves of the slots below k + the lower waves of slot k
    int mine_k = 0, row_k = 0;
    if (lane < EMBED_SLOTS) {
      for (int w = 0; w < 4; ++w) { const int c = s_count[lane][w]; row_k += c; if (w < wave) mine_k += c; }
    }
    int below = row_k;                                   // exclusive prefix of the slot totals over the lanes
#pragma unroll
    for (int d = 1; d < EMBED_SLOTS; d <<= 1) { const int o = __shfl_up(below, d); if (lane >= d) below += o; }
    const int total_all = __shfl(below, EMBED_SLOTS - 1);
    below -= row_k;
    const int start = below + mine_k;
#pragma unroll
    for (int k = 0; k < EMBED_SLOTS; ++k) {
      const int off = __shfl(start, k);
      const bool hit = (hits >> k) & 1u;
      const uint64_t m = __ballot(hit);
      if (hit) s_list[off + __popcll(m & ((1ull << lane) - 1))] = j_begin + k * EMBED_THREADS + threadIdx.x;
    }
    if (threadIdx.x == 0) s_total = total_all;
  }
  __syncthreads();
  const int total = s_total;
  // wave w sums list entries w, w + 4, ... in ascending order; its lanes span 256 columns
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int e = wave; e < total; e += 4 * EMBED_U) {
    f32x4 x[EMBED_U];
#pragma unroll
    for (int u = 0; u < EMBED_U; ++u) {
      x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live && e + 4 * u < total) x[u] = __builtin_nontemporal_load((const f32x4*)(src + (int64_t)s_list[e + 4 * u] * C));
    }
#pragma unroll
    for (int u = 0; u < EMBED_U; ++u) acc += x[u];
  }
  // the four waves combine in a fixed order
  if (wave > 0) *(f32x4*)&s_part[wave - 1][4 * lane] = acc;
  __syncthreads();
  if (wave == 0 && live) {
    for (int w = 0; w < 3; ++w) acc += *(const f32x4*)&s_part[w][4 * lane];
    f32x4* dst = (f32x4*)(dW + t * C + c);
    *dst += acc;
  }
}

}  // namespace bg

extern "C" int bg_embed_fwd(const int64_t* ids, int64_t rows, const int64_t* pos, int64_t pos_rows, const float* wte, int64_t vocab,
                            const float* wpe, int64_t n_pos, int C, float* out, void* stream) {
  using namespace bg;
  BG_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0, "bg_embed_fwd: rows %lld, C %d (C %% 4 == 0)", (long long)rows, C);
  if (rows == 0) return BG_OK;
  BG_REQUIRE(ids && wte && out && vocab > 0, "bg_embed_fwd: null argument");
  BG_REQUIRE(!pos || (wpe && pos_rows > 0 && n_pos > 0), "bg_embed_fwd: positions without a table");
  BG_REQUIRE(rows < (1ll << 31), "bg_embed_fwd: rows %lld", (long long)rows);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)rows, (unsigned)cdiv(C, 4 * EMBED_THREADS)), dim3(EMBED_THREADS), 0, (hipStream_t)stream,
                     ids, pos, pos ? pos_rows : 1, wte, vocab, wpe, n_pos, C, out);
  BG_CHECK_LAUNCH("bg_embed_fwd");
  return BG_OK;
}

extern "C" size_t bg_embed_bwd_workspace_bytes(int64_t rows) {
  return rows <= 0 ? 0 : (size_t)(rows < EMBED_CHUNK ? rows : EMBED_CHUNK) * sizeof(int);
}

extern "C" int bg_embed_bwd(const float* dy, const int64_t* idx, int64_t rows, int64_t idx_rows, int64_t table_rows, int C, float* dW,
                            void* workspace, size_t workspace_bytes, void* stream) {
  using namespace bg;
  BG_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && C <= 65535 * 256, "bg_embed_bwd: rows %lld, C %d (C %% 4 == 0)", (long long)rows, C);
  if (rows == 0) return BG_OK;
  BG_REQUIRE(dy && idx && dW && idx_rows > 0 && table_rows > 0 && table_rows < (1ll << 31), "bg_embed_bwd: null / out-of-range argument");
  BG_REQUIRE(workspace && workspace_bytes >= bg_embed_bwd_workspace_bytes(rows), "bg_embed_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int* first = (int*)workspace;
  for (int64_t base = 0; base < rows; base += EMBED_CHUNK) {       // chunks run back to back on the stream: dW accumulates
    const int n = (int)(rows - base < EMBED_CHUNK ? rows - base : EMBED_CHUNK);
    hipLaunchKernelGGL(embed_first_kernel, dim3((unsigned)cdiv(n, EMBED_FIRST_ROWS)), dim3(EMBED_THREADS), 0, st, idx, base, idx_rows, n, table_rows,
                       first);
    BG_CHECK_LAUNCH("bg_embed_bwd(first)");
    hipLaunchKernelGGL(embed_accum_kernel, dim3((unsigned)n, (unsigned)cdiv(C, 256)), dim3(EMBED_THREADS), 0, st, dy, idx, base, idx_rows, first, n, C,
                       table_rows, dW);
    BG_CHECK_LAUNCH("bg_embed_bwd(accumulate)");
  }
  return BG_OK;
}
