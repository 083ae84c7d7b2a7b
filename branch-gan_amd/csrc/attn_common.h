// Shared pieces of the branch-attention kernels (forward, dQ, dK/dV).
//
// MFMA shapes (gfx950): bf16 inputs use v_mfma_f32_32x32x16_bf16, fp32 inputs use v_mfma_f32_32x32x2_f32
// (exact fp32 FMA chain, so the fp32 path matches the reference's fp32 matmuls to rounding order).
// Both give a 32x32 fp32 tile whose COLUMN is the lane (lane & 31) and whose 16 registers are the rows
//   row(reg, half) = (reg & 3) + 8 * (reg >> 2) + 4 * half,      half = lane >> 5.
// Operand maps:   bf16: A[row = lane&31][k = 8*half + j], B[k = 8*half + j][col = lane&31]   (j = 0..7)
//                 fp32: A[row = lane&31][k = half],         B[k = half][col = lane&31]
// A score tile is reused directly as the B operand of the next product (no LDS round trip): as bf16,
// registers 8s..8s+7 form k-step s and element j of half h is row 16s + 8(j>>2) + 4h + (j&3); as fp32,
// register r is one k-step whose half h is row (r&3) + 8(r>>2) + 4h.  The other operand is gathered
// from LDS in exactly that row order.
#pragma once
#include <type_traits>

#include "common.h"

namespace bg {

constexpr int ATT_WAVES = 4;   // waves per workgroup
constexpr int ATT_TILE = 32;   // rows (queries or keys) per wave tile, keys per LDS tile

struct AttnArgs {
  bg_attn_shape sh;
  const void* q;
  const void* k_text;
  const void* v_text;
  const void* k_slots[BG_MAX_SLOTS];
  const void* v_slots[BG_MAX_SLOTS];
  const int32_t* ctx_len;
  const int32_t* path;
  const int32_t* path_len;
  void* out;        // fwd: out;  bwd: forward output (for delta)
  float* lse;
  // backward only
  const void* dout;
  void* dq;
  void* dk_text;
  void* dv_text;
  const void* dk_text_in;   // optional: added to dK / dV of the text rows before the store (may alias dk_text / dv_text)
  const void* dv_text_in;
  void* dk_slots[BG_MAX_SLOTS];
  void* dv_slots[BG_MAX_SLOTS];
  float* delta;
  // forward, key range of a query tile split over several workgroups (attn_fwd.hip): number of splits and the partial states
  int kv_splits;
  float* part;
};

template <typename K>
static inline void set_lds_limit(K kernel, size_t lds) {
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  The kernels below index their work as
// (tile = x, batch*head = y) with x fastest, so the tiles of one head - which read the SAME K/V - would land on different
// XCDs and each L2 would fetch its own copy.  This remap gives every XCD a contiguous run of (head, tile) pairs instead.
struct BlockXY {
  int x, y;
};
__device__ __forceinline__ BlockXY xcd_block() {
  const unsigned gx = gridDim.x, total = gx * gridDim.y;
  unsigned lin = blockIdx.y * gx + blockIdx.x;
  if (total % 8 == 0) lin = (lin % 8) * (total / 8) + lin / 8;
  return BlockXY{(int)(lin % gx), (int)(lin / gx)};
}

__device__ __forceinline__ int tile_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// LDS layouts ---------------------------------------------------------------------------------------
// "row" layout  R[rows][DH + RPAD]: A operand of a product that reduces over d (scores).
// "col" layout  bf16: C[DH][32 + 4] (transposed at staging);  fp32: the padded row layout serves both.
template <typename T, int DH> struct Lds;
template <int DH> struct Lds<float, DH> {
  static constexpr int RSTRIDE = DH + 1;          // odd stride: conflict-free ds_read_b32 down a column
  static constexpr int ROW_ELEMS = ATT_TILE * RSTRIDE;
  static constexpr int COL_ELEMS = 0;             // not needed
};
template <int DH> struct Lds<bf16_t, DH> {
  static constexpr int RSTRIDE = DH + 8;          // 16-byte pad: conflict-free ds_read_b128 down a column
  static constexpr int ROW_ELEMS = ATT_TILE * RSTRIDE;
  static constexpr int CSTRIDE = ATT_TILE + 4;    // 72-byte rows: 8-byte aligned key quads
  static constexpr int COL_ELEMS = DH * CSTRIDE;
};

// Cooperative staging of a 32-row tile (rows row0 .. row0+31 of a [rows][DH] global matrix with element
// stride `rs` between rows) by `nthreads` threads.  Rows >= nrows are zero-filled.
template <typename T, int DH, bool ROW, bool COL>
__device__ __forceinline__ void stage_tile(const T* g, int64_t rs, int row0, int nrows, T* lds_row, T* lds_col, int tid, int nthreads) {
  constexpr int VEC = 16 / sizeof(T);
  constexpr int CHUNKS = ATT_TILE * DH / VEC;
  for (int c = tid; c < CHUNKS; c += nthreads) {
    const int r = c / (DH / VEC), d0 = (c % (DH / VEC)) * VEC;
    T v[VEC];
    if (row0 + r < nrows) {
      if constexpr (sizeof(T) == 4) {
        f32x4 raw = *(const f32x4*)(g + (int64_t)(row0 + r) * rs + d0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = raw[j];
      } else {
        bf16x8 raw = *(const bf16x8*)(g + (int64_t)(row0 + r) * rs + d0);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = raw[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = (T)0.0f;
    }
    if constexpr (sizeof(T) == 4) {
      if constexpr (ROW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_row[r * Lds<T, DH>::RSTRIDE + d0 + j] = v[j];
      }
    } else {
      if constexpr (ROW) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = v[j];
        *(bf16x8*)(lds_row + r * Lds<T, DH>::RSTRIDE + d0) = o;
      }
      if constexpr (COL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_col[(d0 + j) * Lds<bf16_t, DH>::CSTRIDE + r] = v[j];
      }
    }
  }
}

// The same staging split in two, so that the global loads of tile i+1 are in flight while tile i is consumed:
// tile_load fills per-thread registers, tile_store writes them to LDS (row and / or transposed layout).
template <typename T, int DH, int NTHREADS> struct TileRegs {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int CHUNKS = ATT_TILE * DH / VEC;
  static constexpr int PER = (CHUNKS + NTHREADS - 1) / NTHREADS;
  typedef typename std::conditional<sizeof(T) == 4, f32x4, bf16x8>::type V;
  V v[PER];
  __device__ __forceinline__ void load(const T* g, int64_t rs, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NTHREADS;
      const int r = c / (DH / VEC), d0 = (c % (DH / VEC)) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[i][j] = (T)0.0f;
      if (c < CHUNKS && row0 + r < nrows) v[i] = *(const V*)(g + (int64_t)(row0 + r) * rs + d0);
    }
  }
  // the same for rows that are (n, k)-indexed (queries, outputs and their gradients): row rr = row0 + r lives at
  // g + (rr / K) * sn + (rr % K) * sk
  __device__ __forceinline__ void load_nk(const T* g, int64_t sn, int64_t sk, int K, int nrows, int row0, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NTHREADS;
      const int r = c / (DH / VEC), d0 = (c % (DH / VEC)) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[i][j] = (T)0.0f;
      const int rr = row0 + r;
      if (c < CHUNKS && rr < nrows) v[i] = *(const V*)(g + (int64_t)(rr / K) * sn + (int64_t)(rr % K) * sk + d0);
    }
  }
  template <bool ROW, bool COL>
  __device__ __forceinline__ void store(T* lds_row, T* lds_col, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NTHREADS;
      if (c >= CHUNKS) continue;
      const int r = c / (DH / VEC), d0 = (c % (DH / VEC)) * VEC;
      if constexpr (sizeof(T) == 4) {
        if constexpr (ROW) {
#pragma unroll
          for (int j = 0; j < 4; ++j) lds_row[r * Lds<T, DH>::RSTRIDE + d0 + j] = v[i][j];
        }
      } else {
        if constexpr (ROW) *(bf16x8*)(lds_row + r * Lds<T, DH>::RSTRIDE + d0) = v[i];
        if constexpr (COL) {
#pragma unroll
          for (int j = 0; j < 8; ++j) lds_col[(d0 + j) * Lds<bf16_t, DH>::CSTRIDE + r] = v[i][j];
        }
      }
    }
  }
};

// acc[32 x 32] += X_tile(rows r, reduce over d) * Y^T  where X comes from LDS "row" layout (A operand,
// tile row on the lane) and Y's fragment for this lane's column is supplied by `yfrag(t)`.
// bf16: yfrag(t) -> bf16x8 = Y[col][16t + 8h .. +8];  fp32: yfrag(t) -> float = Y[col][2t + h].
template <typename T, int DH, typename YF>
__device__ __forceinline__ void mma_reduce_d(f32x16& acc, const T* lds_row, int lane, YF yfrag) {
  const int r = lane & 31, h = lane >> 5;
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int t = 0; t < DH / 2; ++t) {
      const float a = lds_row[r * Lds<T, DH>::RSTRIDE + 2 * t + h];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, yfrag(t), acc, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int t = 0; t < DH / 16; ++t) {
      const bf16x8 a = *(const bf16x8*)(lds_row + r * Lds<T, DH>::RSTRIDE + 16 * t + 8 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, yfrag(t), acc, 0, 0, 0);
    }
  }
}

// out[dt][32 d x 32 cols] += Z^T (d x tile rows) * W (tile rows x cols), W = a score-shaped tile held in
// registers `w` (column on the lane), Z = the LDS tile (bf16: "col" layout, fp32: padded "row" layout).
template <typename T, int DH, int DT0, int DT1>
__device__ __forceinline__ void mma_reduce_rows(f32x16 (&out)[DH / 32], const f32x16& w, const T* lds_z, int lane) {
  const int r = lane & 31, h = lane >> 5;
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = tile_row(reg, h);
#pragma unroll
      for (int dt = DT0; dt < DT1; ++dt) {
        const float a = lds_z[row * Lds<T, DH>::RSTRIDE + dt * 32 + r];
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[reg], out[dt], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 b;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = (bf16_t)w[8 * s + j];
#pragma unroll
      for (int dt = DT0; dt < DT1; ++dt) {
        const bf16_t* zr = lds_z + (dt * 32 + r) * Lds<bf16_t, DH>::CSTRIDE + 16 * s + 4 * h;
        const bf16x4 lo = *(const bf16x4*)(zr);
        const bf16x4 hi = *(const bf16x4*)(zr + 8);
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
        out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, out[dt], 0, 0, 0);
      }
    }
  }
}

// The same product with Z read from its ROW image (bf16, [tile row][RSTRIDE]) by gfx950's transposed LDS read: per 16-lane group
// `ds_read_b64_tr_b16` takes a block of 4 rows x 16 columns and hands lane i column i of the four rows - exactly the A
// operand above (d = dt * 32 + r on the lane, tile rows 16 s + 4 h + {0..3} and + 8) without a second, column-major copy of the
// tile and its 2-byte scattered stores.  Lane 4 q + p of a group supplies the address of row q, columns 4 p .. 4 p + 3
// (mapping checked on the device by benchmarks/probes/tr_read_probe.hip).  Needs all 64 lanes active.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
template <int DH, int DT0, int DT1>
__device__ __forceinline__ void mma_reduce_rows_tr(f32x16 (&out)[DH / 32], const f32x16& w, const bf16_t* lds_row, int lane) {
  const int h = lane >> 5, li = lane & 15, q = li >> 2, p = li & 3, half = (lane >> 4) & 1;
  const bf16_t* base = lds_row + (4 * h + q) * Lds<bf16_t, DH>::RSTRIDE + 16 * half + 4 * p;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (bf16_t)w[8 * s + j];
#pragma unroll
    for (int dt = DT0; dt < DT1; ++dt) {
      const bf16_t* zr = base + 16 * s * Lds<bf16_t, DH>::RSTRIDE + dt * 32;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)zr);
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(zr + 8 * Lds<bf16_t, DH>::RSTRIDE));
      union { s16x4_t s4[2]; bf16x8 v; } a;
      a.s4[0] = lo; a.s4[1] = hi;
      out[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b, out[dt], 0, 0, 0);
    }
  }
}

// Per-lane fragments of this lane's own row (query / key) for the B operand of mma_reduce_d.
template <typename T, int DH> struct RowFrag;
template <int DH> struct RowFrag<float, DH> {
  static constexpr bool IN_REGS = (DH <= 64);
  float v[IN_REGS ? DH / 2 : 1];
  const float* p;
  __device__ __forceinline__ void load(const float* row, bool valid, int h) {
    p = valid ? row : nullptr;
    if constexpr (IN_REGS) {
#pragma unroll
      for (int t = 0; t < DH / 2; ++t) v[t] = valid ? row[2 * t + h] : 0.f;
    }
  }
  __device__ __forceinline__ float get(int t, int h) const {
    if constexpr (IN_REGS) return v[t];
    else return p ? p[2 * t + h] : 0.f;
  }
};
template <int DH> struct RowFrag<bf16_t, DH> {
  static constexpr bool IN_REGS = (DH <= 256);      // Dh 256: 64 registers per fragment, at one wave per SIMD (512 registers a lane) cheaper than a global re-read per tile
  bf16x8 v[IN_REGS ? DH / 16 : 1];
  const bf16_t* p;
  __device__ __forceinline__ void load(const bf16_t* row, bool valid, int h) {
    p = valid ? row : nullptr;
    if constexpr (IN_REGS) {
#pragma unroll
      for (int t = 0; t < DH / 16; ++t) {
        if (valid) v[t] = *(const bf16x8*)(row + 16 * t + 8 * h);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[t][j] = (bf16_t)0.0f;
        }
      }
    }
  }
  __device__ __forceinline__ bf16x8 get(int t, int h) const {
    if constexpr (IN_REGS) return v[t];
    else {
      bf16x8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.0f;
      return p ? *(const bf16x8*)(p + 16 * t + 8 * h) : z;
    }
  }
};

// The same fragments for a row that sits in an LDS tile (row layout) for the kernel's whole life: nothing is kept, get()
// reads where the product needs it.
template <typename T, int DH> struct LdsRowFrag {
  const T* p;
  __device__ __forceinline__ void load(const T* row) { p = row; }
  __device__ __forceinline__ auto get(int t, int h) const {
    if constexpr (sizeof(T) == 4) return p[2 * t + h];
    else return *(const bf16x8*)(p + 16 * t + 8 * h);
  }
};

// Dot product of two DH-vectors split over a lane pair (l, l^32): half h covers [h*DH/2, (h+1)*DH/2).
template <typename T, int DH>
__device__ __forceinline__ float pair_dot(const T* a, const T* b, int h, bool valid) {
  float acc = 0.f;
  if (valid) {
    constexpr int VEC = 16 / sizeof(T);
    const T* pa = a + h * (DH / 2);
    const T* pb = b + h * (DH / 2);
#pragma unroll
    for (int c = 0; c < DH / 2 / VEC; ++c) {
      if constexpr (sizeof(T) == 4) {
        f32x4 x = *(const f32x4*)(pa + 4 * c), y = *(const f32x4*)(pb + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(x[j], y[j], acc);
      } else {
        bf16x8 x = *(const bf16x8*)(pa + 8 * c), y = *(const bf16x8*)(pb + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf((float)x[j], (float)y[j], acc);
      }
    }
  }
  return acc + __shfl_xor(acc, 32);
}

// This lane's key / value rows of slot `slot` (nullptr for lanes that are not `act`, or whose slot is out of range).  The
// pointer tables are kernel arguments: they are read with SCALAR loads, once per distinct slot among the wave's active lanes
// (one, when a query tile lies inside a node) - indexed per lane they were a vector load from the argument buffer, one more
// dependent round trip between the path entry and the key row.
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const void*)(((uint64_t)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ void slot_rows(const AttnArgs& a, int slot, bool act, int64_t off, const T*& kr, const T*& vr) {
  kr = nullptr;
  vr = nullptr;
  unsigned long long todo = __ballot(act && slot >= 0 && slot < a.sh.n_slots);
  while (todo) {
    const int s = __builtin_amdgcn_readlane(slot, __builtin_ctzll(todo));
    const T* kp = (const T*)uniform_ptr(a.k_slots[s]);   // (kept out of the branch below, where the index would be per lane again)
    const T* vp = (const T*)uniform_ptr(a.v_slots[s]);
    const bool mine = act && slot == s;
    if (mine) { kr = kp + off; vr = vp + off; }
    todo &= ~__ballot(mine);
  }
}

// A branch key and its value (this lane's own rows of a slot) requested TOGETHER and held in registers: k in pair_dot's layout
// (this half-wave's DH / 2 features, 16-byte pieces), v in axpy_rowvec's.  pair_dot on the key and axpy_rowvec on the value behind
// the softmax update were two dependent round trips per path entry; dot() and axpy() add in the order of those two functions.
template <typename T, int DH>
struct PathRow {
  static constexpr int VEC = 16 / sizeof(T);
  typedef typename std::conditional<sizeof(T) == 4, f32x4, bf16x8>::type KV;
  typedef typename std::conditional<sizeof(T) == 4, f32x4, bf16x4>::type VV;
  KV k[DH / 2 / VEC];
  VV v[DH / 8];
  __device__ __forceinline__ void load(const T* kr, const T* vr, int h, bool valid) {
#pragma unroll
    for (int c = 0; c < DH / 2 / VEC; ++c)
#pragma unroll
      for (int j = 0; j < VEC; ++j) k[c][j] = (T)0.f;
#pragma unroll
    for (int i = 0; i < DH / 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = (T)0.f;
    if (valid) {
#pragma unroll
      for (int c = 0; c < DH / 2 / VEC; ++c) k[c] = *(const KV*)(kr + h * (DH / 2) + VEC * c);
#pragma unroll
      for (int i = 0; i < DH / 8; ++i) v[i] = *(const VV*)(vr + (i >> 2) * 32 + 8 * (i & 3) + 4 * h);
    }
  }
  // a . k over the whole row (both half-waves): `a` is this lane's row, readable by every lane (LDS or global)
  __device__ __forceinline__ float dot(const T* a, int h, bool valid) const {
    float acc = 0.f;
    if (valid) {
#pragma unroll
      for (int c = 0; c < DH / 2 / VEC; ++c) {
        const KV x = *(const KV*)(a + h * (DH / 2) + VEC * c);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc = fmaf((float)x[j], (float)k[c][j], acc);
      }
    }
    return acc + __shfl_xor(acc, 32);
  }
  __device__ __forceinline__ void axpy(f32x16 (&acc)[DH / 32], float w) const {
#pragma unroll
    for (int i = 0; i < DH / 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i >> 2][4 * (i & 3) + j] = fmaf(w, (float)v[i][j], acc[i >> 2][4 * (i & 3) + j]);
  }
};

// The backward's form: key and value rows in pair_dot's layout (for q . k and dO . v) and the key row once more in
// axpy_rowvec's (dQ += dS k), all requested in one round.
template <typename T, int DH>
struct BwdPathRow {
  static constexpr int VEC = 16 / sizeof(T);
  typedef typename PathRow<T, DH>::KV KV;
  typedef typename PathRow<T, DH>::VV VV;
  KV k[DH / 2 / VEC], v[DH / 2 / VEC];
  VV k2[DH / 8];
  __device__ __forceinline__ void load(const T* kr, const T* vr, int h, bool valid) {
#pragma unroll
    for (int c = 0; c < DH / 2 / VEC; ++c)
#pragma unroll
      for (int j = 0; j < VEC; ++j) { k[c][j] = (T)0.f; v[c][j] = (T)0.f; }
#pragma unroll
    for (int i = 0; i < DH / 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) k2[i][j] = (T)0.f;
    if (valid) {
#pragma unroll
      for (int c = 0; c < DH / 2 / VEC; ++c) {
        k[c] = *(const KV*)(kr + h * (DH / 2) + VEC * c);
        v[c] = *(const KV*)(vr + h * (DH / 2) + VEC * c);
      }
#pragma unroll
      for (int i = 0; i < DH / 8; ++i) k2[i] = *(const VV*)(kr + (i >> 2) * 32 + 8 * (i & 3) + 4 * h);
    }
  }
  __device__ __forceinline__ float dot(const KV (&y)[DH / 2 / VEC], const T* a, int h, bool valid) const {
    float acc = 0.f;
    if (valid) {
#pragma unroll
      for (int c = 0; c < DH / 2 / VEC; ++c) {
        const KV x = *(const KV*)(a + h * (DH / 2) + VEC * c);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc = fmaf((float)x[j], (float)y[c][j], acc);
      }
    }
    return acc + __shfl_xor(acc, 32);
  }
  __device__ __forceinline__ float dot_k(const T* a, int h, bool valid) const { return dot(k, a, h, valid); }
  __device__ __forceinline__ float dot_v(const T* a, int h, bool valid) const { return dot(v, a, h, valid); }
  __device__ __forceinline__ void axpy_k(f32x16 (&acc)[DH / 32], float w) const {
#pragma unroll
    for (int i = 0; i < DH / 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i >> 2][4 * (i & 3) + j] = fmaf(w, (float)k2[i][j], acc[i >> 2][4 * (i & 3) + j]);
  }
};

// acc[dt][reg] += w * row[d(dt, reg, h)] : adds a scaled DH-vector (this lane's own row in global memory)
// to an accumulator set whose rows are d (the layout of out[] in mma_reduce_rows).
template <typename T, int DH>
__device__ __forceinline__ void axpy_rowvec(f32x16 (&acc)[DH / 32], float w, const T* row, int h, bool valid) {
  if (!valid) return;
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const T* p = row + dt * 32 + 8 * g + 4 * h;
      float x[4];
      if constexpr (sizeof(T) == 4) {
        f32x4 raw = *(const f32x4*)p;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = raw[j];
      } else {
        bf16x4 raw = *(const bf16x4*)p;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = (float)raw[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[dt][4 * g + j] = fmaf(w, x[j], acc[dt][4 * g + j]);
    }
  }
}

// Stores an accumulator set (rows = d, column = this lane's row) to this lane's row in global memory.
template <typename T, int DH>
__device__ __forceinline__ void store_rowvec(const f32x16 (&acc)[DH / 32], float mul, T* row, int h, bool valid) {
  if (!valid) return;
#pragma unroll
  for (int dt = 0; dt < DH / 32; ++dt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      T* p = row + dt * 32 + 8 * g + 4 * h;
      if constexpr (sizeof(T) == 4) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = acc[dt][4 * g + j] * mul;
        *(f32x4*)p = o;
      } else {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(acc[dt][4 * g + j] * mul);
        *(bf16x4*)p = o;
      }
    }
  }
}

}  // namespace bg
