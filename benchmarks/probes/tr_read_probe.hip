// Checks the operand mapping of ds_read_b64_tr_b16 (gfx950) that attn_common.h's mma_reduce_rows relies on:
// a row-major LDS tile z[32 rows][72] read "by columns": lane (r = lane & 31, h = lane >> 5) must receive
//   z[16 s + 4 h + j][d0 + r]  (j = 0..3)   and   z[16 s + 8 + 4 h + j][d0 + r]
// from two transposed reads.  Build: hipcc -O3 --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define RS 72
__global__ void probe(const short* in, short* out, int s, int d0) {
  __shared__ short z[32 * RS];
  for (int i = threadIdx.x; i < 32 * RS; i += 64) z[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, h = lane >> 5, li = lane & 15, q = li >> 2, p = li & 3, half = (lane >> 4) & 1;
  const short* a0 = z + (16 * s + 4 * h + q) * RS + d0 + 16 * half + 4 * p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * RS));
  for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = lo[j]; out[lane * 8 + 4 + j] = hi[j]; }
}
int main() {
  std::vector<short> h(32 * RS);
  for (int r = 0; r < 32; ++r) for (int c = 0; c < RS; ++c) h[r * RS + c] = (short)(r * 100 + c);
  short *din, *dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 64 * 8 * 2);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  int bad = 0;
  for (int s = 0; s < 2; ++s) for (int d0 = 0; d0 < 64; d0 += 32) {
    probe<<<1, 64>>>(din, dout, s, d0);
    std::vector<short> o(64 * 8);
    hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
      const int r = lane & 31, hh = lane >> 5;
      const int row = 16 * s + (j < 4 ? 0 : 8) + 4 * hh + (j & 3);
      const short want = (short)(row * 100 + d0 + r);
      if (o[lane * 8 + j] != want) { if (bad < 8) printf("s=%d d0=%d lane=%d j=%d got %d want %d\n", s, d0, lane, j, o[lane * 8 + j], want); ++bad; }
    }
  }
  printf(bad ? "FAIL %d\n" : "PASS\n", bad);
  return bad != 0;
}
