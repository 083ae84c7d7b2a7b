// Probe: how much faster than hipBLASLt's first heuristic choice is the best of its candidates for the weight-gradient
// GEMMs (bf16 x bf16 -> fp32 accumulate, beta = 1) of the GPT-2-small shapes?  Build:
//   hipcc -O2 --offload-arch=gfx950 benchmarks/probes/dw_gemm_tune.cpp -lhipblaslt -o benchmarks/probes/dw_gemm_tune.bin
// Row-major view: dW[in, out] += x[M, in]^T @ dy[M, out].  Column-major (hipBLASLt): D[out, in] = dy'[out, M] @ x'[in, M]^T.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAIL %s = %d (line %d)\n", #x, (int)e_, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const bool all = argc > 1 && atoi(argv[1]) != 0;
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  const size_t ws_bytes = 128u << 20;
  void* ws;
  CK(hipMalloc(&ws, ws_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int outs[4] = {2304, 768, 3072, 768}, ins[4] = {768, 768, 768, 3072};
  const int Ms[4] = {256, 768, 2048, 3072};
  void *A, *B, *C;
  CK(hipMalloc(&A, (size_t)3072 * 3072 * 2));
  CK(hipMalloc(&B, (size_t)3072 * 3072 * 2));
  CK(hipMalloc(&C, (size_t)3072 * 3072 * 4));
  CK(hipMemset(A, 0, (size_t)3072 * 3072 * 2));
  CK(hipMemset(B, 0, (size_t)3072 * 3072 * 2));
  CK(hipMemset(C, 0, (size_t)3072 * 3072 * 4));
  for (int mi = 0; mi < 4; ++mi)
    for (int s = 0; s < 4; ++s) {
      const int64_t m = outs[s], n = ins[s], k = Ms[mi];
      hipblasLtMatmulDesc_t desc;
      CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
      hipblasOperation_t opn = HIPBLAS_OP_N, opt = HIPBLAS_OP_T;
      CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
      CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opt, sizeof(opt)));
      hipblasLtMatrixLayout_t la, lb, lc;
      CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, m, k, m));
      CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, n, k, n));
      CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, m, n, m));
      hipblasLtMatmulPreference_t pref;
      CK(hipblasLtMatmulPreferenceCreate(&pref));
      CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
      std::vector<hipblasLtMatmulHeuristicResult_t> res(64);
      int found = 0;
      CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 64, res.data(), &found));
      res.resize(found);
      const float alpha = 1.f, beta = 1.f;
      if (all) {
        std::vector<hipblasLtMatmulHeuristicResult_t> every;
        CK(hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, opn, opt, HIP_R_16BF, HIP_R_16BF, HIP_R_32F, HIP_R_32F,
                                      HIPBLAS_COMPUTE_32F, every));
        for (auto& r : every) {
          size_t need = 0;
          if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &alpha, la, lb, &beta, lc, lc, r.algo, need) == HIPBLAS_STATUS_SUCCESS && need <= ws_bytes)
            res.push_back(r);
        }
      }
      float first = 0.f, best = 1e30f;
      int best_i = -1;
      for (int i = 0; i < (int)res.size(); ++i) {
        bool ok = true;
        for (int it = 0; it < 3 && ok; ++it)
          ok = hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, ws_bytes, st) == HIPBLAS_STATUS_SUCCESS;
        if (!ok) continue;
        CK(hipEventRecord(e0, st));
        const int reps = 20;
        for (int it = 0; it < reps; ++it)
          hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res[i].algo, ws, ws_bytes, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float us = ms * 1e3f / reps;
        if (i == 0) first = us;
        if (us < best) { best = us; best_i = i; }
      }
      printf("out %4d in %4d M %4d: candidates %4d (heuristic %d)  first %7.1f us  best %7.1f us (#%d, index %d)  %.2fx  best = %.0f TFLOP/s\n",
             (int)m, (int)n, (int)k, (int)res.size(), found, first, best, best_i, best_i >= 0 ? hipblaslt_ext::getIndexFromAlgo(res[best_i].algo) : -1,
             first / best, 2.0 * m * n * k / best * 1e-6);
      fflush(stdout);
      hipblasLtMatmulPreferenceDestroy(pref);
      hipblasLtMatrixLayoutDestroy(la);
      hipblasLtMatrixLayoutDestroy(lb);
      hipblasLtMatrixLayoutDestroy(lc);
      hipblasLtMatmulDescDestroy(desc);
    }
  return 0;
}
