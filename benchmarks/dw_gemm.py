#!/usr/bin/env python3
"""dW GEMMs of the fused path (bf16 x bf16 -> fp32, beta = 1, accumulated in place) under rocBLAS vs hipBLASLt, and the
bf16-output GEMM of the same shape for scale."""
import sys
import torch


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1000 / reps


def main():
    shapes = [(768, 2304), (768, 768), (768, 3072), (3072, 768), (768, 50257)]
    for lib in ("cublaslt",):
        torch.backends.cuda.preferred_blas_library(lib)
        for R in (1024, 2048, 3072):
            for (m, n) in shapes:
                x = torch.randn(R, m, device="cuda", dtype=torch.bfloat16)
                dy = torch.randn(R, n, device="cuda", dtype=torch.bfloat16)
                buf = torch.zeros(m, n, device="cuda")
                t_acc = timeit(lambda: torch.addmm(buf, x.t(), dy, out_dtype=torch.float32, out=buf))
                t_bf = timeit(lambda: torch.mm(x.t(), dy))
                xt = x.t().contiguous()
                t_nn = timeit(lambda: torch.addmm(buf, xt, dy, out_dtype=torch.float32, out=buf))
                dyt = dy.t().contiguous()
                t_nt = timeit(lambda: torch.addmm(buf, xt, dyt.t(), out_dtype=torch.float32, out=buf))
                t_tr = timeit(lambda: x.t().contiguous())
                fl = 2.0 * m * n * R
                print(f"{lib:9s} R={R:5d} {m:5d}x{n:<6d} TN fp32-acc {t_acc:6.1f} us {fl / t_acc / 1e6:5.0f} TF | bf16-out {t_bf:6.1f} | NN(x^T stored) {t_nn:6.1f} | NT(both stored T) {t_nt:6.1f} | transpose x {t_tr:5.1f}", flush=True)


if __name__ == "__main__":
    main()
