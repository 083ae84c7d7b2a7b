#!/usr/bin/env python3
"""hipBLASLt timings of the GEMM shapes one configs[1] train step issues (graph-replay timing, bf16), to see how far the
small-M GEMMs of the branch passes are from the MFMA roof."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

from benchmarks.kernel_sweep import timeit_graphed

E, F3, FF, V = 768, 2304, 3072, 50257
dev = "cuda"
bf = torch.bfloat16
print(f"{'op':34s} {'M':>5s} {'us':>8s} {'TFLOP/s':>8s}")
for M in (256, 768, 2048, 3072):
    x = {n: torch.randn(M, n, device=dev, dtype=bf) for n in (E, F3, FF)}
    w = {(a, b): torch.randn(a, b, device=dev, dtype=bf) for a, b in ((E, F3), (E, E), (E, FF), (FF, E))}
    bias = {n: torch.randn(n, device=dev, dtype=bf) for n in (E, F3, FF)}
    g32 = {k: torch.zeros(k, device=dev) for k in w}
    cases = []
    for (a, b) in w:
        cases.append((f"fwd  x[M,{a}] @ W[{a},{b}] + b", lambda a=a, b=b: torch.addmm(bias[b], x[a], w[(a, b)]), 2 * M * a * b))
        cases.append((f"dX   dy[M,{b}] @ W[{a},{b}]^T", lambda a=a, b=b: torch.mm(x[b], w[(a, b)].t()), 2 * M * a * b))
        cases.append((f"dW   x[M,{a}]^T @ dy[M,{b}] -> f32 +=", lambda a=a, b=b: torch.addmm(g32[(a, b)], x[a].t(), x[b], out_dtype=torch.float32, out=g32[(a, b)]), 2 * M * a * b))
    for name, fn, flops in cases:
        ms = timeit_graphed(lambda fn=fn: fn)
        print(f"{name:34s} {M:5d} {ms * 1e3:8.1f} {flops / ms / 1e9:8.1f}")
xh = torch.randn(2048, E, device=dev, dtype=bf)
wv = torch.randn(V, E, device=dev, dtype=bf)
for M in (256, 768, 2048):
    ms = timeit_graphed(lambda M=M: (lambda: torch.mm(xh[:M], wv.t())))
    print(f"{'LM head x[M,768] @ W[V,768]^T':34s} {M:5d} {ms * 1e3:8.1f} {2 * M * E * V / ms / 1e9:8.1f}")
