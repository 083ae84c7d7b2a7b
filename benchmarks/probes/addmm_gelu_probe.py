#!/usr/bin/env python3
"""Is `torch._addmm_activation(use_gelu=True)` ONE hipBLASLt launch (bias + tanh-GELU epilogue) on this image, and what does a
no-grad level pass gain from it?  Chains of 48 (fc GEMM + GELU) pairs at the GPT-2 small MLP shape replayed from a HIP graph,
M = 256 (branch steps), 1024, 2048, 3072; prints us per pair for addmm + F.gelu(tanh) against _addmm_activation, and the
largest difference between the two results (the epilogue applies GELU to the fp32 accumulator, the pair to the rounded bf16)."""
import sys

import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
N_IN, N_OUT, CHAIN = 768, 3072, 48


def timed_graph(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / CHAIN


for dtype in (torch.bfloat16, torch.float32):
    for M in (256, 1024, 2048, 3072):
        x = torch.randn(M, N_IN, device=dev, dtype=dtype)
        w = (torch.randn(N_IN, N_OUT, device=dev, dtype=dtype) * 0.03)
        b = torch.randn(N_OUT, device=dev, dtype=dtype) * 0.1
        outs = [torch.empty(M, N_OUT, device=dev, dtype=dtype) for _ in range(2)]

        def pair():
            for _ in range(CHAIN):
                pre = torch.addmm(b, x, w)
                torch.ops.aten.gelu.out(pre, approximate="tanh", out=outs[0])

        def fused():
            for _ in range(CHAIN):
                torch._addmm_activation(b, x, w, use_gelu=True, out=outs[1])

        def gemm_only():
            for _ in range(CHAIN):
                torch.addmm(b, x, w, out=outs[0])

        t_pair, t_fused, t_gemm = timed_graph(pair), timed_graph(fused), timed_graph(gemm_only)
        pair(); fused()
        torch.cuda.synchronize()
        ref = torch.nn.functional.gelu(torch.addmm(b.double(), x.double(), w.double()), approximate="tanh")
        print(f"{str(dtype)[6:]:>8} M {M:5d}: addmm + gelu {t_pair:6.2f} us   _addmm_activation {t_fused:6.2f} us   addmm alone {t_gemm:6.2f} us   "
              f"max |pair - fused| {float((outs[0].float() - outs[1].float()).abs().max()):.3e}   "
              f"err vs fp64: pair {float((outs[0].double() - ref).abs().max()):.3e}  fused {float((outs[1].double() - ref).abs().max()):.3e}")
sys.stdout.flush()
