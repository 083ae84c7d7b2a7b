#!/usr/bin/env python3
"""Bandwidth of the row-wise glue kernels (csrc/layer.hip) at the GPT-2-small shapes of BASELINE configs[1]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
from branchgan import fused  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
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
    bf = torch.bfloat16
    for rows in (256, 1024, 2048, 3072):
        C = 768
        x = torch.randn(rows, C, device="cuda")
        delta = torch.randn(rows, C, device="cuda").to(bf)
        ln = torch.nn.LayerNorm(C).cuda()
        ln.weight.grad, ln.bias.grad = torch.zeros_like(ln.weight), torch.zeros_like(ln.bias)
        bias = torch.nn.Parameter(torch.zeros(C, device="cuda"))
        bias.grad = torch.zeros_like(bias)
        t = timeit(lambda: fused._ln_fwd(x, delta, ln.weight.detach(), ln.bias.detach(), 1e-5, bf))
        nb = rows * C * (4 + 2 + 4 + 2)
        print(f"add_layernorm_fwd   {rows:5d}x{C}: {t:6.1f} us  {nb / t / 1e6:5.2f} TB/s")
        x_out, y, mean, rstd = fused._ln_fwd(x, delta, ln.weight.detach(), ln.bias.detach(), 1e-5, bf)
        dy = torch.randn(rows, C, device="cuda").to(bf)
        dres = torch.randn(rows, C, device="cuda")
        for cs in (None, bias):
            t = timeit(lambda: fused._ln_bwd(dy, x_out, mean, rstd, ln.weight, ln.bias, dres, bf, cs))
            nb = rows * C * (2 + 4 + 4 + 4 + 2)
            print(f"layernorm_bwd{'+colsum' if cs is not None else '       '} {rows:5d}x{C}: {t:6.1f} us  {nb / t / 1e6:5.2f} TB/s  (incl. partial reduce)")
        N = 3072
        w = torch.nn.Parameter(torch.randn(C, N, device="cuda") * 0.02)
        b = torch.nn.Parameter(torch.zeros(N, device="cuda"))
        w.grad, b.grad = torch.zeros_like(w), torch.zeros_like(b)
        pre = torch.randn(rows, N, device="cuda").to(bf)
        dyn = torch.randn(rows, N, device="cuda").to(bf)
        from branchgan import _capi as Cc
        lib = Cc.lib()
        ws = fused._workspace(lib.bg_bias_act_bwd_workspace_bytes(rows, N), x.device)
        dpre = torch.empty_like(pre)
        for act, name, nbytes in ((1, "gelu", 3 * rows * N * 2), (0, "none", rows * N * 2)):
            t = timeit(lambda: Cc.check(lib.bg_bias_act_bwd(Cc.ptr(dyn), Cc.dtype_code(dyn), rows, N, dyn.stride(0), act, Cc.ptr(pre),
                                                            Cc.ptr(dpre), Cc.ptr(b.grad), Cc.ptr(ws), ws.numel(), Cc.stream_ptr()), "x"))
            print(f"bias_act_bwd[{name}]  {rows:5d}x{N}: {t:6.1f} us  {nbytes / t / 1e6:5.2f} TB/s  (incl. partial reduce)")
        g = timeit(lambda: torch.nn.functional.gelu(pre, approximate="tanh"))
        print(f"torch gelu fwd      {rows:5d}x{N}: {g:6.1f} us  {2 * rows * N * 2 / g / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
