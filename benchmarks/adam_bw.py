#!/usr/bin/env python3
"""bg_adam_step bandwidth on an arena of GPT-2-small size (124.4M parameters): 34 algorithmic bytes per parameter."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "branch-gan_amd"))


def run(n=124_439_808, reps=20):
    import torch
    from branchgan import _capi as C
    p, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    m, v, lp = torch.zeros_like(p), torch.zeros_like(p), torch.empty(n, dtype=torch.bfloat16, device="cuda")
    call = lambda s: C.check(C.lib().bg_adam_step(C.ptr(p), C.ptr(g), C.ptr(m), C.ptr(v), C.ptr(lp), C.BG_BF16, n, 1e-4, 0.9, 0.999,
                                                  1e-8, 0.0, s, 1.0, 1, C.stream_ptr()), "bg_adam_step")
    for s in range(1, 4):
        call(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in range(reps):
        call(4 + s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print(f"bg_adam_step n={n}: {us:8.1f} us  {34 * n / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    # `python benchmarks/adam_bw.py [n]`; under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` for the traffic passes
    for size in (sys.argv[1:] or ["124439808"]):
        run(int(size))
