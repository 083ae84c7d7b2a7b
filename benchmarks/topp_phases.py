#!/usr/bin/env python3
"""Time bg_topp_select at 8192 x 50304.  (Round 1 phase breakdown, taken with temporary early exits in topp_regs_kernel,
p = 0.5: row load 0.29 ms, exponent level +0.19, mantissa levels and cut +0.91, noise read / keys / top-n / merge +1.13.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
from branchgan import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(2)
probs = torch.softmax(3.0 * torch.randn(8192, 50304, device="cuda", generator=g), -1)
noise = -torch.log(torch.rand(8192, 50304, device="cuda", generator=g) + 1e-14)
for p in (0.5, 0.9):
    for _ in range(3):
        ops.topp_select(probs, 4, noise, p)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.topp_select(probs, 4, noise, p)
    e.record()
    torch.cuda.synchronize()
    print(f"p={p}: {s.elapsed_time(e) / 10:.3f} ms")
