#!/usr/bin/env python3
"""The three sampling kernels at the north-star size (8192 x 50304 fp32, n = 4), a few launches each and nothing else of weight:
the program behind `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (benchmarks/probes/pmc_traffic.sh).
usage: sampling_once.py [rows=8192] [V=50304] [reps=3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

from branchgan import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
V = int(sys.argv[2]) if len(sys.argv) > 2 else 50304
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator(device="cuda").manual_seed(2)
probs = torch.softmax(3.0 * torch.randn(rows, V, device="cuda", generator=g), -1)
noise = -torch.log(torch.rand(rows, V, device="cuda", generator=g) + 1e-14)
blocked = torch.randint(0, V, (rows, 1), device="cuda", generator=g)
for _ in range(reps):
    ops.topn_select(probs, 4, blocked=blocked)                    # TOP_K
    ops.topn_select(probs, 4, noise=noise, blocked=blocked)       # SAMPLE_NO_REPLACEMENT
    ops.topp_select(probs, 4, noise, 0.9, blocked=blocked)        # TOP_P_SAMPLE_NO_REPLACEMENT
torch.cuda.synchronize()
print("done")
