#!/usr/bin/env python3
"""A few launches of bg_rank_stats at BASELINE config 5's shape (8192 x 50304, the reference's eight ranks) for counter passes:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 benchmarks/rank_stats_once.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
from branchgan import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(2)
probs = torch.softmax(3.0 * torch.randn(8192, 50304, device="cuda", generator=g), -1)
labels = torch.randint(0, 50304, (8192,), device="cuda", generator=g)
for _ in range(6):
    ops.rank_stats(probs, labels, (1, 8, 16, 64, 128, 256, 512, 1024))
torch.cuda.synchronize()
