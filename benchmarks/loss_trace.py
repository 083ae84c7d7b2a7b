#!/usr/bin/env python3
"""Per-step losses of the bench workload for the fused / module-by-module block path, eager or HIP-graph replay:
the four trajectories must agree to bf16 noise (same seeds, same data)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

import bench
from branchgan import configs, fused, training

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--unfused", action="store_true")
ap.add_argument("--fp32", action="store_true")
a = ap.parse_args()
fused.ENABLED = not a.unfused
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS["c2"])
torch.manual_seed(0)
model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(16 * wl["B"], wl["S"], V, 1234).view(16, wl["B"], wl["S"]).to(dev)
amp = None if a.fp32 else torch.bfloat16
if a.graph:
    from branchgan.graphs import GraphedTrainStep
    gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, amp, warmup=0 if False else 3)
    step = lambda i: gs.step(data[i % 16], i)
else:
    def step(i):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
            return training.train_step(model, data[i % 16], conf, g_opt, d_opt, i, 9999, rl)
for i in range(a.steps):
    d, g = step(i)
    print(i, "disc %.4f gen %.4f" % (float(d["loss"].detach()), float(g["loss"].detach())), {k: round(float(v), 4) for k, v in d.items() if k != "loss" and (not torch.is_tensor(v) or v.numel() == 1)}, flush=True)
