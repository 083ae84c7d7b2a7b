#!/usr/bin/env python3
"""Replays ONLY the attention launches one train step makes (recorded from a real eager step, bench.record_attention_launches:
same tensors, shapes, strides) - the program to put behind `rocprofv3 --kernel-trace --stats` or `--pmc ...` when looking at the
attention kernels alone.   usage: attn_mix.py [workload=c2] [kind=bwd|fwd|both] [reps=5] [dtype=bf16|fp32]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

import bench
from branchgan import training

key = sys.argv[1] if len(sys.argv) > 1 else "c2"
kind = sys.argv[2] if len(sys.argv) > 2 else "bwd"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
use_amp = (sys.argv[4] if len(sys.argv) > 4 else "bf16") == "bf16"
wl = dict(bench.WORKLOADS[key])
dev = torch.device("cuda", 0)
model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
from branchgan import configs
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(2 * wl["B"], wl["S"], V, 1234).view(2, wl["B"], wl["S"]).to(dev)


def eager_step(i):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
        return training.train_step(model, data[i % 2], conf, g_opt, d_opt, i, 9999, rl)


eager_step(0)
rec = bench.record_attention_launches(eager_step, 1)
calls = [r for r in rec if kind == "both" or r[0] == kind]
torch.cuda.synchronize()
for _ in range(reps):
    for r in calls:
        r[1]()
torch.cuda.synchronize()
print(json.dumps({"workload": key, "kind": kind, "launches_per_step": len(calls), "reps": reps,
                  "classes": {c: sum(1 for r in calls if bench._launch_class(r[2]) == c) for c in ("root causal", "branch step", "tree pass")}}))
