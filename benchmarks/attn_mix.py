#!/usr/bin/env python3
"""Runs ONLY the attention-forward launch mix of one configs[1] train step (bench.attention_forward_roofline) - the
program to put behind `rocprofv3 --pmc ...` when collecting HBM traffic counters for the roofline's `traffic` field."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

import bench

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
print(json.dumps(bench.attention_forward_roofline(wl, torch.device("cuda", 0), torch.bfloat16, reps=2)))
