#!/usr/bin/env python3
"""Where the attention kernels saturate: root causal attention (forward, backward) at shapes far beyond the BASELINE
configs - 256 (batch, head) pairs, S 1024 / 2048 / 4096, Dh 64 / 128, bf16 - in TFLOP/s of the visible pairs (SURVEY.md §8d:
forward 4 Dh B H S (S + 1) / 2, backward 2.5 x) against the dense bf16 MFMA peak.  The BASELINE shapes are latency-bound
(a workgroup lives for a handful of tiles); this is the kernels' throughput once a wave has hundreds of tiles to chew on.
usage: attn_large.py [reps=5]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

from branchgan import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
PEAK = 2500.0
torch.manual_seed(0)
out = []
for Dh in (64, 128):
    for S in (1024, 2048, 4096):
        B, H = 16, 16
        q, k, v = (torch.randn(B, H, S, Dh, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
        scale = Dh ** -0.5
        o, _ = ops.causal_attention(q, k, v, scale=scale)
        do = torch.randn_like(o)
        o.backward(do)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(reps):
            q.grad = k.grad = v.grad = None
            e[0].record()
            o, _ = ops.causal_attention(q, k, v, scale=scale)
            e[1].record()
            o.backward(do)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        tf, tb = tf / reps, tb / reps
        flops = 4.0 * Dh * B * H * S * (S + 1) / 2
        row = {"Dh": Dh, "S": S, "BH": B * H, "fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3),
               "fwd_TFLOPs": round(flops / tf / 1e9, 1), "bwd_TFLOPs": round(2.5 * flops / tb / 1e9, 1),
               "fwd_frac_mfma": round(flops / tf / 1e9 / PEAK, 3), "bwd_frac_mfma": round(2.5 * flops / tb / 1e9 / PEAK, 3)}
        out.append(row)
        print(json.dumps(row), flush=True)
        del q, k, v, o, do
