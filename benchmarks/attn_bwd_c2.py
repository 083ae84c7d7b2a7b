#!/usr/bin/env python3
"""The root causal attention backward at BASELINE configs[1] (GPT-2 small: B 8, H 12, S 256, Dh 64, bf16) on packed-QKV strides,
timed with HIP events around `reps` launches replayed from a graph.  usage: attn_bwd_c2.py [S=256] [B=8]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
from branchgan import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, Dh, reps = 12, 64, 50
gen = torch.Generator().manual_seed(1)
qkv = torch.randn(B, S, 3, H, Dh, generator=gen).bfloat16().cuda()
q, k, v = [qkv[:, :, i].permute(0, 2, 1, 3).detach().requires_grad_(True) for i in range(3)]
ops.ATTN_RECORD = []
out, lse = ops.causal_attention(q, k, v, scale=1 / math.sqrt(Dh))
torch.autograd.grad(out, (q, k, v), torch.randn_like(out))
launch = [r for r in ops.ATTN_RECORD if r[0] == "bwd"][0][1]          # the backward launch itself, tensors kept alive by the closure
ops.ATTN_RECORD = None
torch.cuda.synchronize()
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    for _ in range(3):
        launch()
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            launch()
g.replay()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
g.replay()
t1.record()
torch.cuda.synchronize()
env = {k_: v_ for k_, v_ in os.environ.items() if k_.startswith("BRANCHGAN_ATTN")}
print(f"S={S} B={B} {env}: {1000 * t0.elapsed_time(t1) / reps:.2f} us per backward call", flush=True)
