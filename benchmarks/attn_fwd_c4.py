#!/usr/bin/env python3
"""The generator's branch step at BASELINE configs[3] (Pythia-1B: B 4, H 8, Dh 256, 64 queries against 512 text keys + n_anc
ancestor slots, bf16) - forward only, timed with HIP events around `reps` back-to-back launches.
usage: attn_fwd_c4.py [n_anc ...]      (BRANCHGAN_ATTN_FWD_KVSPLIT=0|N is read by the library once per process)"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
from branchgan import ops  # noqa: E402

B, H, Dh, K, S = 4, 8, 256, 64, 512
reps = 50
gen = torch.Generator().manual_seed(1)
mk = lambda *s: torch.randn(*s, generator=gen).to(torch.bfloat16).cuda()
kt, vt = mk(B, H, S, Dh), mk(B, H, S, Dh)
ctx_len = torch.sort(torch.randperm(S - 1, generator=gen)[:K]).values.add(1).int().cuda()
for n_anc in [int(a) for a in sys.argv[1:]] or [0, 3, 6]:
    q = mk(B, H, 1, K, Dh)
    ks, vs = [mk(B, H, K, Dh) for _ in range(n_anc + 1)], [mk(B, H, K, Dh) for _ in range(n_anc + 1)]
    run = lambda: ops.branch_attention(q, kt, vt, ctx_len, ks[:-1], vs[:-1], scale=1 / math.sqrt(Dh), own_k=ks[-1].unsqueeze(2), own_v=vs[-1].unsqueeze(2))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay()
    torch.cuda.synchronize()
    t0.record()
    g.replay()
    t1.record()
    torch.cuda.synchronize()
    print(f"KVSPLIT={os.environ.get('BRANCHGAN_ATTN_FWD_KVSPLIT', 'auto')} n_anc={n_anc}: {1000 * t0.elapsed_time(t1) / reps:.2f} us per call", flush=True)
