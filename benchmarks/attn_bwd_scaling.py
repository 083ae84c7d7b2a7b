#!/usr/bin/env python3
"""Root causal attention forward + backward at growing text lengths: run under
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 benchmarks/attn_bwd_scaling.py
and summarise <dir>/*kernel_trace.csv with `--summarise <csv>`: average duration per (kernel, grid), i.e. per text length -
the slope is the cost of one more key / query tile on a wave's critical path, the intercept the fixed cost of a launch."""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))

LENGTHS = (32, 64, 128, 256, 512, 1024)
B, H, DH, REPS = 8, 12, 64, 20


def run():
    import torch
    from branchgan import ops
    dev = torch.device("cuda", 0)
    for S in LENGTHS:
        q, k, v = (torch.randn(B, H, S, DH, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
        ctx = ops.causal_ctx_len(S, S, dev)
        for _ in range(REPS):
            out, _ = ops.branch_attention(q.view(B, H, 1, S, DH), k, v, ctx)
            torch.autograd.grad(out, (q, k, v), torch.ones_like(out))
        torch.cuda.synchronize()


def summarise(path):
    rows = list(csv.DictReader(open(path)))
    acc = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        if "attn" not in name:
            continue
        short = name.split("branch_attn_")[-1].split("_kernel")[0]
        grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]))
        acc[(short, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"{'kernel':12s} {'workgroups':>14s} {'launches':>8s} {'median us':>10s} {'min us':>8s}")
    for (short, grid), d in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][1][0] * kv[0][1][1])):
        d.sort()
        print(f"{short:12s} {str(grid):>14s} {len(d):8d} {d[len(d) // 2]:10.1f} {d[0]:8.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run()
