#!/usr/bin/env python3
"""Which lines of branchgan/ launch the small ATen kernels of one eager train step (torch.profiler with stacks)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import bench  # noqa: E402


def main():
    from branchgan import configs, training
    wl = dict(bench.WORKLOADS["c2"])
    dev = torch.device("cuda", 0)
    model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
    V = configs.LMConfig.from_any(wl["gen"]).vocab_size
    data = bench.zipf_ids(4 * wl["B"], wl["S"], V, 1234).view(4, wl["B"], wl["S"]).to(dev)

    def step(i):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            training.train_step(model, data[i % 4], conf, g_opt, d_opt, i, 9999, rl)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step(3)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        dev_us = getattr(ev, "self_device_time_total", 0) or 0
        if dev_us <= 0:
            continue
        frame = next((f for f in (ev.stack or []) if "branchgan" in f or "bench.py" in f), "")
        frame = frame.split("branch-gan_amd/")[-1] + " " + str(ev.input_shapes)
        agg[(ev.name, frame)][0] += 1
        agg[(ev.name, frame)][1] += dev_us
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for (name, frame), (n, us) in rows[:70]:
        if name in ("aten::mm", "aten::addmm"):
            continue
        print(f"{n:4d} {us:9.1f} us  {name:28s} {frame[:110]}")


if __name__ == "__main__":
    main()
