#!/usr/bin/env python3
"""Which ATen ops of one eager train step end in a device-to-device memcpy (hipMemcpyAsync -> __amd_rocclr_copyBuffer)."""
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
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(3)
        torch.cuda.synchronize()
    agg = collections.Counter()
    for ev in prof.events():
        for k in getattr(ev, "kernels", []) or []:
            if "emcpy" in k.name or "copyBuffer" in k.name:
                agg[(ev.name, str(ev.input_shapes)[:90], k.name[:30])] += 1
    for (name, shapes, kname), n in agg.most_common(40):
        print(f"{n:4d}  {name:28s} {kname:30s} {shapes}")


if __name__ == "__main__":
    main()
