#!/usr/bin/env python3
"""Which lines of branchgan/ issue the small ATen ops of one eager train step: a TorchDispatchMode that books every op under
the innermost branchgan/ frame of the Python stack (ops of autograd's own backward nodes have none: booked as '<autograd>')."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import bench  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.permute", "aten.transpose", "aten.t.", "aten.slice", "aten.select", "aten.expand", "aten.unsqueeze",
        "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.empty", "aten.reshape", "aten.unflatten", "aten.split", "aten.unbind",
        "aten.mm.", "aten.addmm", "aten.lift_fresh", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.numel", "aten._local_scalar_dense")


class Book(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "<autograd>"
            for fr in reversed(traceback.extract_stack(limit=30)):
                if "branchgan/" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            self.ops[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


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
    with Book() as book:
        step(3)
    torch.cuda.synchronize()
    total = 0
    for (name, where, shape), n in sorted(book.ops.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        total += n
        print(f"{n:4d}  {where:22s} {name:34s} {shape}")
    print("total", total)


if __name__ == "__main__":
    main()
