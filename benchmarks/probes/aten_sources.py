"""Where do the small ATen launches of a training step come from?  One eager configs[1] step under torch.profiler with Python
stacks; device time and launch count of every aten op that launches something, grouped by the innermost branchgan/ source line."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import bench  # noqa: E402
from branchgan import training  # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
dev = torch.device("cuda", 0)
training.use_tuned_gemms()
model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
from branchgan import configs  # noqa: E402
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(4 * wl["B"], wl["S"], V, 1).view(4, wl["B"], wl["S"]).to(dev)


def step(i):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return training.train_step(model, data[i % 4], conf, g_opt, d_opt, i, 9999, rl)


for i in range(3):
    step(i)
torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

SKIP = ("aten.view", "aten.as_strided", "aten.permute", "aten.transpose", "aten.t.", "aten.slice", "aten.select", "aten.expand",
        "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten._unsafe_view", "aten.reshape", "aten.unflatten",
        "aten.empty", "aten.mm.", "aten.addmm", "aten.baddbmm", "aten.split", "aten.unbind", "aten.lift_fresh", "aten._reshape_alias")
agg = collections.defaultdict(int)


class Where(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "branchgan" in fr.filename and "_python_dispatch" not in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            big = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
            agg[(name, where, "big" if big >= 1 << 16 else "small")] += 1
        return func(*args, **(kwargs or {}))


with Where():
    step(3)
    torch.cuda.synchronize()
print(f"{'op':34s} {'calls':>6s} size   where")
for (name, where, size), n in sorted(agg.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{name:34s} {n:6d} {size:6s} {where}")
