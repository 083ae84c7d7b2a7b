"""Which autograd nodes launch the small ATen kernels of the backward passes?  One eager configs[1] step under torch.profiler;
every device kernel is attributed to the aten op that launched it and to the enclosing `autograd::engine::evaluate_function: X`
record (the C++ engine's thread has no Python stack, so py_copy_sources.py cannot see these)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import bench  # noqa: E402
from branchgan import configs, training  # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
dev = torch.device("cuda", 0)
training.use_tuned_gemms()
model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(4 * wl["B"], wl["S"], V, 1).view(4, wl["B"], wl["S"]).to(dev)


def step(i):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return training.train_step(model, data[i % 4], conf, g_opt, d_opt, i, 9999, rl)


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3)
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
shapes = collections.Counter()
for e in prof.events():
    if not e.kernels:
        continue
    dev_us = sum(k.duration for k in e.kernels)
    # innermost op that owns the kernels; walk up to the autograd node (if any)
    node, p = "(forward / python)", e.cpu_parent
    chain = [e.name]
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function:"):
            node = p.name.split(": ", 1)[1]
            break
        chain.append(p.name)
        p = p.cpu_parent
    top = chain[-1] if len(chain) > 1 else e.name
    if "Cijk" in "".join(k.name for k in e.kernels):
        continue
    if node in ("SelectBackward0", "SliceBackward0") and e.name == "aten::copy_":
        shapes[(node, str(e.input_shapes[:1]))] += 1
    key = (node, top if top != e.name else "", e.name)
    agg[key][0] += len(e.kernels)
    agg[key][1] += dev_us
print(f"{'autograd node':44s} {'outer op':28s} {'launching op':30s} launches  device us")
for (node, top, name), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if name.startswith("aten::") or "Memcpy" in name or "Memset" in name:
        print(f"{node[:44]:44s} {top[:28]:28s} {name[:30]:30s} {n:8d} {us:10.1f}")
print()
for (node, shp), n in shapes.most_common():
    print(f"{node}: copy_ into {shp} x {n}")
