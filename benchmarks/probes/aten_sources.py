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


# the backward pass runs on the autograd engine's thread, which a dispatch mode of this thread does not see: the profiler's
# Python stacks instead (innermost branchgan frame per aten op that launches a kernel)
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(4)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=12):
    if not e.key.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    where = "?"
    for fr in e.stack:
        if "branchgan" in fr:
            where = fr.split("branch-gan_amd/")[-1]
            break
    rows.append((e.count, e.self_device_time_total, e.key, where))
rows.sort(key=lambda r: -r[1])
print(f"\n{'op':26s} {'calls':>6s} {'device us':>10s}  where (whole step, forward and backward)")
for n, us, key, where in rows[:60]:
    if key in ("aten::mm", "aten::addmm", "aten::baddbmm"):
        continue
    print(f"{key:26s} {n:6d} {us:10.1f}  {where[:140]}")

# per phase of the step: which aten ops launch kernels (the backward passes' ops included: they fall inside the CPU time range of
# their .backward() call)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof2:
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.profiler.record_function("PHASE generator forward"):
            gen_state = model._generatorForward(data[0], conf)
            torch.cuda.synchronize()
        with torch.profiler.record_function("PHASE discriminator forward + losses"):
            d, g = model.trainForwardPassNoMetrics(data[0], conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl,
                                                   generatorState=gen_state)
            torch.cuda.synchronize()
    with torch.profiler.record_function("PHASE generator backward"):
        g["loss"].backward()
        torch.cuda.synchronize()
    with torch.profiler.record_function("PHASE discriminator backward"):
        d["loss"].backward()
        torch.cuda.synchronize()
evs = prof2.events()
phases = [(e.name, e.time_range.start, e.time_range.end) for e in evs if e.name.startswith("PHASE")]
per = {name: collections.Counter() for name, _, _ in phases}
us = {name: collections.Counter() for name, _, _ in phases}
for e in evs:
    if not e.name.startswith("aten::") or e.self_device_time_total <= 0 or e.name in ("aten::mm", "aten::addmm", "aten::baddbmm"):
        continue
    for name, a, b in phases:
        if a <= e.time_range.start <= b:
            per[name][e.name] += 1
            us[name][e.name] += e.self_device_time_total
for name, _, _ in phases:
    print("\\n" + name)
    for k, n in per[name].most_common(12):
        print(f"    {k:26s} {n:5d} launches {us[name][k]:9.1f} us")
g_opt.arena.zero()
d_opt.arena.zero()
