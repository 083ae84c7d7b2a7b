"""Which Python call sites of one eager configs[1] step launch copy / cast / fill / add kernels?  torch.Tensor.contiguous / to /
float / copy_ / add_ / clone and torch.zeros / zeros_like / cat / stack are wrapped (this works on the autograd engine's thread
too, which dispatch modes and the profiler's Python stacks do not reach here); a call is counted when it really launches
(non-contiguous input, dtype change, ...), with the innermost branchgan frame and the tensor size."""
import collections
import os
import sys
import traceback

import torch

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
agg = collections.defaultdict(lambda: [0, 0])
on = [False]


def where():
    for fr in reversed(traceback.extract_stack(limit=30)):
        if "branchgan" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def note(kind, t):
    if on[0] and isinstance(t, torch.Tensor) and t.is_cuda:
        a = agg[(kind, where())]
        a[0] += 1
        a[1] += t.numel() * t.element_size()


def wrap_method(name, launches):
    orig = getattr(torch.Tensor, name)

    def f(self, *args, **kw):
        if launches(self, args, kw):
            note(name, self)
        return orig(self, *args, **kw)
    setattr(torch.Tensor, name, f)


def to_launches(self, args, kw):
    dt = kw.get("dtype") or next((a for a in args if isinstance(a, torch.dtype)), None)
    return dt is not None and dt != self.dtype


wrap_method("contiguous", lambda s, a, k: not s.is_contiguous())
wrap_method("to", to_launches)
wrap_method("float", lambda s, a, k: s.dtype != torch.float32)
wrap_method("bfloat16", lambda s, a, k: s.dtype != torch.bfloat16)
wrap_method("copy_", lambda s, a, k: True)
wrap_method("add_", lambda s, a, k: True)
wrap_method("clone", lambda s, a, k: True)
wrap_method("zero_", lambda s, a, k: True)
wrap_method("reshape", lambda s, a, k: not s.is_contiguous())
for fn in ("zeros", "zeros_like", "cat", "stack", "full", "ones_like"):
    orig = getattr(torch, fn)

    def g(*args, _o=orig, _n=fn, **kw):
        out = _o(*args, **kw)
        note(_n, out)
        return out
    setattr(torch, fn, g)

on[0] = True
step(3)
torch.cuda.synchronize()
on[0] = False
print(f"{'call':12s} {'count':>6s} {'MB':>9s}  where")
for (kind, w), (n, nbytes) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{kind:12s} {n:6d} {nbytes / 1e6:9.2f}  {w}")
