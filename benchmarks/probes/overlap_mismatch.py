"""Which gradients differ between the overlapped graph replays and an eager step on the module-by-module path (bf16)?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_graphs import _setup  # noqa: E402
from branchgan import configs, fused  # noqa: E402
from branchgan.graphs import GraphedTrainStep  # noqa: E402

fused.ENABLED = sys.argv[1] != "unfused"
amp = torch.bfloat16
V = configs.LMConfig.from_any("gpt2-small").vocab_size
data = torch.randint(0, V, (5, 8, 256), generator=torch.Generator().manual_seed(6)).cuda()
model, conf, rl, g_opt, d_opt = _setup(models=("gpt2-small", "gpt2-small"), depth=4, K=32)
conf.sparseIndex = torch.tensor(np.sort(np.random.RandomState(3).choice(255, 32, replace=False)), device="cuda")
gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, amp, warmup=1)
names = {}
for tag, opt, mod in (("G", g_opt, model.generator), ("D", d_opt, model.discriminator)):
    by_id = {id(p): n for n, p in mod.named_parameters()}
    names[tag] = [(by_id[id(p)], off, p.numel()) for p, off in zip(opt.arena.params, opt.arena.offsets)]
for r in range(3):
    ids = data[1 + r]
    gs.static_ids.copy_(ids)
    gs.replay_graphs()
    torch.cuda.synchronize()
    got = {"G": g_opt.arena.arena.clone(), "D": d_opt.arena.arena.clone()}
    g_opt.arena.zero(); d_opt.arena.zero()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        d, g = model.trainForwardPassNoMetrics(ids, conf, 9999, collectTreeDepthData=False, reinforcementLearningConfig=rl)
        g["loss"].backward()
        d["loss"].backward()
    torch.cuda.synchronize()
    for tag, opt in (("G", g_opt), ("D", d_opt)):
        a, b = got[tag], opt.arena.arena
        scale = float(b.abs().max())
        bad = []
        for n, off, cnt in names[tag]:
            e = float((a[off:off + cnt] - b[off:off + cnt]).abs().max())
            if e > 0:
                bad.append((n, e / scale, int(((a[off:off + cnt] - b[off:off + cnt]) != 0).sum()), cnt))
        print(f"replay {r} {tag}: {len(bad)} of {len(names[tag])} parameters differ", flush=True)
        for n, e, k, cnt in bad[:12]:
            print(f"    {n:50s} max diff {e:.3e} of max|grad|, {k} of {cnt} elements")
    g_opt.arena.zero(); d_opt.arena.zero()
