import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "branch-gan_amd"))
import torch
from branchgan import ops
rows, V = 8192, 50304
g = torch.Generator(device="cuda").manual_seed(2)
probs = torch.softmax(3.0 * torch.randn(rows, V, device="cuda", generator=g), -1)
noise = -torch.log(torch.rand(rows, V, device="cuda", generator=g) + 1e-14)
blocked = torch.randint(0, V, (rows, 1), device="cuda", generator=g)
for p in (0.5, 0.9):
    ops.topp_select(probs, 4, noise, p, blocked=blocked)
    torch.cuda.synchronize()
    ws = ops._workspace(4 * (rows + 1), probs.device)
    t = ws.view(torch.int32)[:rows + 1].cpu()
    cnt = int(t[0])
    why = (t[1:1 + cnt] >> 24).tolist()
    names = {1: "p beyond the row's mass", 2: "cut among denormals", 3: "list overflow", 4: "tie at the cut", 5: "mass not finite / zero", 6: "zero or NaN noise"}
    print("p", p, "rows left to the first form:", cnt, {names.get(k, k): why.count(k) for k in sorted(set(why))})
