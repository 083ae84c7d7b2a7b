"""How long does each captured graph of the configs[1] step take on its own, and what runs between them?  (Sizing what an overlap of
the generator's backward with the discriminator's could hide.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import bench  # noqa: E402
from branchgan import configs, training  # noqa: E402
from branchgan.graphs import GraphedTrainStep  # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"])
dev = torch.device("cuda", 0)
training.use_tuned_gemms()
model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(4 * wl["B"], wl["S"], V, 1).view(4, wl["B"], wl["S"]).to(dev)
gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, torch.bfloat16)
for i in range(3):
    gs.step(data[i % 4], i)
torch.cuda.synchronize()


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("whole step            %.3f ms" % timed(lambda: gs.step(data[0], 1)))
print("graph_g (G forward)   %.3f ms" % timed(gs.graph_g.replay))
print("graph   (D forward + losses%s) %.3f ms" % ("" if gs.graph_gb is not None else " + G backward", timed(gs.graph.replay)))
if gs.graph_gb is not None:
    print("graph_gb (G backward) %.3f ms" % timed(gs.graph_gb.replay))
print("graph_d (D backward)  %.3f ms" % timed(gs.graph_d.replay))
print("all graphs as a step replays them %.3f ms" % timed(gs.replay_graphs))
print("G adam + zero         %.3f ms" % timed(lambda: g_opt._step_and_zero()))
print("D adam + zero         %.3f ms" % timed(lambda: d_opt._step_and_zero()))
print("census (nodes, kernels, memsets replaced, memcpys per graph):", gs.census)

# potential of the discriminator's forward beside the generator's (TIMING ONLY: the D forward reads the samples the G forward
# writes; only its pass over the real text, 2048 of 3072 rows, could really run there)
aux2 = torch.cuda.Stream()


def fwd_both():
    main = torch.cuda.current_stream()
    aux2.wait_stream(main)
    with torch.cuda.stream(aux2):
        gs.graph.replay()
    gs.graph_g.replay()
    main.wait_stream(aux2)


print("graph (D forward + losses) on a second stream beside graph_g: %.3f ms (one after the other: the sum above)" % timed(fwd_both))
