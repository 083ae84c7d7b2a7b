#!/usr/bin/env python3
"""Headline benchmark: Branch-GAN train tokens/sec (generator + discriminator forward/backward, losses, gradient reduction,
both Adam steps) on synthetic Pile-shaped token data.  Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0.  For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL); per-GPU work is fixed (weak scaling).

Workload at N = 1 = BASELINE.json configs[1]: GPT-2 small generator and discriminator, seq 256, 32 branch points, depth 4
(4 x TOP_K rules, 1 alteration each, like ConfigInstances/PileTraining.py:59-71), per-GPU batch 8.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "branch-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

WORKLOADS = {
    "c2": dict(gen="gpt2-small", disc="gpt2-small", S=256, K=32, D=4, B=8,
               name="GPT-2 small G+D Branch-GAN step, seq 256, 32 branches, depth 4, batch 8/GPU (BASELINE configs[1])"),
    "c1": dict(gen="pythia-70m", disc="pythia-70m", S=128, K=16, D=2, B=4,
               name="Pythia-70M G+D, seq 128, 16 branches, depth 2, batch 4 (BASELINE configs[0])"),
    "c3": dict(gen="pythia-410m", disc="pythia-14m", S=128, K=32, D=8, B=8,
               name="Pythia-410M G / Pythia-14m D PileTraining config, seq 128, 32 branches, depth 8, batch 8/GPU (BASELINE configs[2])"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (≈6.3 TB/s achievable)


def zipf_ids(n, S, V, seed):
    """int64 (n, S) token ids, Zipf(alpha=1) over [0, V) by inverse CDF on a seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    cdf = torch.cumsum(w / w.sum(), 0)
    u = torch.rand(n * S, generator=g, dtype=torch.float64)
    return torch.searchsorted(cdf, u).clamp_(max=V - 1).view(n, S)


def build(workload, device, seed=0):
    from branchgan import configs, gan, sampling, training
    torch.manual_seed(seed)          # identical initial weights on every rank
    model = gan.GannSetup(workload["gen"], workload["disc"], mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25, device=device).to(device)
    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(workload["D"])]
    rank = int(os.environ.get("RANK", "0"))
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=workload["K"], seed=1234 + rank)
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-5, betas=(0.5, 0.999)))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-4))
    return model, conf, rl, g_opt, d_opt


def cpu_baseline(workload, sample_batch, dtype_note):
    """The oracle (CPU restatement of the reference path, plain PyTorch fp32) on a bounded sample of the same workload:
    one step (forward of the whole GAN step + both backwards) at a reduced batch, timed on this host's cores."""
    from branchgan import configs
    from oracle import gan as OG, models as OM
    from branchgan import gan
    torch.manual_seed(0)
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    model = gan.GannSetup(workload["gen"], workload["disc"])
    sd = {k: v.detach().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}

    def ocfg(name):
        c = configs.LMConfig.from_any(name)
        return OM.LMConfig("neox" if c.model_type == "gpt_neox" else "gpt2", c.num_hidden_layers, c.num_attention_heads, c.hidden_size,
                           c.vocab_size, c.max_position_embeddings)
    S, K, D = workload["S"], workload["K"], workload["D"]
    ids = zipf_ids(sample_batch, S, configs.LMConfig.from_any(workload["gen"]).vocab_size, 1234)
    sparse = np.sort(np.random.RandomState(1234).choice(S - 1, K, replace=False))
    rl = dict(label_mode=3, wA=1, wB=1, wC=1.5, teacher_forcing=True)
    t0 = time.time()
    d, g, _ = OG.train_forward(sd, ocfg(workload["gen"]), ocfg(workload["disc"]), ids, [(1, 2, {})] * D, sparse, rl=rl)
    g["loss"].backward()
    d["loss"].backward()
    dt = time.time() - t0
    return {"value": sample_batch * S / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"1 step of the same workload at batch {sample_batch} (fwd + both bwd, no optimizer), fp32, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--cpu-batch", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a HIP graph")
    args = ap.parse_args()

    from branchgan import ops, training
    rank, world, local = training.init_data_parallel()
    if world != args.gpus and rank == 0:
        print(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; reporting n_gpus={world}", file=sys.stderr)
    device = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(device)
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    model, conf, rl, g_opt, d_opt = build(wl, device)
    from branchgan import configs
    V = configs.LMConfig.from_any(wl["gen"]).vocab_size
    B, S = wl["B"], wl["S"]
    n_batches = 16
    data = zipf_ids(n_batches * B, S, V, 1234 + rank).view(n_batches, B, S).to(device)
    use_amp = args.dtype == "bf16"

    def eager_step(i):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
            return training.train_step(model, data[i % n_batches], conf, g_opt, d_opt, i, 9999, rl)

    if args.eager:
        step = eager_step
    else:
        from branchgan.graphs import GraphedTrainStep
        graphed = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, torch.bfloat16 if use_amp else None)
        step = lambda i: graphed.step(data[i % n_batches], i)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    ops.TIMING = {}
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        d, g = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if not args.eager:
        graphed.check_nan()
        for i in range(3):      # per-kernel HIP-event timing needs eager launches (events cannot sit inside a captured graph)
            eager_step(args.warmup + args.steps + i)
    kernels = ops.timing_summary()
    ops.TIMING = None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms = 1000.0 * elapsed / args.steps
    tokens = world * B * S * args.steps
    # dominant hand-written kernel: the vocabulary backward of the root pass (read logits, write dlogits)
    e = 2 if use_amp else 4
    rows = B * S
    name = f"softmax_stats_bwd[{rows}]"
    n_launch, k_ms = kernels.get(name, (0, float("nan")))
    algo_bytes = 2.0 * rows * V * e
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if n_launch else float("nan")
    out = {
        "metric": "branch-GAN train tokens/sec (gen+disc fwd/bwd)", "value": tokens / elapsed, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if use_amp else "f32", "data": "synthetic",
        "config": {"workload": wl["name"], "per_gpu_batch": B, "seq_len": S, "branch_points": wl["K"], "depth": wl["D"],
                   "generator": wl["gen"], "discriminator": wl["disc"], "parallelism": f"dp{world}", "launch": "eager" if args.eager else "hipGraph replay of fwd+bwd; all-reduce and Adam eager",
                   "precision": "bf16 autocast, fp32 master weights and accumulation" if use_amp else "fp32",
                   "weights": "random init", "final_losses": {"gen": float(g["loss"]), "disc": float(d["loss"])}},
        "roofline": {"bound": "hbm", "kernel": "softmax_stats_bwd_kernel (bg_softmax_stats_bwd, root pass)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": k_ms, "launches": n_launch},
        "kernels_ms": {k: {"launches": v[0], "avg_ms": round(v[1], 4)} for k, v in sorted(kernels.items())},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, args.cpu_batch, args.dtype)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
