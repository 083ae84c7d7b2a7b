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
WORKLOADS["tiny"] = dict(gen="gpt2-tiny", disc="gpt2-tiny", S=64, K=8, D=3, B=4, name="rehearsal only: 2-layer 128-wide GPT-2 pair")
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


def attention_forward_roofline(wl, device, dtype, reps=5):
    """Roofline of the dominant hand-written kernel, branch_attn_fwd_kernel (the largest share of GPU time among this
    library's kernels in a step, see profiles/).  One train step launches it with three shape classes per layer - root
    causal pass, generator branch step at depth d (ancestor slots), tree pass over N nodes (generator gradient pass,
    discriminator) - so the replay below issues EXACTLY one step's mix (same counts, shapes, strides, dtype: slices of a
    packed QKV buffer like the fused block path) captured in a HIP graph, and times `reps` replays with HIP events on the
    stream they are launched on.  Algorithmic bytes / FLOPs per launch follow SURVEY.md §8(d):
      bytes = visible K/V read once (max ctx_len text rows + every slot's K rows) + Q read + O write;
      flops = 4 * Dh * (visible query-key pairs)."""
    from branchgan import configs, fused, ops
    S, K, D, B = wl["S"], wl["K"], wl["D"], wl["B"]
    e = 2 if dtype == torch.bfloat16 else 4
    sparse = np.sort(np.random.RandomState(1234).choice(S - 1, K, replace=False))
    ctx = torch.tensor(sparse + 1, dtype=torch.int32, device=device)
    ctx_sum, ctx_max = int(ctx.sum()), int(ctx.max())
    calls, tot_bytes, tot_flops = [], 0.0, 0.0

    def packed(T, H, Dh):
        return torch.randn(B, T, 3, H, Dh, device=device).to(dtype)

    def add(fn, nbytes, flops, count):
        nonlocal tot_bytes, tot_flops
        calls.extend([fn] * count)
        tot_bytes += nbytes * count
        tot_flops += flops * count

    for name, has_branch_steps, n_tree in ((wl["gen"], True, D - 1), (wl["disc"], False, D)):
        cfg = configs.LMConfig.from_any(name)
        H, Dh, L = cfg.num_attention_heads, cfg.head_dim, cfg.num_hidden_layers
        scale = 1.0 / Dh ** 0.5
        root = packed(S, H, Dh)
        att_r = torch.empty(B, S, H, Dh, device=device, dtype=dtype)
        lse_r = torch.empty(B, H, 1, S, device=device)
        kt, vt = root[:, :, 1].permute(0, 2, 1, 3), root[:, :, 2].permute(0, 2, 1, 3)
        hf = fused._heads_first
        add(lambda root=root, att_r=att_r, lse_r=lse_r, kt=kt, vt=vt, scale=scale:
            fused._attn_fwd_call(hf(root[:, :, 0], 1, S), kt, vt, [], [], ops.causal_ctx_len(S, S, device), [[]], scale, hf(att_r, 1, S), lse_r),
            4 * B * H * S * Dh * e, 4 * Dh * H * B * S * (S + 1) // 2, L)
        steps = []
        if has_branch_steps:                       # sampling passes: node at depth d sees d - 1 ancestor slots + itself
            nodes = [packed(K, H, Dh) for _ in range(D - 1)]
            for d in range(1, D):
                steps.append((nodes[d - 1], [n[:, :, 1].permute(0, 2, 1, 3) for n in nodes[:d]], [n[:, :, 2].permute(0, 2, 1, 3) for n in nodes[:d]],
                              [list(range(d))], 1))
        tree = packed(n_tree * K, H, Dh)           # one pass over n_tree chain nodes
        tk, tv = tree[:, :, 1].permute(0, 2, 1, 3), tree[:, :, 2].permute(0, 2, 1, 3)
        steps.append((tree, [tk[:, :, n * K:(n + 1) * K] for n in range(n_tree)], [tv[:, :, n * K:(n + 1) * K] for n in range(n_tree)],
                      [list(range(n + 1)) for n in range(n_tree)], n_tree))
        for qkv, sk, sv, paths, NQ in steps:
            att = torch.empty(B, NQ * K, H, Dh, device=device, dtype=dtype)
            lse = torch.empty(B, H, NQ, K, device=device)
            pairs = NQ * ctx_sum + K * sum(len(p) for p in paths)
            add(lambda qkv=qkv, sk=sk, sv=sv, paths=paths, NQ=NQ, att=att, lse=lse, kt=kt, vt=vt, scale=scale:
                fused._attn_fwd_call(hf(qkv[:, :, 0], NQ, K), kt, vt, sk, sv, ctx, paths, scale, hf(att, NQ, K), lse),
                (2 * B * H * Dh * (ctx_max + len(sk) * K) + 2 * B * H * NQ * K * Dh) * e, 4 * Dh * H * B * pairs, L)
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for fn in calls:
            fn()
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for fn in calls:
            fn()
    graph.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        graph.replay()
    t1.record()
    torch.cuda.synchronize()
    n = len(calls)
    avg_ms = t0.elapsed_time(t1) / (reps * n)
    # HBM traffic per launch from the PMC passes over this same launch mix (benchmarks/attn_mix.py under rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 corrections applied): collected offline, a profiler cannot run in here
    traffic, pmc = None, os.path.join(ROOT, "profiles", "r1_f_pmc_attn_fwd.json")
    if wl.get("name") == WORKLOADS["c2"]["name"] and wl["B"] == WORKLOADS["c2"]["B"] and e == 2 and os.path.exists(pmc):
        traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
    achieved = tot_bytes / n / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "branch_attn_fwd_kernel (bg_branch_attn_fwd): one step's launch mix (root causal / branch step / tree pass)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/r1_f_pmc_attn_fwd.json (rocprofv3 --pmc passes over benchmarks/attn_mix.py)" if traffic else None,
            "algorithmic_bytes_per_launch": tot_bytes / n, "avg_launch_ms": avg_ms, "launches_per_step": n,
            "mfma": {"achieved_tflops": tot_flops / n / (avg_ms * 1e-3) / 1e12, "peak_tflops": 2500.0 if e == 2 else 157.3,
                     "note": "arithmetic intensity 2*Q_vis/e flop/B is far below the ridge at these shapes: HBM is the binding roof (SURVEY.md §8d)"}}


def adam_roofline(optims):
    """Roofline of bg_adam_step (adam_step_kernel), the hand-written kernel with the largest share of a step's GPU time
    (profiles/r1_n_bench_c2_by_class.txt): two launches per step, one per model.  `optims` hold the (start, end) HIP events
    branchgan.optim.ArenaAdam.step recorded around every launch INSIDE the timed region, on the stream it launched on.
    Algorithmic bytes per launch = 34 B per parameter slot of the arena (read p, g, m, v fp32; write p, m, v fp32, the bf16
    weight copy and the zeroed gradient), DESIGN.md."""
    ms, nbytes, launches = 0.0, 0.0, 0
    for opt in optims:
        ev = opt.timing or []
        ms += sum(a.elapsed_time(b) for a, b in ev)
        nbytes += 34.0 * opt.param_arena.numel() * len(ev)
        launches += len(ev)
    if not launches:
        return None
    avg_ms = ms / launches
    achieved = nbytes / launches / (avg_ms * 1e-3) / 1e9
    traffic, pmc = None, os.path.join(ROOT, "profiles", "r1_h_pmc_adam.json")
    if os.path.exists(pmc):
        rec = json.load(open(pmc))
        if abs(rec["parameters"] - nbytes / launches / 34.0) < 1e-3 * rec["parameters"]:
            traffic = rec["traffic_bytes_per_launch"]
    return {"bound": "hbm", "kernel": "adam_step_kernel (bg_adam_step): both models' optimizer launches of the timed steps",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/r1_h_pmc_adam.json (rocprofv3 --pmc passes over benchmarks/adam_bw.py)" if traffic else None,
            "algorithmic_bytes_per_launch": nbytes / launches, "avg_launch_ms": avg_ms, "launches_per_step": len(optims)}


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
    steps, t0 = 0, time.time()
    while steps < 1 or (time.time() - t0 < 12.0 and steps < 8):          # a bounded sample: ~10-30 s of CPU work
        d, g, _ = OG.train_forward(sd, ocfg(workload["gen"]), ocfg(workload["disc"]), ids, [(1, 2, {})] * D, sparse, rl=rl)
        g["loss"].backward()
        d["loss"].backward()
        for v in sd.values():
            v.grad = None
        steps += 1
    dt = time.time() - t0
    return {"value": steps * sample_batch * S / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{steps} step(s) of the same workload at batch {sample_batch} (fwd + both bwd, no optimizer), fp32, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a HIP graph")
    ap.add_argument("--no-tuned-gemm", action="store_true", help="hipBLASLt default heuristics instead of the recorded TunableOp choices")
    args = ap.parse_args()
    if os.environ.get("BRANCHGAN_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BRANCHGAN_BENCH_WATCHDOG"]), exit=True)
        probe = os.path.join(ROOT, "benchmarks", "probes", "libsegvbt.so")      # native stacks of every thread 3 s earlier
        if os.path.exists(probe):
            import ctypes
            ctypes.CDLL(probe).install_all_thread_dump(int(float(os.environ["BRANCHGAN_BENCH_WATCHDOG"])) - 3)

    from branchgan import ops, training
    tuned = (not args.no_tuned_gemm) and training.use_tuned_gemms()
    rank, world, local = training.init_data_parallel()
    if world != args.gpus and rank == 0:
        print(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; reporting n_gpus={world}", file=sys.stderr)
    device = torch.device("cuda", local % torch.cuda.device_count() if world > 1 else 0)
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
    arena_optims = [o.optimizer for o in (g_opt, d_opt) if hasattr(o.optimizer, "timing")]
    for o in arena_optims:
        o.timing = []
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        d, g = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    adam = adam_roofline(arena_optims)
    for o in arena_optims:
        o.timing = None
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
    attn = attention_forward_roofline(wl, device, torch.bfloat16 if use_amp else torch.float32)
    out = {
        "metric": "branch-GAN train tokens/sec (gen+disc fwd/bwd)", "value": tokens / elapsed, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if use_amp else "f32", "data": "synthetic",
        "config": {"workload": wl["name"], "per_gpu_batch": B, "seq_len": S, "branch_points": wl["K"], "depth": wl["D"],
                   "generator": wl["gen"], "discriminator": wl["disc"], "parallelism": f"dp{world}", "launch": "eager" if args.eager else "hipGraph replay of fwd+bwd; all-reduce and Adam eager",
                   "precision": "bf16 autocast, fp32 master weights and accumulation" if use_amp else "fp32",
                   "weights": "random init", "gemm": "hipBLASLt, recorded TunableOp choices" if tuned else "hipBLASLt defaults", "final_losses": {"gen": float(g["loss"].detach()), "disc": float(d["loss"].detach())}},
        # the dominant hand-written kernel by GPU time per step first; the attention forward mix (the dominant one before the
        # optimizer moved into this library) stays beside it
        "roofline": adam if adam is not None else attn,
        "roofline_attention_fwd": attn,
        "eager_event_ms": {k: {"launches": v[0], "avg_ms": round(v[1], 4)} for k, v in sorted(kernels.items())},   # includes host launch gaps
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, args.cpu_batch, args.dtype)
    print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
