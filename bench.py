#!/usr/bin/env python3
"""Headline benchmark: Branch-GAN train tokens/sec (generator + discriminator forward/backward, losses, gradient reduction,
both Adam steps) on synthetic Pile-shaped token data.  Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0.  For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL); per-GPU work is fixed (weak scaling).

Workload at N = 1 = BASELINE.json configs[1]: GPT-2 small generator and discriminator, seq 256, 32 branch points, depth 4
(4 x TOP_K rules, 1 alteration each, like ConfigInstances/PileTraining.py:59-71), per-GPU batch 8.
Other workloads (parity-test cases, not the driver's line): --workload c1 | c3 | c4 (train steps of BASELINE configs[0], [2], [3]),
--workload c5 (BASELINE configs[4]: the branch-sampling sweep at 8192 x 50304).

`roofline` = the dominant hand-written hot-path entry point, bg_branch_attn_bwd: ONE eager step records every attention launch
it makes (ops.ATTN_RECORD: same tensors, shapes, strides), the recorded launches are captured into a HIP graph and replayed,
timed with HIP events on the stream they run on.  bg_branch_attn_fwd (same method) and bg_adam_step (events around the launches
of the timed steps) sit beside it.
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
    "c4": dict(gen="pythia-1b", disc="pythia-410m", S=512, K=64, D=8, B=4,
               name="Pythia-1B G / Pythia-410M D, seq 512, 64 branches, depth 8, batch 4/GPU (BASELINE configs[3], attention-bound stress)"),
    "c5": dict(rows=8192, V=50304,
               name="branch sampling sweep, 64 x 128 rows over a 50304 vocabulary: TOP_K n 1/4/16, TOP_P p 0.5/0.9, SAMPLE_NO_REPLACEMENT n 1/4 (BASELINE configs[4])"),
}
WORKLOADS["tiny"] = dict(gen="gpt2-tiny", disc="gpt2-tiny", S=64, K=8, D=3, B=4, name="rehearsal only: 2-layer 128-wide GPT-2 pair")
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (≈6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
PMC_TOPN, PMC_TOPP = "r3_pmc_topn.json", "r3_pmc_topp.json"      # profiles/: rocprofv3 --pmc passes over benchmarks/sampling_mix.py


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
    lm = lambda name: configs.LMConfig.from_any(name)
    if workload.get("dropout"):      # --dropout P: the stock GPT-2 rates are 0.1 (attention probabilities: inside the attention kernels)
        import dataclasses
        p = workload["dropout"]
        lm = lambda name: dataclasses.replace(configs.LMConfig.from_any(name), attn_pdrop=p, resid_pdrop=p,
                                              **({"embd_pdrop": p} if configs.LMConfig.from_any(name).model_type != "gpt_neox" else {}))
    model = gan.GannSetup(lm(workload["gen"]), lm(workload["disc"]), mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25, device=device).to(device)
    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(workload["D"])]
    rank = int(os.environ.get("RANK", "0"))
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=workload["K"], seed=1234 + rank)
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-5, betas=(0.5, 0.999)))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-4))
    return model, conf, rl, g_opt, d_opt


def _pmc_traffic(name, expect_launches):
    """HBM bytes per launch from the rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE runs over benchmarks/attn_mix.py or
    benchmarks/sampling_mix.py, gfx950 corrections applied; a profiler cannot run inside this process).  Only used when the record
    was taken on the same launch mix."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path) and name.startswith("r3_"):      # the round-3 records first, round 2's where there is none
        name = "r2_" + name[3:]
        path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    if expect_launches is not None and rec.get("launches_per_step") not in (None, expect_launches):
        return None, None
    # (records taken per kernel dispatch carry the per-entry-point-call figure beside it)
    return rec.get("traffic_bytes_per_api_launch", rec.get("traffic_bytes_per_launch")), f"profiles/{name}"


def _time_graph(calls, stream, reps):
    """Captures the launch closures into one HIP graph on `stream` and returns the mean ms per launch over `reps` replays,
    timed with HIP events recorded on that stream."""
    with torch.cuda.stream(stream):
        for fn in calls:
            fn()
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        for fn in calls:
            fn()
    with torch.cuda.stream(stream):
        graph.replay()
        stream.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            graph.replay()
        t1.record()
        stream.synchronize()
    return t0.elapsed_time(t1) / (reps * len(calls))


def record_attention_launches(eager_step, index):
    """Runs ONE eager train step with ops.ATTN_RECORD on: every bg_branch_attn_fwd / _bwd launch of the step, with the tensors it
    ran on kept alive."""
    from branchgan import ops
    ops.ATTN_RECORD = []
    try:
        eager_step(index)
        torch.cuda.synchronize()
    finally:
        rec, ops.ATTN_RECORD = ops.ATTN_RECORD, None
    return rec


def _launch_class(sh):
    if sh.n_slots == 0:
        return "root causal"
    return "branch step" if sh.NQ == 1 else "tree pass"


def attention_roofline(rec, kind, dtype_name, device, workload_key, reps=5):
    """Roofline of bg_branch_attn_fwd / bg_branch_attn_bwd on one step's real launch mix (see the module docstring).
    Algorithmic bytes / FLOPs per launch: ops.attn_launch_work.  For the backward a launch is the entry point's
    kernels together (dq, dkv, slots)."""
    from branchgan import ops
    calls = [r for r in rec if r[0] == kind]
    if not calls:
        return None
    work = [ops.attn_launch_work(r[2], r[3], r[4], kind, r[5][2]) for r in calls]
    stream = torch.cuda.Stream(device=device)
    stream.wait_stream(torch.cuda.current_stream())
    n = len(calls)
    avg_ms = _time_graph([r[1] for r in calls], stream, reps)
    tot_bytes, tot_flops = sum(w[0] for w in work), sum(w[1] for w in work)
    classes = {}
    for cls in ("root causal", "branch step", "tree pass"):
        idx = [i for i, r in enumerate(calls) if _launch_class(r[2]) == cls]
        if idx:
            ms = _time_graph([calls[i][1] for i in idx], stream, reps)
            b, f = sum(work[i][0] for i in idx) / len(idx), sum(work[i][1] for i in idx) / len(idx)
            classes[cls] = {"launches_per_step": len(idx), "avg_launch_us": round(1e3 * ms, 2), "GBps": round(b / ms / 1e6, 1),
                            "TFLOPs": round(f / ms / 1e9, 2)}
    achieved = tot_bytes / n / (avg_ms * 1e-3) / 1e9
    traffic, src = _pmc_traffic(f"r3_pmc_attn_{kind}_{workload_key}.json", n) if dtype_name == "bf16" else (None, None)
    entry = "bg_branch_attn_fwd (branch_attn_fwd_kernel / _split_kernel)" if kind == "fwd" else \
        "bg_branch_attn_bwd (branch_attn_bwd_one_kernel: key tiles, query tiles and slot rows of a call in one grid; branch_attn_bwd_head_kernel for slot-free calls on >= 64 heads; the split kernels at Dh 256)"
    tf = tot_flops / n / (avg_ms * 1e-3) / 1e12
    return {"bound": "hbm", "kernel": f"{entry}: one train step's recorded launch mix, timed in a side replay of the step's recorded launches "
                                      "(a HIP graph on its own stream, HIP events on that stream) after the timed region",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": src, "algorithmic_bytes_per_launch": tot_bytes / n, "avg_launch_ms": avg_ms, "launches_per_step": n,
            "by_class": classes,
            "mfma": {"achieved_tflops": tf, "peak_tflops": MFMA_PEAK_TFLOPS[dtype_name], "frac": tf / MFMA_PEAK_TFLOPS[dtype_name],
                     "note": "arithmetic intensity 2*Q_vis/e flop/B is far below the ridge at these shapes: HBM is the binding roof (SURVEY.md §8d)"}}


def adam_roofline(optims):
    """bg_adam_step (adam_step_kernel): two launches per step, one per model.  `optims` hold the (start, end) HIP events
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


def host_cpu():
    """(threads this process may use, CPU model string, CPUs the host reports).  The GPU box gives a one-GPU job a 16-CPU share
    of a much larger machine and exports OMP_NUM_THREADS accordingly: that share is what the baseline runs on."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = min(avail, int(os.environ.get("OMP_NUM_THREADS", str(avail))))
    return max(threads, 1), model, os.cpu_count()


def cpu_baseline(workload, sample_batch):
    """The oracle (CPU restatement of the reference path, plain PyTorch fp32) on a bounded sample of the same workload:
    whole steps (forward of the GAN step + both backwards) at a reduced batch, timed on this host's cores."""
    from branchgan import configs
    from oracle import gan as OG, models as OM
    from branchgan import gan
    torch.manual_seed(0)
    threads, model_name, host_cpus = host_cpu()
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
    return {"value": steps * sample_batch * S / dt, "unit": "tokens/s", "cores": threads, "kind": "port", "cpu_model": model_name,
            "host_cpus": host_cpus,
            "sample": f"{steps} step(s) of the same workload at batch {sample_batch} (fwd + both bwd, no optimizer), fp32, {dt:.1f} s"}


# ------------------------------------------------------------------------------------------------------------------------------
def sampling_cases(rows, V, dev):
    """The seven cases of the sweep on resident inputs: (name, launch, algorithmic bytes, kernel family)."""
    from branchgan import ops
    g = torch.Generator(device=dev).manual_seed(2)
    probs = torch.softmax(3.0 * torch.randn(rows, V, device=dev, generator=g), -1)
    noise = -torch.log(torch.rand(rows, V, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) + 1e-14)
    blocked = torch.randint(0, V, (rows, 1), device=dev, generator=g)
    cases = [(f"TOP_K n={n}", lambda n=n: ops.topn_select(probs, n, blocked=blocked), rows * V * 4 + rows * n * 12, "topn") for n in (1, 4, 16)]
    cases += [(f"SAMPLE_NO_REPLACEMENT n={n}", lambda n=n: ops.topn_select(probs, n, noise=noise, blocked=blocked), 2 * rows * V * 4 + rows * n * 12, "topn")
              for n in (1, 4)]
    cases += [(f"TOP_P p={p} n=4", lambda p=p: ops.topp_select(probs, 4, noise, p, blocked=blocked), 2 * rows * V * 4 + rows * 4 * 12, "topp")
              for p in (0.5, 0.9)]
    return cases, probs, noise, blocked


def sampling_rooflines(rows, V, dev, steps, warmup):
    """The seven cases of the sweep on resident (rows, V) inputs, each launch between HIP events on the stream it runs on (the
    library launches on torch's current stream).  -> (per-case table, roofline of the top-n cases, roofline of the top-p
    cases, wall seconds of the timed passes, resident inputs).  Algorithmic bytes per case: one read of the scores (+ one of
    the noise) + rows * n * 12 bytes of output; `traffic` = the PMC record taken on the same mix (benchmarks/sampling_mix.py)."""
    cases, probs, noise, _ = sampling_cases(rows, V, dev)
    for _ in range(warmup):
        for _, fn, _, _ in cases:
            fn()
    torch.cuda.synchronize()
    per_case = {name: [] for name, *_ in cases}
    t_wall = time.perf_counter()
    for _ in range(steps):
        for name, fn, _, _ in cases:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            fn()
            t1.record()
            per_case[name].append((t0, t1))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_wall
    table, groups = {}, {"topn": [0.0, 0.0], "topp": [0.0, 0.0]}
    for name, _, nbytes, grp in cases:
        ms = sum(a.elapsed_time(b) for a, b in per_case[name]) / len(per_case[name])
        table[name] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1), "hbm_frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 3)}
        groups[grp][0] += nbytes
        groups[grp][1] += ms

    def roof(grp, kernel, pmc):
        b, ms = groups[grp]
        n = sum(1 for c in cases if c[3] == grp)
        traffic, src = _pmc_traffic(pmc, None)
        ach = b / ms / 1e6
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": b / n, "avg_launch_ms": ms / n,
                "rows": rows, "vocab": V}
    topn = roof("topn", "topn_partial_kernel (bg_topn_select): the five TOP_K / SAMPLE_NO_REPLACEMENT cases of the BASELINE configs[4] sweep "
                "(TOP_K n 1/4/16, SAMPLE_NO_REPLACEMENT n 1/4), each launch between HIP events", PMC_TOPN)
    topp = roof("topp", "topp kernels (bg_topp_select): the two TOP_P cases of the sweep (p 0.5 / 0.9, n 4), each call between HIP events", PMC_TOPP)
    return table, topn, topp, elapsed, len(cases), (probs, noise)


def sampling_sweep(args):
    """BASELINE configs[4]: the per-step branch sampling over the vocabulary at batch 64 x 128 rows, V = 50304 (SURVEY.md §8d:
    TOP_K n in {1,4,16}; TOP_P p in {0.5,0.9}, n = 4; SAMPLE_NO_REPLACEMENT n in {1,4}).  A "step" = one pass of the whole sweep
    over the resident (rows, V) probabilities and noise."""
    from oracle import sampling as OS
    wl = WORKLOADS["c5"]
    rows, V = wl["rows"], wl["V"]
    dev = torch.device("cuda", 0)
    table, topn, topp, elapsed, n_cases, (probs, noise) = sampling_rooflines(rows, V, dev, args.steps, args.warmup)
    out = {"metric": "branch sampling sweep rows/sec (top-k / top-p / no-replacement over a 50k vocabulary)",
           "value": rows * n_cases * args.steps / elapsed, "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "config": {"workload": wl["name"], "rows": rows, "vocab": V, "inputs": "probs = softmax(3 N(0,1)), noise = -log(U + 1e-14), resident in HBM"},
           "roofline": topn, "roofline_topp": topp, "cases": table}
    if not args.no_cpu_baseline:
        threads, model_name, host_cpus = host_cpu()
        torch.set_num_threads(threads)
        sub = 256
        p_cpu, n_cpu = probs[:sub].cpu().unsqueeze(0), noise[:sub].cpu().unsqueeze(0)      # the oracle's (B, T, V) layout
        t0, done = time.time(), 0
        while done < 1 or time.time() - t0 < 10.0:
            for n in (1, 4, 16):
                OS.top_n(p_cpu, n)
            for n in (1, 4):
                OS.sample_without_replacement(p_cpu, n, n_cpu)
            for p in (0.5, 0.9):
                OS.sample_without_replacement(OS.top_p_filter(p_cpu, p), 4, n_cpu)
            done += 1
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": done * sub * n_cases / dt, "unit": "rows/s", "cores": threads, "kind": "port", "cpu_model": model_name,
                               "host_cpus": host_cpus, "sample": f"{done} pass(es) of the same sweep over the first {sub} rows (the reference's argsort path), {dt:.1f} s"}
    print(json.dumps(out))


# ------------------------------------------------------------------------------------------------------------------------------
def measure_steps(step, steps, warmup, fence, arena_optims):
    for i in range(warmup):
        step(i)
    for o in arena_optims:
        o.timing = []
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        d, g = step(warmup + i)
    fence()
    return time.perf_counter() - t0, d, g


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
    ap.add_argument("--no-fp32", action="store_true", help="skip the secondary fp32 timing of the same workload")
    ap.add_argument("--no-rooflines", action="store_true", help="timed steps only: no replay of the attention launch mix (a clean kernel trace under rocprofv3)")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a HIP graph")
    ap.add_argument("--dropout", type=float, default=0.0, help="attention / residual / embedding dropout rate of both models (the default "
                    "workload has none, like every parity run; stock GPT-2 configs carry 0.1)")
    ap.add_argument("--no-tuned-gemm", action="store_true", help="hipBLASLt default heuristics instead of the recorded TunableOp choices")
    args = ap.parse_args()
    if os.environ.get("BRANCHGAN_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BRANCHGAN_BENCH_WATCHDOG"]), exit=True)
        probe = os.path.join(ROOT, "benchmarks", "probes", "libsegvbt.so")      # native stacks of every thread 3 s earlier
        if os.path.exists(probe):
            import ctypes
            ctypes.CDLL(probe).install_all_thread_dump(int(float(os.environ["BRANCHGAN_BENCH_WATCHDOG"])) - 3)
    if args.workload == "c5":
        return sampling_sweep(args)

    from branchgan import configs, ops, training
    tuned = (not args.no_tuned_gemm) and training.use_tuned_gemms()
    rank, world, local = training.init_data_parallel()
    if world != args.gpus and rank == 0:
        print(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; reporting n_gpus={world}", file=sys.stderr)
    device = torch.device("cuda", local % torch.cuda.device_count() if world > 1 else 0)
    torch.cuda.set_device(device)
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    if args.dropout:
        wl["dropout"] = args.dropout
        wl["name"] += f", dropout {args.dropout}"
    V = configs.LMConfig.from_any(wl["gen"]).vocab_size
    B, S = wl["B"], wl["S"]
    n_batches = 16
    data = zipf_ids(n_batches * B, S, V, 1234 + rank).view(n_batches, B, S).to(device)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    launch_notes = {}

    def run(dtype_name, steps, warmup, with_rooflines, dropout=None):
        """Builds the models fresh and times `steps` steps; -> (elapsed s, final losses, roofline objects)."""
        use_amp = dtype_name == "bf16"
        model, conf, rl, g_opt, d_opt = build(wl if dropout is None else dict(wl, dropout=dropout), device)

        def eager_step(i):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
                return training.train_step(model, data[i % n_batches], conf, g_opt, d_opt, i, 9999, rl)
        graphed = None
        if args.eager:
            step = eager_step
        else:
            from branchgan.graphs import GraphedTrainStep
            failure = None
            try:
                graphed = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, torch.bfloat16 if use_amp else None)
            except Exception as exc:        # noqa: BLE001 - a capture that fails must not take the measurement with it
                failure = f"{type(exc).__name__}: {exc}"[:300]
                graphed = None
            if world > 1:                   # every rank runs the same form of the step, or their collectives do not meet
                flag = torch.tensor([1 if failure else 0], device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()) and graphed is not None:
                    failure, graphed = "another rank could not capture its step", None
            if graphed is None:
                print(f"[bench] rank {rank}: HIP-graph capture of the step failed ({failure}); timing the eager step instead", file=sys.stderr)
                launch_notes[dtype_name] = f"eager (graph capture failed: {failure})"
                torch.cuda.synchronize()
                step = eager_step
            else:
                step = lambda i: graphed.step(data[i % n_batches], i)
        arena_optims = [o.optimizer for o in (g_opt, d_opt) if hasattr(o.optimizer, "timing")]
        elapsed, d, g = measure_steps(step, steps, warmup, fence, arena_optims)
        if graphed is not None:
            graphed.check_nan()
        losses = {"gen": float(g["loss"].detach()), "disc": float(d["loss"].detach())}
        roofs = {}
        if with_rooflines:
            roofs["adam"] = adam_roofline(arena_optims)
        for o in arena_optims:
            o.timing = None
        del d, g
        if with_rooflines and rank == 0:
            ops.TIMING = {}
            rec = record_attention_launches(eager_step, warmup + steps)      # one eager step: every attention launch it makes
            roofs["eager_event_ms"] = {k: {"launches": v[0], "avg_ms": round(v[1], 4)} for k, v in sorted(ops.timing_summary().items())}
            ops.TIMING = None
            roofs["bwd"] = attention_roofline(rec, "bwd", dtype_name, device, args.workload)
            roofs["fwd"] = attention_roofline(rec, "fwd", dtype_name, device, args.workload)
            del rec
        elif with_rooflines and world > 1:
            eager_step(warmup + steps)       # keeps this rank's collectives in step with rank 0's recorded eager step
        return elapsed, losses, roofs

    main_dtype = "bf16" if args.dtype == "bf16" else "f32"
    elapsed, losses, roofs = run(main_dtype, args.steps, args.warmup, not args.no_rooflines)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    other = None
    if world == 1 and not args.no_fp32 and main_dtype == "bf16":      # the reference's shipped config is fp32: the same workload in fp32 beside it
        torch.cuda.empty_cache()
        n32 = max(3, args.steps // 2)
        e32, l32, _ = run("f32", n32, 2, False)
        other = {"dtype": "f32", "ms_per_step": 1000.0 * e32 / n32, "value": B * S * n32 / e32, "unit": "tokens/s", "steps": n32, "warmup": 2,
                 "final_losses": l32}
    stock = None
    if world == 1 and not args.no_fp32 and main_dtype == "bf16" and not args.dropout and wl["gen"].startswith("gpt2"):
        # The reference's GPT-2 modules apply F.dropout at the config's rates whatever the mode (quirk 1) and stock GPT-2 configs carry
        # 0.1; the Pythia configs of its training runs carry 0.  The headline runs without (like every parity run); the same workload
        # at the stock rates is timed beside it.
        torch.cuda.empty_cache()
        nd = max(3, args.steps // 2)
        try:
            ed, ld, _ = run("bf16", nd, 2, False, dropout=0.1)
            stock = {"dropout": 0.1, "ms_per_step": 1000.0 * ed / nd, "value": B * S * nd / ed, "unit": "tokens/s", "steps": nd, "warmup": 2,
                     "final_losses": ld}
        except Exception as exc:            # noqa: BLE001 - a secondary line must not take the headline with it
            stock = {"dropout": 0.1, "error": f"{type(exc).__name__}: {exc}"[:300]}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    use_amp = main_dtype == "bf16"
    ms = 1000.0 * elapsed / args.steps
    tokens = world * B * S * args.steps
    out = {
        "metric": "branch-GAN train tokens/sec (gen+disc fwd/bwd)", "value": tokens / elapsed, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": main_dtype, "data": "synthetic",
        "config": {"workload": wl["name"], "per_gpu_batch": B, "seq_len": S, "branch_points": wl["K"], "depth": wl["D"],
                   "generator": wl["gen"], "discriminator": wl["disc"], "dropout": args.dropout, "parallelism": f"dp{world}", "launch": "eager" if args.eager else launch_notes.get(main_dtype, "hipGraph replay of fwd+bwd; all-reduce and Adam eager"),
                   "precision": "bf16 autocast, fp32 master weights and accumulation" if use_amp else "fp32",
                   "weights": "random init", "gemm": "hipBLASLt, recorded TunableOp choices" if tuned else "hipBLASLt defaults", "final_losses": losses},
        # the dominant hand-written hot-path entry point: attention backward; the forward and the optimizer kernel beside it
        "roofline": roofs.get("bwd") or roofs.get("fwd") or roofs.get("adam"),
        "roofline_attention_fwd": roofs.get("fwd"),
        "roofline_adam": roofs.get("adam"),
        "eager_event_ms": roofs.get("eager_event_ms"),   # includes host launch gaps
    }
    if other is not None:
        out["fp32"] = other
    if stock is not None:
        out["stock_gpt2_dropout"] = stock
    if world == 1 and not args.no_rooflines:
        # the other north-star target (>= 70 % of HBM on sampling): BASELINE configs[4]'s sweep on resident 8192 x 50304 inputs,
        # after the timed steps (the models are gone by now; 3.3 GB of inputs, ~10 ms a pass)
        torch.cuda.empty_cache()
        c5 = WORKLOADS["c5"]
        table, topn, topp, *_ = sampling_rooflines(c5["rows"], c5["V"], device, 5, 2)
        out["roofline_sampling"], out["roofline_topp"], out["sampling_cases"] = topn, topp, table
        torch.cuda.empty_cache()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, args.cpu_batch)
    print(json.dumps(out))
    if os.environ.get("BRANCHGAN_TUNE_GEMMS"):           # training.use_tuned_gemms: the table + what this run timed
        import torch.cuda.tunable as tunable
        with open(os.environ["BRANCHGAN_TUNE_GEMMS"], "w") as f:
            for k, v in tunable.get_validators():
                f.write(f"Validator,{k},{v}\n")
            for r in tunable.get_results():
                f.write(",".join(str(x) for x in r) + "\n")
        print(f"tuned GEMM table: {len(tunable.get_results())} entries -> {os.environ['BRANCHGAN_TUNE_GEMMS']}", file=sys.stderr)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
