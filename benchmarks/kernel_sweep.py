#!/usr/bin/env python3
"""Kernel-level sweep (BASELINE configs[4]: branch sampling at batch 64 x 128 rows over a 50k vocabulary, plus the attention
shapes of configs[1]/[3]): achieved GB/s and TFLOP/s of the hand-written kernels against MI355X peaks (8 TB/s HBM spec, 6.3 TB/s
achievable copy rate; 2.5 PFLOP/s dense bf16).  Algorithmic bytes/FLOPs are the SURVEY.md §8(d) figures.  Prints one JSON line
per case."""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import numpy as np
import torch

from branchgan import ops

HBM, MFMA_BF16, MFMA_F32 = 8000.0, 2500.0, 157.3


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def timeit_graphed(make_fn, reps=20, iters=10):
    """Kernels of a few microseconds: capture `reps` back-to-back calls in a HIP graph and time replays, so that the host's
    launch cost (tens of microseconds through Python + ctypes) is not what is measured.  `make_fn()` runs on the capture
    stream and returns the callable to time (autograd runs a backward node on its forward's stream, so the forward that a
    timed backward belongs to has to be made there too)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn = make_fn()
        for _ in range(3):
            fn()
    side.synchronize()
    timing, ops.TIMING = ops.TIMING, None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    ops.TIMING = timing
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps)


def report(name, ms, nbytes=None, flops=None, **kw):
    out = {"case": name, "ms": round(ms, 4)}
    if nbytes:
        out.update(GBps=round(nbytes / ms / 1e6, 1), hbm_frac=round(nbytes / ms / 1e6 / HBM, 3))
    if flops:
        out.update(TFLOPs=round(flops / ms / 1e9, 2))
    out.update(kw)
    print(json.dumps(out), flush=True)


def sampling(rows, V):
    small = rows <= 1024
    timeit = (lambda f, **kw: timeit_graphed(lambda: f)) if small else globals()['timeit']
    g = torch.Generator(device="cuda").manual_seed(2)
    logits = 3.0 * torch.randn(rows, V, device="cuda", generator=g)
    probs = torch.softmax(logits, -1)
    noise = -torch.log(torch.rand(rows, V, device="cuda", generator=g) + 1e-14)
    blocked = torch.randint(0, V, (rows, 1), device="cuda", generator=g)
    e = 4
    for n in (1, 4, 16):
        ms = timeit(lambda: ops.topn_select(probs, n, blocked=blocked))
        report(f"TOP_K n={n} rows={rows} V={V} fp32", ms, rows * V * e + rows * n * 12)
    pb = probs.bfloat16()
    ms = timeit(lambda: ops.topn_select(pb, 4, blocked=blocked))
    report(f"TOP_K n=4 rows={rows} V={V} bf16", ms, rows * V * 2 + rows * 4 * 12)
    for n in (1, 4):
        ms = timeit(lambda: ops.topn_select(probs, n, noise=noise, blocked=blocked))
        report(f"SAMPLE_NO_REPLACEMENT n={n} rows={rows} V={V}", ms, 2 * rows * V * e + rows * n * 12)
    for p in (0.5, 0.9):
        ms = timeit(lambda: ops.topp_select(probs, 4, noise, p, blocked=blocked))
        report(f"TOP_P p={p} n=4 rows={rows} V={V}", ms, 2 * rows * V * e + rows * 4 * 12)
    ids = torch.randint(0, V, (rows, 2), device="cuda", generator=g)
    for dt, es in ((torch.float32, 4), (torch.bfloat16, 2)):
        # a LEAF of its own: `logits.to(float32)` is logits itself, so requires_grad_ on it would make the bf16 copy of the next
        # iteration a non-leaf whose grad_fn lives on the default stream - autograd then pulls the null stream into the capture
        # and ROCm 7.2's hipStreamEndCapture segfaults on it (benchmarks/probes/graph_anomalies.py segv-nonleaf)
        x = logits.detach().to(dt, copy=True).requires_grad_(True)
        ms = timeit(lambda: ops.softmax_stats(x.detach(), ids))
        report(f"softmax_stats_fwd rows={rows} V={V} {dt}", ms, rows * V * es)
        def make_bwd(x=x):
            lse, lp = ops.softmax_stats(x, ids)
            gl, gp = torch.randn_like(lse), torch.randn_like(lp)
            return lambda: torch.autograd.grad((lse, lp), x, (gl, gp), retain_graph=True)
        ms = timeit_graphed(make_bwd) if small else globals()['timeit'](make_bwd())
        report(f"softmax_stats_bwd rows={rows} V={V} {dt}", ms, 2 * rows * V * es)
    # evaluation metrics: one read of the probabilities per row (the reference: three full sorts of the same tensor)
    labels = torch.randint(0, V, (rows,), device="cuda", generator=g)
    for ranks in ((), (1, 8, 16, 64, 128, 256, 512, 1024)):
        ms = timeit(lambda: ops.rank_stats(probs, labels, ranks))
        report(f"rank_stats ranks={len(ranks)} rows={rows} V={V}", ms, rows * V * e)
    # reference op chain on the same GPU for scale (torch.argsort over V = what _topKSampling does)
    sub = probs[:1024]
    ms = globals()['timeit'](lambda: torch.argsort(sub, dim=-1, descending=True)[:, :4], iters=3, warmup=1)
    report(f"[torch] argsort-based top-k rows=1024 V={V} (scaled x{rows // 1024} for {rows} rows)", ms * rows / 1024, rows * V * e)


def attention(B, H, Dh, S, K, depth, dtype):
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g).to(dtype).requires_grad_(True)
    sparse = np.sort(np.random.RandomState(1234).choice(S - 1, K, replace=False))
    ctx = torch.tensor(sparse + 1, dtype=torch.int32, device="cuda")
    e = 2 if dtype == torch.bfloat16 else 4
    # one generator branch step at depth `depth` (NQ = 1), and the one-pass discriminator over all `depth` nodes
    for label, NQ in (("gen step", 1), ("disc one-pass", depth)):
        q = mk(B, H, NQ, K, Dh)
        kt, vt = mk(B, H, S, Dh), mk(B, H, S, Dh)
        n_slots = depth if NQ == 1 else depth
        ks, vs = [mk(B, H, K, Dh) for _ in range(n_slots)], [mk(B, H, K, Dh) for _ in range(n_slots)]
        paths = [list(range(depth))] if NQ == 1 else [list(range(i + 1)) for i in range(depth)]
        pairs = sum(int(c) for c in ctx.tolist()) * NQ + K * sum(len(p) for p in paths)
        flops = 4 * Dh * H * B * pairs
        nbytes = (2 * B * H * Dh * (int(ctx.max()) + n_slots * K) + 2 * B * H * NQ * K * Dh) * e
        f = lambda: ops.branch_attention(q, kt, vt, ctx, ks, vs, paths=paths)[0]
        ms = timeit_graphed(lambda: f)
        report(f"branch_attn fwd {label} B{B} H{H} Dh{Dh} S{S} K{K} d{depth} {dtype}", ms, nbytes, flops, mfma_frac=round(flops / ms / 1e9 / (MFMA_BF16 if e == 2 else MFMA_F32), 4))
        def make_bwd():
            out = f()
            go = torch.randn_like(out)
            return lambda: torch.autograd.grad(out, [q, kt, vt] + ks + vs, go, retain_graph=True)
        ms = timeit_graphed(make_bwd)
        report(f"branch_attn bwd {label} B{B} H{H} Dh{Dh} S{S} K{K} d{depth} {dtype}", ms, 2.5 * nbytes, 2.5 * flops)
    # root causal pass
    q, k, v = mk(B, H, S, Dh), mk(B, H, S, Dh), mk(B, H, S, Dh)
    flops = 4 * Dh * H * B * S * (S + 1) // 2
    ms = timeit_graphed(lambda: (lambda: ops.causal_attention(q, k, v)[0]))
    report(f"causal_attn fwd B{B} H{H} Dh{Dh} S{S} {dtype}", ms, 4 * B * H * S * Dh * e, flops, mfma_frac=round(flops / ms / 1e9 / (MFMA_BF16 if e == 2 else MFMA_F32), 4))
    def make_cbwd():
        out = ops.causal_attention(q, k, v)[0]
        go = torch.randn_like(out)
        return lambda: torch.autograd.grad(out, [q, k, v], go, retain_graph=True)
    ms = timeit_graphed(make_cbwd)
    report(f"causal_attn bwd B{B} H{H} Dh{Dh} S{S} {dtype}", ms, 10 * B * H * S * Dh * e, 2.5 * flops)


if __name__ == "__main__":
    if os.environ.get("BRANCHGAN_SEGV_BT"):      # probe: native backtrace on SIGSEGV (benchmarks/probes/segv_backtrace.cpp)
        import ctypes
        ctypes.CDLL(os.path.join(ROOT, "benchmarks", "probes", "libsegvbt.so")).install_segv_backtrace()
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all", choices=["all", "sampling", "attention", "metrics"])
    ap.add_argument("--rows", type=int, default=8192)
    a = ap.parse_args()
    if a.what in ("all", "sampling"):
        if not os.environ.get("BRANCHGAN_SWEEP_SMALL_ONLY"):
            sampling(a.rows, 50304)
        sampling(256, 50257)
    if a.what == "metrics":
        g = torch.Generator(device="cuda").manual_seed(2)
        for rows, V in ((a.rows, 50304), (2048, 50257)):
            probs = torch.softmax(3.0 * torch.randn(rows, V, device="cuda", generator=g), -1)
            labels = torch.randint(0, V, (rows,), device="cuda", generator=g)
            for ranks in ((), (1,), (1, 8, 16, 64, 128, 256, 512, 1024)):
                ms = timeit(lambda: ops.rank_stats(probs, labels, ranks))
                report(f"rank_stats ranks={len(ranks)} rows={rows} V={V}", ms, rows * V * 4)
            sub = probs[:512]
            ms = timeit(lambda: torch.sort(sub, dim=-1, descending=True), iters=3, warmup=1)
            report(f"[torch] one full sort, rows=512 scaled to {rows} (the reference runs three)", ms * rows / 512, rows * V * 4)
    if a.what in ("all", "attention"):
        attention(8, 12, 64, 256, 32, 4, torch.bfloat16)      # configs[1]
        attention(8, 12, 64, 256, 32, 4, torch.float32)
        attention(4, 8, 256, 512, 64, 8, torch.bfloat16)      # configs[3]
        attention(8, 16, 64, 128, 32, 8, torch.bfloat16)      # configs[2]
