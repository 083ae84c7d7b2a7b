#!/usr/bin/env python3
"""Probes for the three hipGraph anomalies of round 1 ("What's weak" 1-3).  Each sub-command is one small
experiment that prints JSON lines; run one per process (`python benchmarks/probes/graph_anomalies.py <name>`).

  memset     : a captured hipMemsetAsync of N bytes followed by an increment kernel, replayed 3x: does the memset node run?
  atensum    : ATen's column sum (the bias gradient of the module-by-module path: a multi-block reduction with a semaphore
               buffer it clears with hipMemsetAsync) captured and replayed against its eager value
  unfused    : the whole unfused C2 step captured (BRANCHGAN_FUSED=0), replayed 3x with identical inputs and no optimizer step:
               which gradient tensors differ from replay 1 / hold NaNs
  segv-<var> : capture 20 backward calls of softmax_stats (the round-1 capture_end crash) in variants
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "branch-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch


def out(**kw):
    print(json.dumps(kw), flush=True)


def memset():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    side = torch.cuda.Stream()
    for nbytes in (4, 16, 64, 96, 128, 252, 256, 1024, 4096):
        buf = torch.zeros(max(nbytes // 4, 1), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            buf.add_(1)
        seen = []
        for _ in range(3):
            g.replay()
            torch.cuda.synchronize()
            seen.append(sorted(set(buf.tolist())))
        out(probe="memset", nbytes=nbytes, rc=rc, values_after_each_replay=seen, ok=all(s == [1] for s in seen))


def _fix(g):
    """keep_graph=True capture -> replace memset nodes -> instantiate; returns (census before, replaced)."""
    from branchgan import _capi as C
    raw = ctypes.c_void_p(g.raw_cuda_graph())
    c = [ctypes.c_int() for _ in range(4)]
    C.check(C.lib().bg_graph_node_census(raw, *[ctypes.byref(x) for x in c]), "census")
    n = ctypes.c_int()
    C.check(C.lib().bg_graph_replace_memset_nodes(raw, ctypes.byref(n)), "replace")
    g.instantiate()
    return [x.value for x in c], n.value


def memsetfix():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    side = torch.cuda.Stream()
    for nbytes, value in ((16, 0), (96, 0), (252, 0), (4096, 0), (1024, 7)):
        buf = torch.zeros(max(nbytes // 4, 1) + 4, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g, stream=side):
            buf.add_(5)
            hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), value, nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            buf.add_(1)
        census, replaced = _fix(g)
        seen = []
        for _ in range(3):
            g.replay()
            torch.cuda.synchronize()
            seen.append((sorted(set(buf[:nbytes // 4].tolist())), buf[nbytes // 4:].tolist()))
        want = value * 0x01010101 + 1
        out(probe="memsetfix", nbytes=nbytes, census=census, replaced=replaced, after_each_replay=seen,
            ok=all(s[0] == [want] for s in seen) and [s[1][0] for s in seen] == [6, 12, 18])
    # ATen's multi-block column sum, the case that fails unfixed
    x = torch.randn(3072, 3072, device="cuda").bfloat16()
    with torch.cuda.stream(side):
        x.sum(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=side):
        y = x.sum(0)
    census, replaced = _fix(g)
    res = []
    for r in range(4):
        x.copy_(torch.randn(3072, 3072, device="cuda").bfloat16())
        g.replay()
        torch.cuda.synchronize()
        res.append(int((~torch.isclose(y.float(), x.sum(0).float(), rtol=2e-2, atol=2e-1)).sum()))
    out(probe="memsetfix-atensum", census=census, replaced=replaced, mismatched_per_replay=res, ok=not any(res))


def atensum():
    side = torch.cuda.Stream()
    for rows, cols, dt in ((3072, 768, torch.bfloat16), (2048, 768, torch.bfloat16), (3072, 3072, torch.bfloat16), (3072, 768, torch.float32),
                           (256, 768, torch.bfloat16)):
        x = torch.randn(rows, cols, device="cuda").to(dt)
        with torch.cuda.stream(side):
            for _ in range(2):
                x.sum(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            junk = torch.empty(1 << 20, device="cuda")       # something that frees memory into the pool before the reduction
            junk.fill_(float("nan"))
            del junk
            y = x.sum(0)
        res = []
        for r in range(4):
            x.copy_(torch.randn(rows, cols, device="cuda").to(dt))
            g.replay()
            torch.cuda.synchronize()
            ref = x.sum(0)
            bad = int((~torch.isclose(y.float(), ref.float(), rtol=2e-2, atol=2e-1)).sum())
            res.append(dict(replay=r, mismatched=bad, nans=int(torch.isnan(y).sum())))
        out(probe="atensum", rows=rows, cols=cols, dtype=str(dt), replays=res)


def unfused():
    os.environ["BRANCHGAN_FUSED"] = os.environ.get("BRANCHGAN_FUSED", "0")
    os.environ["BRANCHGAN_GRAPH_UNFUSED"] = "1"
    import bench
    from branchgan.graphs import GraphedTrainStep
    dev = torch.device("cuda", 0)
    wl = dict(bench.WORKLOADS[os.environ.get("PROBE_WORKLOAD", "c2")])
    model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
    from branchgan import configs
    V = configs.LMConfig.from_any(wl["gen"]).vocab_size
    data = bench.zipf_ids(2 * wl["B"], wl["S"], V, 1234).view(2, wl["B"], wl["S"]).to(dev)
    amp = torch.bfloat16 if os.environ.get("PROBE_DTYPE", "bf16") == "bf16" else None
    gs = GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, amp, warmup=1)
    names = {}
    for tag, opt, mod in (("G", g_opt, model.generator), ("D", d_opt, model.discriminator)):
        by_id = {id(p): n for n, p in mod.named_parameters()}
        names[tag] = [(by_id[id(p)], off, p.numel()) for p, off in zip(opt.arena.params, opt.arena.offsets)]
    first = {}
    for r in range(4):
        gs.static_ids.copy_(data[1])
        g_opt.arena.zero()
        d_opt.arena.zero()
        gs.replay_graphs()
        torch.cuda.synchronize()
        for tag, opt in (("G", g_opt), ("D", d_opt)):
            a = opt.arena.arena.clone()
            if r == 0:
                first[tag] = a
            bad = []
            for n, off, cnt in names[tag]:
                cur, ref = a[off:off + cnt], first[tag][off:off + cnt]
                nn = int(torch.isnan(cur).sum())
                nd = int((cur != ref).sum()) if r else 0
                if nn or nd:
                    cols = torch.nonzero(torch.isnan(cur) | (cur != ref)).flatten()[:6].tolist()
                    bad.append(dict(param=n, numel=cnt, nans=nn, differs_from_replay0=nd, first_idx=cols))
            out(probe="unfused", replay=r, model=tag, losses=[float(gs.losses[0]["loss"]), float(gs.losses[1]["loss"])],
                n_bad_params=len(bad), bad=bad[:12])


def segv(variant):
    from branchgan import ops, _capi as C
    rows = 256
    V = 50304 if variant == "alignedV" else 50257
    dt = torch.float32 if variant == "fp32" else torch.bfloat16
    reps = 2 if variant == "reps2" else 20
    g0 = torch.Generator(device="cuda").manual_seed(2)
    logits = 3.0 * torch.randn(rows, V, device="cuda", generator=g0)
    ids = torch.randint(0, V, (rows, 2), device="cuda", generator=g0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    mode = "thread_local" if variant == "threadlocal" else "global"
    x_nonleaf = None
    if variant == "nonleaf":      # x = a copy made on the DEFAULT stream of a tensor that requires grad: its grad_fn (ToCopyBackward) lives there
        x_nonleaf = logits.requires_grad_(True).to(dt)
        torch.cuda.synchronize()
    with torch.cuda.stream(side):
        x = x_nonleaf if x_nonleaf is not None else (torch.randn(rows, V, device="cuda").to(dt) if variant == "freshx" else logits.to(dt)).requires_grad_(True)
        if variant == "capi":                 # the kernel alone, no autograd, output preallocated
            lse = torch.empty(rows, device="cuda")
            lp = torch.empty(rows, 2, device="cuda")
            dx = torch.empty_like(x)
            gl, gp = torch.randn(rows, device="cuda"), torch.randn(rows, 2, device="cuda")
            xd = x.detach()
            C.check(C.lib().bg_softmax_stats_fwd(C.ptr(xd), C.dtype_code(xd), rows, V, xd.stride(0), C.ptr(ids), 2, C.ptr(lse), C.ptr(lp), C.stream_ptr()), "fwd")
            fn = lambda: C.check(C.lib().bg_softmax_stats_bwd(C.ptr(xd), C.dtype_code(xd), rows, V, xd.stride(0), C.ptr(ids), 2, C.ptr(lse),
                                                              C.ptr(gl), C.ptr(gp), C.ptr(dx), C.stream_ptr()), "bwd")
        else:
            lse, lp = ops.softmax_stats(x, ids)
            gl, gp = torch.randn_like(lse), torch.randn_like(lp)
            fn = lambda: torch.autograd.grad((lse, lp), x, (gl, gp), retain_graph=True)
        for _ in range(3):
            fn()
    side.synchronize()
    out(probe="segv", variant=variant, stage="before capture")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode=mode):
        for _ in range(reps):
            fn()
    out(probe="segv", variant=variant, stage="captured")
    g.replay()
    torch.cuda.synchronize()
    out(probe="segv", variant=variant, stage="replayed", ok=True)


if __name__ == "__main__":
    import faulthandler
    faulthandler.enable()
    what = sys.argv[1]
    if what.startswith("segv-"):
        segv(what[5:])
    else:
        globals()[what]()
