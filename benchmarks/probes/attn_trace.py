#!/usr/bin/env python3
"""Where a wave of the split attention backward spends its life: loads lib/libbranchgan_hip_trace.so (csrc `make trace`:
attn_bwd_split.hip with BG_ATTN_TRACE), replays recorded backward launches of one workload and prints, per kernel and per
launch class, the wall-clock (100 MHz) phase stamps every wave left - relative to the first wave's start.
usage: attn_trace.py [workload=c2] [dtype=bf16|fp32]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

from branchgan import _capi

_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libbranchgan_hip_trace.so")
import bench
from branchgan import configs, training

key = sys.argv[1] if len(sys.argv) > 1 else "c2"
use_amp = (sys.argv[2] if len(sys.argv) > 2 else "bf16") == "bf16"
wl = dict(bench.WORKLOADS[key])
dev = torch.device("cuda", 0)
model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(2 * wl["B"], wl["S"], V, 1234).view(2, wl["B"], wl["S"]).to(dev)


def eager_step(i):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
        return training.train_step(model, data[i % 2], conf, g_opt, d_opt, i, 9999, rl)


eager_step(0)
rec = bench.record_attention_launches(eager_step, 1)
calls = [r for r in rec if r[0] == "bwd"]
handle = _capi.lib()
handle.bg_debug_attn_trace.restype = ctypes.c_int
handle.bg_debug_attn_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
WAVES = 8192
NAMES = {0: ["start", "ctx/ntiles", "loads in", "text loop", "slot loop", "barrier", "stored"],
         1: ["start", "K/V frags+reach", "barrier", "first gather", "tile loop", "dK stored", "dV stored"]}


def read():
    buf = np.zeros((2, WAVES, 16), dtype=np.uint64)
    assert handle.bg_debug_attn_trace(buf.ctypes.data, buf.nbytes) == 0
    return buf


def clear():
    # the trace array keeps the previous launch's stamps: wipe by reading after a launch only the waves this grid has
    pass


seen = set()
for r in calls:
    cls = bench._launch_class(r[2])
    sh = r[2]
    tag = (cls, sh.NQ, sh.K, sh.S_text, sh.max_path)
    if tag in seen:
        continue
    seen.add(tag)
    for _ in range(3):
        r[1]()
    torch.cuda.synchronize()
    tr = read().astype(np.int64)
    R = sh.NQ * sh.K
    SUB = ["staged", "S, dP", "exp", "grad MFMAs"]
    for kind, nwg in ((0, -(-R // 32) * sh.B * sh.H), (1, -(-sh.S_text // 32) * sh.B * sh.H)):
        nw = min(nwg * 4, WAVES)
        if nw == 0:
            continue
        t = tr[kind, :nw]
        t0 = t[:, 0].min()
        print(f"== {cls}: NQ {sh.NQ} K {sh.K} S_text {sh.S_text} max_path {sh.max_path} B*H {sh.B * sh.H} Dh {sh.Dh} | "
              f"{'dq' if kind == 0 else 'dkv'} split: {nwg} workgroups ({nw} waves traced), tiles/wave mean {t[:, 7].mean():.2f} max {t[:, 7].max()}")
        for i, name in enumerate(NAMES[kind]):
            rel = (t[:, i] - t0) * 0.01          # us
            dur = (t[:, i] - t[:, i - 1]) * 0.01 if i else rel
            print(f"   {name:>16}: at  min {rel.min():6.2f}  med {np.median(rel):6.2f}  p90 {np.percentile(rel, 90):6.2f}  max {rel.max():6.2f} us"
                  f"   | phase  med {np.median(dur):5.2f}  p90 {np.percentile(dur, 90):5.2f}  max {dur.max():5.2f} us")
        has = t[:, 7] > 0                      # first tile of the waves that had one
        first = 2 if kind == 0 else 3
        prev = t[has, first]
        for i, name in enumerate(SUB):
            dur = (t[has, 8 + i] - prev) * 0.01
            prev = t[has, 8 + i]
            print(f"      first tile {name:>12}: phase  med {np.median(dur):5.2f}  p90 {np.percentile(dur, 90):5.2f}  max {dur.max():5.2f} us")
