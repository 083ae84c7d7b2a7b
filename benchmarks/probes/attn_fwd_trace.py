#!/usr/bin/env python3
"""Where a wave of the key-split attention FORWARD spends its life: loads lib/libbranchgan_hip_trace.so (csrc `make trace`:
attn_fwd.hip with BG_ATTN_TRACE), replays recorded forward launches of one workload and prints, per launch class and shape,
the wall-clock (100 MHz) phase stamps every wave left - relative to the first wave's start.
usage: attn_fwd_trace.py [workload=c2]"""
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
wl = dict(bench.WORKLOADS[key])
dev = torch.device("cuda", 0)
model, conf, rl, g_opt, d_opt = bench.build(wl, dev)
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(2 * wl["B"], wl["S"], V, 1234).view(2, wl["B"], wl["S"]).to(dev)


def eager_step(i):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return training.train_step(model, data[i % 2], conf, g_opt, d_opt, i, 9999, rl)


eager_step(0)
rec = bench.record_attention_launches(eager_step, 1)
calls = [r for r in rec if r[0] == "fwd"]
handle = _capi.lib()
handle.bg_debug_attn_fwd_trace.restype = ctypes.c_int
handle.bg_debug_attn_fwd_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
WAVES = 8192
NAMES = ["start", "loads in", "Q staged", "text tiles", "branch keys", "merge barrier", "stored"]


def read():
    buf = np.zeros((WAVES, 16), dtype=np.uint64)
    assert handle.bg_debug_attn_fwd_trace(buf.ctypes.data, buf.nbytes) == 0
    return buf


seen = set()
for r in calls:
    sh = r[2]
    cls = bench._launch_class(sh)
    tag = (cls, sh.NQ, sh.K, sh.S_text, sh.max_path, sh.n_slots)
    if tag in seen:
        continue
    seen.add(tag)
    for _ in range(3):
        r[1]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        r[1]()
    e1.record()
    torch.cuda.synchronize()
    tr = read().astype(np.int64)
    R = sh.NQ * sh.K
    nwg = -(-R // 32) * sh.B * sh.H
    nw = min(nwg * 4, WAVES)
    t = tr[:nw]
    t0 = t[:, 0].min()
    print(f"== {cls}: NQ {sh.NQ} K {sh.K} S_text {sh.S_text} max_path {sh.max_path} slots {sh.n_slots} B*H {sh.B * sh.H} Dh {sh.Dh}: "
          f"{nwg} workgroups ({nw} waves traced), {e0.elapsed_time(e1) / 20 * 1000:.2f} us per eager launch back to back")
    for i, name in enumerate(NAMES):
        rel = (t[:, i] - t0) * 0.01          # us
        dur = (t[:, i] - t[:, i - 1]) * 0.01 if i else rel
        print(f"   {name:>14}: at  min {rel.min():6.2f}  med {np.median(rel):6.2f}  p90 {np.percentile(rel, 90):6.2f}  max {rel.max():6.2f} us"
              f"   | phase  med {np.median(dur):5.2f}  p90 {np.percentile(dur, 90):5.2f}  max {dur.max():5.2f} us")
