#!/usr/bin/env python3
"""Phase times of the top-p kernel per row (lib/libbranchgan_hip_trace.so, csrc `make trace`): the kernel_sweep inputs
(softmax of 3 * randn logits, 8192 x 50304, p = 0.5 and 0.9), stamps of thread 0 of every row's workgroup.
usage: topp_trace.py [rows=8192] [V=50304]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
import torch

from branchgan import _capi

_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libbranchgan_hip_trace.so")
from branchgan import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
V = int(sys.argv[2]) if len(sys.argv) > 2 else 50304
g = torch.Generator(device="cuda").manual_seed(2)
logits = 3.0 * torch.randn(rows, V, device="cuda", generator=g)
probs = torch.softmax(logits, -1)
noise = -torch.log(torch.rand(rows, V, device="cuda", generator=g) + 1e-14)
blocked = torch.randint(0, V, (rows, 1), device="cuda", generator=g)
h = _capi.lib()
h.bg_debug_topp_trace.restype = ctypes.c_int
h.bg_debug_topp_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
FIRST = os.environ.get("BRANCHGAN_TOPP_FIRST_FORM") == "1"
NAMES = (["row in registers", "level-1 histogram", "level-1 scan", "level-2 histogram", "level-2 scan", "level-3 histogram",
          "level-3 scan", "count / fit / cut", "noise pass + lists", "merge + store"] if FIRST else
         ["row in registers", "level-1 histogram", "level-1 scan", "candidates listed", "levels + ranked list", None, None, "fit / theta",
          "kept to front, keys, noise row", "merge + store"])
for p in (0.5, 0.9):
    for _ in range(2):
        ops.topp_select(probs, 4, noise, p, blocked=blocked)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.topp_select(probs, 4, noise, p, blocked=blocked)
    e.record()
    torch.cuda.synchronize()
    buf = np.zeros((8192, 12), dtype=np.uint64)
    assert h.bg_debug_topp_trace(buf.ctypes.data, buf.nbytes) == 0
    t = buf[:min(rows, 8192)].astype(np.int64)
    life = (t[:, 10] - t[:, 0]) * 0.01
    print(f"== top_p {p}: kernel {s.elapsed_time(e):.3f} ms for {rows} rows; a row's workgroup lives {np.median(life):.1f} us (p90 {np.percentile(life, 90):.1f}); "
          f"{rows} rows / 256 CUs x that = {rows / 256 * np.median(life) / 1000:.2f} ms")
    prev = 0
    for i, name in enumerate(NAMES):
        if name is None:
            continue
        d = (t[:, i + 1] - t[:, prev]) * 0.01
        prev = i + 1
        print(f"   {name:>20}: med {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
