#!/usr/bin/env python3
"""How does a CU-masked stream (bg_stream_create_cu_mask) behave on this part?  (1) bg_adam_step on 86 M parameters (the
discriminator's arena at configs[1], 2.9 GB of traffic) under masks of 256 / 128 / 64 / 32 CUs, the set bits either the LOWEST
ones or every k-th: which layout spreads over the XCDs shows in the bandwidth; (2) the same Adam launch beside a chain of
small kernels (256-row LayerNorms + GEMMs, the shape of a generator branch step) on the default stream: how much does each side
lose, masked vs unmasked."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
from branchgan import _capi as C  # noqa: E402

dev = torch.device("cuda", 0)
N = 86_000_000
p, grad = torch.randn(N, device=dev), torch.randn(N, device=dev)
m, v, lp = torch.zeros_like(p), torch.zeros_like(p), torch.empty(N, dtype=torch.bfloat16, device=dev)
_step = [0]


class opt:                      # bg_adam_step on the current stream
    @staticmethod
    def step():
        _step[0] += 1
        C.check(C.lib().bg_adam_step(C.ptr(p), C.ptr(grad), C.ptr(m), C.ptr(v), C.ptr(lp), C.BG_BF16, N, 1e-4, 0.9, 0.999, 1e-8, 0.0,
                                     _step[0], 1.0, 1, C.stream_ptr()), "bg_adam_step")


hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """hipExtStreamCreateWithCUMask straight from the HIP runtime (the library has no entry point for it: the schedule this probe
    was written for was measured and rejected, profiles/r3_q_cu_mask_probe.txt)."""
    words = (ctypes.c_uint32 * 8)(*[0] * 8)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    out = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(out), ctypes.c_uint32(8), words)
    if rc != 0:
        hip.hipGetLastError()
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: error {rc}")
    st = torch.cuda.ExternalStream(out.value, dev)
    st.raw = out.value          # a masked stream owns a hardware queue: released after use
    return st


def release(st):
    st.synchronize()
    hip.hipStreamDestroy(ctypes.c_void_p(st.raw))


def time_adam(stream, reps=5):
    with torch.cuda.stream(stream):
        opt.step()
        stream.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        for _ in range(reps):
            opt.step()
        t1.record(stream)
        stream.synchronize()
    return t0.elapsed_time(t1) / reps


print(f"unmasked stream: {time_adam(torch.cuda.Stream()):.3f} ms per Adam launch ({N * 34 / 1e9:.2f} GB)")
def timed_under(bits):
    try:
        st = masked_stream(bits)
    except RuntimeError as e:
        return None
    t = time_adam(st)
    release(st)
    return t


PATTERNS = {
    "lowest 128": range(128), "lowest 64": range(64), "lowest 32": range(32),
    "every 4th bit (64)": range(0, 256, 4),
    "bits 64-127 (64)": range(64, 128), "bits 192-255 (64)": range(192, 256),
}
for name, bits in PATTERNS.items():
    t = timed_under(bits)
    print(f"{name:36s}: " + ("rejected" if t is None else f"{t:.3f} ms per Adam launch = {N * 34 / t / 1e9:.2f} TB/s"), flush=True)

# (2) beside a chain of small kernels
x = torch.randn(256, 768, device=dev, dtype=torch.bfloat16)
w1 = torch.randn(768, 3072, device=dev, dtype=torch.bfloat16)
w2 = torch.randn(3072, 768, device=dev, dtype=torch.bfloat16)
ln = torch.nn.LayerNorm(768).to(dev).bfloat16()


def chain(n=200):
    y = x
    for _ in range(n):
        y = ln(y)
        y = torch.nn.functional.gelu(y @ w1) @ w2
    return y


g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    chain(3)
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        chain()


def both(adam_stream):
    torch.cuda.synchronize()
    t0, t1, a0, a1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(side):
        t0.record(side)
        g.replay()
        t1.record(side)
    if adam_stream is not None:
        with torch.cuda.stream(adam_stream):
            a0.record(adam_stream)
            for _ in range(3):
                opt.step()
            a1.record(adam_stream)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1), (a0.elapsed_time(a1) / 3 if adam_stream is not None else 0.0)


print(f"chain of 800 small kernels alone: {both(None)[0]:.3f} ms")
c, a = both(torch.cuda.Stream())
print(f"chain beside 3 Adam launches on an unmasked stream: chain {c:.3f} ms, Adam {a:.3f} ms each")
for name, bits in (("lowest 64 CUs", range(64)), ("lowest 32 CUs", range(32)), ("lowest 128 CUs", range(128)), ("CUs 192-255", range(192, 256))):
    try:
        st = masked_stream(bits)
    except RuntimeError:
        print(name, "rejected")
        continue
    c, a = both(st)
    release(st)
    print(f"chain beside 3 Adam launches on a {name} stream: chain {c:.3f} ms, Adam {a:.3f} ms each")

# (3) complementary masks: the chain captured and replayed on a stream of CUs 64-255 (or 32-255), Adam on CUs 0-63 (0-31)
for n_adam in (64, 32, 96):
    chain_stream = masked_stream(range(n_adam, 256))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(chain_stream):
        chain(3)
        chain_stream.synchronize()
        with torch.cuda.graph(g2, stream=chain_stream):
            chain()
    adam_stream = masked_stream(range(n_adam))

    def both2(with_adam):
        torch.cuda.synchronize()
        t0, t1, a0, a1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(chain_stream):
            t0.record(chain_stream)
            g2.replay()
            t1.record(chain_stream)
        if with_adam:
            with torch.cuda.stream(adam_stream):
                a0.record(adam_stream)
                for _ in range(3):
                    opt.step()
                a1.record(adam_stream)
        torch.cuda.synchronize()
        return t0.elapsed_time(t1), (a0.elapsed_time(a1) / 3 if with_adam else 0.0)

    alone = both2(False)[0]
    c, a = both2(True)
    print(f"chain on CUs {n_adam}-255: alone {alone:.3f} ms; beside 3 Adam launches on CUs 0-{n_adam - 1}: chain {c:.3f} ms, Adam {a:.3f} ms each", flush=True)
    del g2
    release(adam_stream)
    release(chain_stream)
