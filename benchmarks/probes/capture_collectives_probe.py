#!/usr/bin/env python3
"""Two questions about building a GraphedTrainStep under an RCCL process group (one-rank rehearsal, sharded form):

1. ProcessGroupNCCL's watchdog aborts the process when hipEventQuery of a collective's end event fails with
   hipErrorCapturedEvent during a capture.  WHICH collectives' events fail, i.e. on which stream were they enqueued?  Every Work
   handle the library creates is kept here with its call site and stream, and a thread polls `is_completed()` on all of them
   while the step is built (warm-up + captures); a failing query names its Work.
2. hipBLASLt prints "Hip error: 'operation would make the legacy stream depend on a capturing blocking stream'(906) at
   hipblaslt.cpp:172" once during the captures.  WHICH call, on which thread and stream?  stderr goes to a file; every dense
   GEMM entry point torch offers is wrapped and the file is looked at after each call.

usage (on the GPU box): BRANCHGAN_REHEARSE_RCCL=1 BRANCHGAN_DP_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
  MASTER_PORT=29577 python benchmarks/probes/capture_collectives_probe.py [workload=c1]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "branch-gan_amd")):
    sys.path.insert(0, p)
ERR = os.path.join(ROOT, "gpurun_out", "capture_probe_stderr.txt")
os.makedirs(os.path.dirname(ERR), exist_ok=True)
real_err = os.dup(2)
fd = os.open(ERR, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
os.dup2(fd, 2)


def say(*a):
    os.write(real_err, (" ".join(str(x) for x in a) + "\n").encode())


import torch
import torch.distributed as dist

import bench
from branchgan import graphs, training

works, lock = [], threading.Lock()


def keep(label, w):
    if w is not None and hasattr(w, "is_completed"):
        with lock:
            works.append((label, w))
    return w


def stream_name():
    return f"{torch.cuda.current_stream().cuda_stream:#x}"


for name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
    orig = getattr(dist, name)

    def wrapped(*a, _orig=orig, _name=name, **kw):
        w = _orig(*a, **kw)
        return keep(f"{_name} enqueued from stream {stream_name()} (async_op={kw.get('async_op')})", w)
    setattr(dist, name, wrapped)

seen_906 = [0]


def check_906(what):
    size = os.path.getsize(ERR)
    if size > seen_906[0]:
        new = open(ERR).read()[seen_906[0]:]
        seen_906[0] = size
        if "legacy stream" in new:
            say(f"[906] first seen after {what}; thread {threading.current_thread().name}; current stream {stream_name()}; "
                f"capturing {torch.cuda.is_current_stream_capturing()}")


for name in ("addmm", "baddbmm", "mm", "bmm", "matmul"):
    orig = getattr(torch, name)

    def gemm(*a, _orig=orig, _name=name, **kw):
        out = _orig(*a, **kw)
        check_906(f"torch.{_name}{tuple(tuple(t.shape) for t in a if isinstance(t, torch.Tensor))}")
        return out
    setattr(torch, name, gemm)
orig_linear = torch.nn.functional.linear


def linear(*a, **kw):
    out = orig_linear(*a, **kw)
    check_906(f"F.linear{tuple(tuple(t.shape) for t in a if isinstance(t, torch.Tensor))}")
    return out


torch.nn.functional.linear = linear

stop = threading.Event()
failed = {}


def poll():
    while not stop.is_set():
        with lock:
            items = list(works)
        for label, w in items:
            try:
                w.is_completed()
            except Exception as exc:       # noqa: BLE001
                failed.setdefault(label, str(exc).splitlines()[0][:140])
        time.sleep(0.005)


wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c1"])
rank, world, local = training.init_data_parallel()
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
model, conf, rl, g_opt, d_opt = bench.build(wl, device)
from branchgan import configs
V = configs.LMConfig.from_any(wl["gen"]).vocab_size
data = bench.zipf_ids(4 * wl["B"], wl["S"], V, 1234).view(4, wl["B"], wl["S"]).to(device)
say(f"main thread current stream {stream_name()}; arena streams are created at the first reduction")
# the bounded sleep stays ON: it keeps the real watchdog out of the way while OUR handles are polled through the captures
t = threading.Thread(target=poll, name="poller")
t.start()
in_capture = []
orig_new_graph = graphs.new_graph


def new_graph():
    with lock:
        in_capture.append(len(works))
    return orig_new_graph()


graphs.new_graph = new_graph
step = graphs.GraphedTrainStep(model, conf, g_opt, d_opt, data[0], 9999, rl, torch.bfloat16)
stop.set()
t.join()
say(f"side (capture) stream {step.stream.cuda_stream:#x}; generator arena stream {g_opt.arena.stream.cuda_stream:#x}; "
    f"discriminator arena stream {d_opt.arena.stream.cuda_stream:#x}")
say(f"{len(works)} Work handles were created by the warm-up; {in_capture[0] if in_capture else '?'} before the first capture began")
by_label = {}
for label, _ in works:
    by_label[label] = by_label.get(label, 0) + 1
for label, n in sorted(by_label.items()):
    say(f"  {n:3d} x {label}: {'QUERY FAILED: ' + failed[label] if label in failed else 'queries ok throughout'}")
for i in range(3):
    step.step(data[i % 4], i)
step.check_nan()
torch.cuda.synchronize()
say("three graphed steps ran; stderr of the run:", ERR)
os.dup2(real_err, 2)
for line in open(ERR):
    if "Hip error" in line or "legacy" in line:
        say("[stderr]", line.strip()[:300])
dist.destroy_process_group()
