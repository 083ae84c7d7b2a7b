"""Which hipEventQuery calls fail while ANOTHER thread captures a stream?  (ProcessGroupNCCL's watchdog thread polls the end events of
finished collectives; an exception there aborts the process.)  Cases: an event recorded eagerly before the capture on (a) another
stream, (b) the stream that then captures, (c) a stream that joins the capture through an event wait."""
import threading
import time

import torch

dev = torch.device("cuda:0")
side, other, joins = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 20, device=dev)
events = {}
for name, st in (("other stream", other), ("capturing stream", side), ("joining stream", joins)):
    with torch.cuda.stream(st):
        x.add_(1)
        e = torch.cuda.Event()
        e.record(st)
        events[name] = e
torch.cuda.synchronize()
results, stop = {}, threading.Event()


def poll():
    while not stop.is_set():
        for name, e in events.items():
            try:
                e.query()
                results.setdefault(name, "ok")
            except Exception as exc:       # noqa: BLE001
                results[name] = str(exc).splitlines()[0][:120]
        time.sleep(0.01)


g = torch.cuda.CUDAGraph()
y = torch.zeros(1 << 20, device=dev)
with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
    y.add_(1)
    joins.wait_stream(side)
    with torch.cuda.stream(joins):
        y.mul_(2)
    side.wait_stream(joins)
    t = threading.Thread(target=poll)
    results.clear()
    t.start()
    time.sleep(0.3)
    stop.set()
    t.join()
    for name in events:
        print(f"query during capture, event last recorded (eagerly, before the capture) on the {name}: {results.get(name)}", flush=True)
g.replay()
torch.cuda.synchronize()
print("after capture:", {n: e.query() for n, e in events.items()})
