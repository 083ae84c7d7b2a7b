"""One replayed step cut out of a rocprofv3 kernel trace of bench.py (the launches between the optimizer kernels of two consecutive
steps): launches, busy time and share per kernel class.  Small kernels carry the profiler's per-dispatch cost here; the span is the
profiled step, not the bench number.   usage: step_census.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys


def short(n):
    for k in ("copyBuffer", "FillFunctor", "bfloat16_copy", "bfloat16tofloat32", "direct_copy", "CUDAFunctor_add", "Gelu", "MulFunctor"):
        if k in n:
            return "aten:" + k
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
        return "GEMM (hipBLASLt)"
    if "_ZN2bg" in n:
        m = re.search(r"_ZN2bg\d+(\w+?)I", n)
        return "bg:" + (m.group(1) if m else n[:30])
    if "bg::" in n:
        return "bg:" + n.split("bg::")[1].split("<")[0].split("(")[0]
    return "aten:other"


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_step" in r["Kernel_Name"]]
a, b = adam[-5], adam[-3]            # two optimizer launches per step: one whole step, its own optimizer launches last
seg = rows[a + 1:b + 1]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
cnt, dur = collections.Counter(), collections.Counter()
for r in seg:
    s = short(r["Kernel_Name"])
    cnt[s] += 1
    dur[s] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(dur.values())
print(f"one replayed step: {len(seg)} launches, span {span:.3f} ms, summed kernel time {busy / 1e3:.3f} ms (kernels of the two streams overlap)")
print(f"{'class':44s} {'launches':>8s} {'us':>10s} {'share':>6s}")
for k, v in dur.most_common():
    print(f"{k:44s} {cnt[k]:8d} {v:10.1f} {100 * v / busy:5.1f}%")
