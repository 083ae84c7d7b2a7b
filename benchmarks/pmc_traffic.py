#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass together with anything else) into a
traffic record: HBM bytes per launch of the kernels whose name contains one of the given patterns.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads at
64 bytes - doubled here; WRITE_SIZE is exact.  Both counters are in KB.
usage: pmc_traffic.py <fetch dir> <write dir> <out.json> <launches per dispatch group> <label> <pattern> [pattern ...]
  launches = number of API-level launches the matched dispatches make up (a launch may be several kernels)."""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def total(directory, counter, patterns):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    kb, n, names = 0.0, 0, {}
    with open(max(files, key=os.path.getmtime), newline="") as fh:      # the newest pass (gpurun merges earlier ones into the same tree)
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            if any(p in name for p in patterns):
                kb += float(row["Counter_Value"])
                n += 1
                short = name.split("(")[0][-70:]
                names[short] = names.get(short, 0) + 1
    return kb, n, names


fetch_dir, write_dir, out, launches, label = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
patterns = sys.argv[6:]
f_kb, f_n, names = total(fetch_dir, "FETCH_SIZE", patterns)
w_kb, w_n, _ = total(write_dir, "WRITE_SIZE", patterns)
assert f_n == w_n and f_n > 0, (f_n, w_n)
if launches <= 0:            # one kernel per launch
    launches = f_n
rec = {
    "kernel": label,
    "command": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- <program>, second pass --pmc WRITE_SIZE (benchmarks/probes/pmc_traffic.sh)",
    "dispatches": f_n, "dispatches_by_kernel": names, "launches": launches,
    "FETCH_SIZE_KB_per_launch": f_kb / launches, "WRITE_SIZE_KB_per_launch": w_kb / launches,
    "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B for 16-B-per-lane loads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact",
    "traffic_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0 / launches,
}
with open(out, "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec))
