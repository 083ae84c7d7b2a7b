#!/usr/bin/env python3
"""Static checks on the gfx950 assembly of one csrc file (no GPU): per kernel, the registers and scratch it takes and the
global loads that are WAITED FOR ON THE SPOT (`s_waitcnt vmcnt(0)` within three instructions of the load) - the mark of a
dependent round trip per load: a load guarded by its own branch, a conversion inside that branch, one load per loop iteration.
Round 3 found the serial row load of top-p, LayerNorm's operands, the attention backward's slot rows and incoming gradients,
the embedding backward's index lists and the 310 spilled registers of the loss diagnostics this way.

usage: asm_scan.py <file.hip> [kernel-name substring]      (run from anywhere; compiles with the Makefile's flags + line tables)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
if not os.path.exists(src):
    src = os.path.join(ROOT, "branch-gan_amd", "csrc", src)
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-gline-tables-only",
                "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)

files, cur, loc, recent = {}, None, None, []
loads, waits, spot = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)
meta, name = {}, None
for ln in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        cur, recent = m.group(1), []
        continue
    m = re.match(r'\s+\.name:\s+(\S+)', ln)
    if m:
        name = m.group(1)
        meta[name] = {}
    m = re.match(r'\s+\.(private_segment_fixed_size|vgpr_count|sgpr_spill_count|vgpr_spill_count):\s+(\d+)', ln)
    if m and name:
        meta[name][m.group(1)] = int(m.group(2))
    if cur is None:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        loc = f"{files.get(m.group(1), m.group(1))}:{m.group(2)}"
        continue
    t = ln.strip()
    if not t or t[0] in ';.':
        continue
    recent = (recent + [(t, loc)])[-4:]
    if t.startswith(('global_load', 'buffer_load')):
        loads[cur] += 1
    if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
        waits[cur] += 1
        prev = [x for x in recent[:-1] if x[0].startswith(('global_load', 'buffer_load'))]
        if prev:
            spot[cur][prev[-1][1]] += 1

for k in loads:
    if want not in k:
        continue
    md = meta.get(k, {})
    print(f"{k[:120]}\n   vgprs {md.get('vgpr_count')}  spilled {md.get('vgpr_spill_count')}  scratch {md.get('private_segment_fixed_size')} B"
          f"  | global loads {loads[k]}  vmcnt(0) waits {waits[k]}  waited on the spot {sum(spot[k].values())}")
    for where, n in sorted(spot[k].items(), key=lambda x: -x[1])[:6]:
        print(f"      {n:4d} x at {where}")
