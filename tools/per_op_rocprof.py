#!/usr/bin/env python3
"""Join a rocprofv3 kernel trace of bench.py with the op names of tools/profile_ops.py (same plan order:
one kernel per op) -> median GPU duration per op over the steady-state steps.
Usage: per_op_rocprof.py <ops.txt> <kernel_trace.csv> [out.txt]"""
import collections
import csv
import re
import statistics
import sys

ops = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*[\d.]+ us\s+[\d.]+ TF/s\s+[\d.]+ GB/s\s+(.*)$", l)
    if m:
        ops.append(m.group(1).strip())
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a sampler step = len(ops) consecutive dispatches after a k_ddim_init (see summarize_profile.py); the longest run is the timed one
runs, start = [], None
for i, r in enumerate(rows):
    if "k_ddim_init" in r["Kernel_Name"]:
        start = i + 1
    elif start is not None and "k_step_sinusoid" in r["Kernel_Name"]:
        runs.append((start, i))
        start = None
if start is not None:
    runs.append((start, len(rows)))
s0, s1 = max(runs, key=lambda se: se[1] - se[0])
n = len(ops)
idx = [s0 + j * n for j in range((s1 - s0) // n + 1)]
per = collections.defaultdict(list)
for a, b in zip(idx[20:-2], idx[21:-1]):
    ks = rows[a:b]
    for j, (name, r) in enumerate(zip(ops, ks)):
        per[(j, name)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tab = [(statistics.median(v) / 1e3, k[1]) for k, v in sorted(per.items())]
fam = collections.defaultdict(lambda: [0, 0.0])
for us, k in tab:
    f = k.split(":")[0]
    if f in ("conv3", "conv1"):
        f += " M" + re.search(r"\[(\d+)x", k).group(1)
    if f == "attn":
        mm = re.search(r"\[L(\d+) d(\d+) (\w+)", k)
        f += f" L{mm.group(1)} {mm.group(3)}" + (" blk" if " blk " in k else "")
    fam[f][0] += 1
    fam[f][1] += us
out = [f"# median GPU duration per launch of one DDIM step (rocprofv3 kernel trace, {len(idx) - 23} steady-state steps), plan order"]
out += [f"{us:8.2f} us  {k}" for us, k in tab]
out.append(f"# total {sum(t for t, _ in tab):.1f} us; by family:")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    out.append(f"#   {k:16s} n={c:3d} {t:8.1f} us  avg {t / c:6.2f}")
txt = "\n".join(out) + "\n"
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt)
print("\n".join(out[-24:]))
