#!/usr/bin/env python3
"""Join rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (MTV_EAGER=1: plain launches, one dispatch per op) with the op
names of tools/profile_ops.py (plan order) -> measured HBM-side bytes per op vs the plan's algorithmic bytes.
Usage: per_op_traffic.py <ops.txt> <fetch_counter_collection.csv> [<write_counter_collection.csv>] [--out file]
FETCH_SIZE / WRITE_SIZE are in KB; gfx950's FETCH_SIZE reports 1/2 of wide streaming reads (MI355X_MICROARCH.md, HBM):
the 'x2' column applies that correction, the ratio column is (2 x FETCH + WRITE) / algorithmic."""
import argparse
import collections
import csv
import re
import statistics

ap = argparse.ArgumentParser()
ap.add_argument("ops")
ap.add_argument("fetch")
ap.add_argument("write", nargs="?")
ap.add_argument("--out")
a = ap.parse_args()

ops = []
for l in open(a.ops):
    m = re.match(r"\s*([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) GB/s\s+(.*)$", l)
    if m:
        us, gbs = float(m.group(1)), float(m.group(3))
        ops.append((m.group(4).strip(), us * 1e-6 * gbs * 1e9))       # name, algorithmic bytes (bytes = GB/s x time)


def per_op(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
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
    per = collections.defaultdict(list)
    for j in range(3, (s1 - s0) // n - 1):
        for k, r in enumerate(rows[s0 + j * n:s0 + (j + 1) * n]):
            per[k].append(float(r["Counter_Value"]) * 1024.0)
    return [statistics.median(per[k]) for k in range(n)]


F = per_op(a.fetch)
W = per_op(a.write) if a.write else [0.0] * len(ops)
lines = ["# per op of one DDIM step (plan order): FETCH_SIZE raw MB | x2 MB | WRITE_SIZE MB | algorithmic MB | (2 FETCH + WRITE) / algorithmic"]
fam = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for (name, alg), f, w in zip(ops, F, W):
    ratio = (2 * f + w) / alg if alg > 0 else 0.0
    lines.append(f"{f / 1e6:8.2f} {2 * f / 1e6:8.2f} {w / 1e6:8.2f} {alg / 1e6:8.2f} {ratio:6.2f}  {name}")
    k = name.split(":")[0]
    if k in ("conv3", "conv1"):
        k += " M" + re.search(r"\[(\d+)x", name).group(1)
    fam[k][0] += 2 * f + w
    fam[k][1] += alg
    fam[k][2] += 1
lines.append("# by family: measured MB (2 FETCH + WRITE) | algorithmic MB | ratio")
for k, (m, alg, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    lines.append(f"#   {k:14s} n={int(c):3d} {m / 1e6:9.1f} {alg / 1e6:9.1f} {m / alg if alg else 0:6.2f}")
tm, ta = sum(v[0] for v in fam.values()), sum(v[1] for v in fam.values())
lines.append(f"#   total {tm / 1e6:.1f} MB measured vs {ta / 1e6:.1f} MB algorithmic = {tm / ta:.2f}x")
txt = "\n".join(lines) + "\n"
if a.out:
    open(a.out, "w").write(txt)
print("\n".join(lines[-14:]))
