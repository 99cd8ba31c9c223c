#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace CSV of a bench.py run into a per-STEP kernel summary, optionally joined with PMC passes.
A sampler step is `--launches` consecutive dispatches (the UNet's own launches; bench.py prints the number as
launches_per_step): the dispatches between a k_ddim_init (end of the sampler set-up) and the next set-up's
k_step_sinusoid are cut into steps of that length; the longest such run (the timed one) is summarised, steady state only.
Usage: summarize_profile.py <kernel_trace.csv> --launches N [--fetch f_counter_collection.csv] [--write w_counter_collection.csv]"""
import argparse
import collections
import csv
import statistics


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("mtv::", "")
    return n


def steps_of(rows, key_ts, n):
    """-> rows (sorted), list of (first, last+1) row ranges, one per step of the longest sampler run"""
    rows.sort(key=key_ts)
    runs, start = [], None
    for i, r in enumerate(rows):
        if "k_ddim_init" in r["Kernel_Name"]:
            start = i + 1
        elif start is not None and "k_step_sinusoid" in r["Kernel_Name"]:
            runs.append((start, i))
            start = None
    if start is not None:
        runs.append((start, len(rows)))
    # (memcpy nodes are not kernels; anything that is not part of a step -- the final gather's copy kernels -- is cut
    # off by taking whole multiples of n)
    best = max(runs, key=lambda se: se[1] - se[0])
    k = (best[1] - best[0]) // n
    return rows, [(best[0] + j * n, best[0] + (j + 1) * n) for j in range(k)]


ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--fetch")
ap.add_argument("--write")


def family(k):
    # the conv family = k_conv / k_conv_lds / k_conv_x3 / k_lin and the K-sliced k_deep_conv of the <= 128-token levels (csrc/deep.hip)
    if k.startswith("k_conv") or k.startswith("k_deep_conv") or k.startswith("k_lin"):
        return "k_conv<*> + k_deep_conv<*>"
    if k.startswith("k_attention"):
        return "k_attention<*>"
    return k


ap.add_argument("--launches", type=int, required=True, help="dispatches per sampler step (bench.py: launches_per_step)")
ap.add_argument("--skip", type=int, default=15, help="steps to skip at the start of the run")
a = ap.parse_args()
rows, steps = steps_of(list(csv.DictReader(open(a.trace))), lambda r: int(r["Start_Timestamp"]), a.launches)
sel = steps[a.skip:-1]
agg = collections.defaultdict(lambda: [0, 0.0])
spans, busy = [], []
for s, e in sel:
    step = rows[s:e]
    spans.append(int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"]))
    b = 0
    for r in step:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        b += d
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += d
    busy.append(b)
n = len(sel)
print(f"# {n} steady-state DDIM steps; step span median {statistics.median(spans) / 1e3:.1f} us, "
      f"kernel-busy median {statistics.median(busy) / 1e3:.1f} us")
print(f"# {'kernel':34s} {'launches/step':>13s} {'us/step':>10s} {'avg us':>8s} {'%':>6s}")
tot = sum(v[1] for v in agg.values())
fam = collections.defaultdict(lambda: [0, 0.0])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:34s} {c / n:13.1f} {t / n / 1e3:10.1f} {t / c / 1e3:8.2f} {100 * t / tot:6.1f}")
    f = family(k)
    fam[f][0] += c
    fam[f][1] += t
print("# by family")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:34s} {c / n:13.1f} {t / n / 1e3:10.1f} {t / c / 1e3:8.2f} {100 * t / tot:6.1f}")
for label, path in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
    if not path:
        continue
    rows2, steps2 = steps_of(list(csv.DictReader(open(path))), lambda r: int(r["Dispatch_Id"]), a.launches)
    sel2 = steps2[5:-1]
    fam2 = collections.defaultdict(float)
    for s, e in sel2:
        for r in rows2[s:e]:
            k = short(r["Kernel_Name"])
            f = family(k)
            fam2[f] += float(r["Counter_Value"])
    print(f"# {label} per step, raw counter (KB -> MB); gfx950 FETCH_SIZE reads 1/2 of wide streaming reads (MI355X_MICROARCH.md)")
    for k, v in sorted(fam2.items(), key=lambda kv: -kv[1])[:6]:
        print(f"  {k:34s} {v / len(sel2) / 1024:10.1f} MB")
