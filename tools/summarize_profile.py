#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace CSV of a bench.py run into a per-STEP kernel summary (steady state:
one DDIM step between two consecutive k_ddim_update dispatches), optionally joined with PMC passes.
Usage: summarize_profile.py <kernel_trace.csv> [--fetch f_counter_collection.csv] [--write w_counter_collection.csv]"""
import argparse
import collections
import csv
import statistics


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("mtv::", "")
    return n


def steps_of(rows, key_ts):
    rows.sort(key=key_ts)
    idx = [i for i, r in enumerate(rows) if "k_ddim_update" in r["Kernel_Name"]]
    return rows, idx


ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--fetch")
ap.add_argument("--write")
ap.add_argument("--skip", type=int, default=15, help="steps to skip (warm-up / tuning)")
a = ap.parse_args()
rows, idx = steps_of(list(csv.DictReader(open(a.trace))), lambda r: int(r["Start_Timestamp"]))
sel = list(zip(idx[a.skip:-1], idx[a.skip + 1:]))
agg = collections.defaultdict(lambda: [0, 0.0])
spans, busy = [], []
for s, e in sel:
    step = rows[s + 1:e + 1]
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
    f = "k_conv<*>" if k.startswith("k_conv") else ("k_attention<*>" if k.startswith("k_attention") else k)
    fam[f][0] += c
    fam[f][1] += t
print("# by family")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:34s} {c / n:13.1f} {t / n / 1e3:10.1f} {t / c / 1e3:8.2f} {100 * t / tot:6.1f}")
for label, path in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
    if not path:
        continue
    rows2, idx2 = steps_of(list(csv.DictReader(open(path))), lambda r: int(r["Dispatch_Id"]))
    sel2 = list(zip(idx2[5:-1], idx2[6:]))
    fam2 = collections.defaultdict(float)
    for s, e in sel2:
        for r in rows2[s + 1:e + 1]:
            k = short(r["Kernel_Name"])
            f = "k_conv<*>" if k.startswith("k_conv") else ("k_attention<*>" if k.startswith("k_attention") else k)
            fam2[f] += float(r["Counter_Value"])
    print(f"# {label} per step, raw counter (KB -> MB); gfx950 FETCH_SIZE reads 1/2 of wide streaming reads (MI355X_MICROARCH.md)")
    for k, v in sorted(fam2.items(), key=lambda kv: -kv[1])[:6]:
        print(f"  {k:34s} {v / len(sel2) / 1024:10.1f} MB")
