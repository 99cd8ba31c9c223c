#!/usr/bin/env python3
"""Counter-based utilisation of the step's kernel families from a rocprofv3 --kernel-trace --pmc pass of bench.py
(MTV_EAGER=1: one dispatch per op):
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
              SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python bench.py ...
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the fraction of SIMD-cycles in which
the matrix pipe is busy while the kernel runs (an MFMA-only kernel reads 0.94-0.97, profiles/r03_mfma_valu_counters.txt).
Optionally joins the FETCH_SIZE / WRITE_SIZE passes and a kernel trace for counter-based GB/s.
Usage: pmc_util.py <sq_counter_collection.csv> --launches N [--fetch f.csv --write w.csv --trace kernel_trace.csv] [--json out.json]"""
import argparse
import collections
import csv
import json
import statistics


def fam_of(name):
    n = name.split("(")[0].replace("void ", "").replace("mtv::", "")
    for k in ("k_conv_lds", "k_conv", "k_lin", "k_deep_conv", "k_deep_attn", "k_deep_block", "k_deep_finalize", "k_attention_b3", "k_attention", "k_pool_down"):
        if n.startswith(k):
            # (k_deep_conv: the K-sliced convs of the <= 128-token levels, csrc/deep.hip -- part of the conv family)
            return {"k_conv_lds": "k_conv", "k_lin": "k_conv", "k_deep_conv": "k_conv", "k_attention_b3": "k_attention"}.get(k, k)
    return n


def step_rows(rows, key, n):
    rows.sort(key=key)
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
    k = (s1 - s0) // n
    return [rows[s0 + j * n:s0 + (j + 1) * n] for j in range(3, k - 1)]


ap = argparse.ArgumentParser()
ap.add_argument("sq")
ap.add_argument("--launches", type=int, required=True)
ap.add_argument("--fetch")
ap.add_argument("--write")
ap.add_argument("--trace")
ap.add_argument("--json")
a = ap.parse_args()

# one row per (dispatch, counter) -> one dict per dispatch
disp = collections.OrderedDict()
for r in csv.DictReader(open(a.sq)):
    d = disp.setdefault(int(r["Dispatch_Id"]), dict(Kernel_Name=r["Kernel_Name"], Dispatch_Id=r["Dispatch_Id"]))
    d[r["Counter_Name"]] = float(r["Counter_Value"])
steps = step_rows(list(disp.values()), lambda r: int(r["Dispatch_Id"]), a.launches)
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for st in steps:
    for r in st:
        f = fam[fam_of(r["Kernel_Name"])]
        for k, v in r.items():
            if isinstance(v, float):
                f[k] += v
        f["n"] += 1
out = {}
print(f"# {len(steps)} steady-state DDIM steps (MTV_EAGER=1 launches); per family and step")
print(f"# {'family':14s} {'launches':>8s} {'MFMA util':>9s} {'coexec/MFMA busy':>16s} {'VALU-active/wave-cycles':>23s} {'MFMA insts':>11s} {'VALU insts':>11s}")
for k, f in sorted(fam.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    simd_cyc = f["GRBM_GUI_ACTIVE"] / 8 * 1024
    util = f["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cyc if simd_cyc else 0.0
    co = f["SQ_VALU_MFMA_COEXEC_CYCLES"] / f["SQ_VALU_MFMA_BUSY_CYCLES"] if f["SQ_VALU_MFMA_BUSY_CYCLES"] else 0.0
    va = f["SQ_ACTIVE_INST_VALU"] / f["SQ_WAVE_CYCLES"] if f["SQ_WAVE_CYCLES"] else 0.0
    ns = len(steps)
    print(f"  {k:14s} {f['n'] / ns:8.0f} {util:9.3f} {co:16.3f} {va:23.3f} {f['SQ_INSTS_MFMA'] / ns:11.0f} {f['SQ_INSTS_VALU'] / ns:11.0f}")
    out[k] = dict(launches_per_step=round(f["n"] / ns), mfma_util_eager=round(util, 4), coexec_over_mfma_busy=round(co, 4),
                  mfma_busy_cycles_per_step=round(f["SQ_VALU_MFMA_BUSY_CYCLES"] / ns), mfma_insts_per_step=round(f["SQ_INSTS_MFMA"] / ns))
if a.fetch and a.write and a.trace:
    DEEP = "k_deep_conv_only"          # the weight-streaming convs of the <= 128-token levels on their own (also part of the k_conv family)

    def is_deep(name):
        return name.split("(")[0].replace("void ", "").replace("mtv::", "").startswith("k_deep_conv")

    def per_family(path):
        rows = list(csv.DictReader(open(path)))
        acc = collections.defaultdict(float)
        sts = step_rows(rows, lambda r: int(r["Dispatch_Id"]), a.launches)
        for st in sts:
            for r in st:
                acc[fam_of(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0
                if is_deep(r["Kernel_Name"]):
                    acc[DEEP] += float(r["Counter_Value"]) * 1024.0
        return {k: v / len(sts) for k, v in acc.items()}
    F, W = per_family(a.fetch), per_family(a.write)
    rows = list(csv.DictReader(open(a.trace)))
    sts = step_rows(rows, lambda r: int(r["Start_Timestamp"]), a.launches)
    T = collections.defaultdict(float)
    NL = collections.defaultdict(int)
    for st in sts:
        for r in st:
            T[fam_of(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            if is_deep(r["Kernel_Name"]):
                T[DEEP] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                NL[DEEP] += 1
    print("# counter-based HBM-side traffic: (2 x FETCH_SIZE + WRITE_SIZE) / family time of the kernel trace (graph replay)")
    for k in sorted(F, key=lambda k: -F[k]):
        if k not in T or T[k] <= 0:
            continue
        t = T[k] / len(sts)
        b = 2 * F[k] + W.get(k, 0.0)
        print(f"  {k:14s} FETCH {F[k] / 1e6:8.1f} MB  WRITE {W.get(k, 0) / 1e6:7.1f} MB  time {t * 1e6:8.1f} us/step  -> {b / t / 1e9:8.1f} GB/s")
        out.setdefault(k, {}).update(fetch_raw_MB_per_step=round(F[k] / 1e6, 1), write_raw_MB_per_step=round(W.get(k, 0) / 1e6, 1),
                                     ms_per_step_trace=round(t * 1e3, 4), hbm_counter_GBs=round(b / t / 1e9, 1))
        if k == DEEP:
            out[k]["launches_per_step"] = round(NL[DEEP] / len(sts))
        if "mfma_busy_cycles_per_step" in out[k]:      # matrix-pipe busy cycles (all SIMDs) over the SIMD-cycles of the family's graph-replay time at 2.4 GHz
            out[k]["mfma_util"] = round(out[k]["mfma_busy_cycles_per_step"] / (t * 2.4e9 * 1024), 4)
            print(f"  {k:14s} matrix-pipe busy / (kernel-trace family time x 2.4 GHz x 1024 SIMDs) = {out[k]['mfma_util']:.3f}")
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
