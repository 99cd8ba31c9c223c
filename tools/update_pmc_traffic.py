#!/usr/bin/env python3
"""Folds a tools/pmc_util.py --json result into profiles/pmc_traffic.json (the committed counter record bench.py quotes as
`roofline.traffic` / `hbm_counter_GBs` / `mfma_util_counter`).  The file is keyed by workload ("R32" = BASELINE configs[1],
"R64" = configs[3]); bench.py only ever quotes the entry of the workload it runs.
Usage: update_pmc_traffic.py <pmc_util.json> <label> [commit] [workload = R32]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "pmc_traffic.json")
new = json.load(open(sys.argv[1]))
wl = sys.argv[4] if len(sys.argv) > 4 else "R32"
doc = json.load(open(path))
cur = doc.setdefault("workloads", {}).get(wl, {})
prev = {k: cur[k] for k in ("k_conv", "k_attention") if k in cur}
doc["source"] = ("rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE, and the SQ / GRBM set of tools/pmc_util.py; separate runs, "
                 "graph replay first, MTV_EAGER=1 -- the same launches as plain launches -- where the graph path fails) of `python bench.py [--res 64] "
                 "--steps 30 --warmup 8 --no-cpu-baseline --batched-clips 0 --no-autoencoder --no-res64`, per steady-state DDIM step; bytes = 2 x FETCH_SIZE "
                 "(gfx950 reports 1/2 of wide streaming reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / "
                 "(family time of the graph-replay kernel trace x 2.4 GHz x 1024 SIMDs); not collected live; one entry per workload")
out = dict(collected=sys.argv[2], commit=sys.argv[3] if len(sys.argv) > 3 else "")
for k in ("k_conv", "k_attention", "k_deep_conv_only"):
    if k in new and "fetch_raw_MB_per_step" in new[k]:
        out[k] = {f: new[k][f] for f in ("fetch_raw_MB_per_step", "write_raw_MB_per_step", "launches_per_step", "mfma_util", "mfma_insts_per_step", "ms_per_step_trace") if f in new[k]}
    elif k in prev:
        out[k] = prev[k]
if prev:
    out["previous"] = dict(collected=cur.get("collected"), **prev)
doc["workloads"][wl] = out
json.dump(doc, open(path, "w"), indent=2)
print(json.dumps({wl: {k: out[k] for k in ("k_conv", "k_attention") if k in out}}))
