#!/usr/bin/env python3
"""LDS bank-conflict share per kernel template from a rocprofv3 --kernel-trace --pmc pass of bench.py (MTV_EAGER=1):
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -- python bench.py ...
conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent re-issuing for bank conflicts / cycles it was active).
Usage: pmc_lds.py <counter_collection.csv>"""
import collections, csv, re, sys
disp = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    d = disp.setdefault(int(r["Dispatch_Id"]), dict(name=r["Kernel_Name"]))
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    n = re.sub(r"^void ", "", d["name"]).replace("mtv::", "").split("(")[0]
    agg[n]["n"] += 1
    for k, v in d.items():
        if k != "name":
            agg[n][k] += v
print(f"{'kernel':34s} {'disp':>6s} {'LDS insts':>10s} {'LDS active':>11s} {'conflict':>10s} {'conflict/active':>15s} {'LDS active / CU-cycles':>22s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    c = a["n"]
    act = a.get("SQ_LDS_IDX_ACTIVE", 0) / c
    conf = a.get("SQ_LDS_BANK_CONFLICT", 0) / c
    cu = a.get("GRBM_GUI_ACTIVE", 0) / c / 8 * 256 or 1
    print(f"{n[:34]:34s} {int(c):6d} {a.get('SQ_INSTS_LDS', 0) / c:10.0f} {act:11.0f} {conf:10.0f} {conf / max(act, 1):15.3f} {act / cu:22.3f}")
