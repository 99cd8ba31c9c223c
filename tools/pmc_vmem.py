#!/usr/bin/env python3
"""Vector-memory path counters per kernel template of the step, from a rocprofv3 --kernel-trace --pmc pass of bench.py:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum \
              TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -- python bench.py ...
What to read: tag look-ups per read instruction (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD; SQ_INSTS counts per wave: a 16-byte-per-lane
load that touches whole 64-byte pieces needs 16, a badly laid-out one up to 64), L1 -> L2 read requests per instruction, the texture-address
unit's busy share of the kernel, and address-translation misses.
Usage: pmc_vmem.py <counter_collection.csv> [more passes ...]"""
import collections
import csv
import re
import sys

# (the counters may come from several passes -- one csv each: every pass is averaged per dispatch of a kernel template on its own)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[1:]:
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), dict(name=r["Kernel_Name"]))
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    part = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in disp.values():
        n = re.sub(r"^void ", "", d["name"]).replace("mtv::", "").split("(")[0]
        part[n]["n"] += 1
        for k, v in d.items():
            if k != "name":
                part[n][k] += v
    for n, a in part.items():
        for k, v in a.items():
            if k != "n" and k not in agg[n]:
                agg[n][k] = v / a["n"]
        agg[n]["n"] = a["n"]
for a in agg.values():                      # (below: per-dispatch means times the dispatch count, so that the ratios read as before)
    for k in list(a):
        if k != "n":
            a[k] *= a["n"]
print(f"{'kernel':34s} {'disp':>6s} {'rd inst':>9s} {'wr inst':>8s} {'tag/rd':>7s} {'L2req/rd':>8s} {'TA busy':>8s} {'TLB miss':>9s} {'pend stall':>10s}   (per dispatch; TA busy / pending stall = share of GRBM_GUI_ACTIVE x 256 CUs)")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    c = a["n"]
    rd, wr = a.get("SQ_INSTS_VMEM_RD", 0) / c, a.get("SQ_INSTS_VMEM_WR", 0) / c
    tag = a.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / c
    l2 = a.get("TCP_TCC_READ_REQ_sum", 0) / c
    act = a.get("GRBM_GUI_ACTIVE", 0) / c
    cu_cycles = act / 8 * 256 if act else 1            # GRBM_GUI_ACTIVE sums the 8 XCDs
    print(f"{n[:34]:34s} {int(c):6d} {rd:9.0f} {wr:8.0f} {tag / max(rd + wr, 1):7.1f} {l2 / max(rd, 1):8.1f} {a.get('TA_TA_BUSY_sum', 0) / c / cu_cycles:8.3f} {a.get('TCP_UTCL1_TRANSLATION_MISS_sum', 0) / c:9.0f} {a.get('TCP_PENDING_STALL_CYCLES_sum', 0) / c / cu_cycles:10.3f}")
