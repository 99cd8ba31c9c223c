#!/bin/bash
# Per-op A/B of k_conv_win tiles with K slices (MT, NT, KS) on the 3x3 convs of one sampler step: the per-launch hipEvent table of
# tools/profile_ops.py under MTV_FORCE_WIN=mt,nt,ks, one process per tile, then the best forced tile per op beside the default.
# Usage (GPU box): bash tools/win_ks_ab.sh [outdir]
O=${1:-gpurun_out/win_ks}; mkdir -p $O
cd "$(dirname "$0")/.."
python tools/profile_ops.py --iters 30 > $O/default.txt 2>/dev/null
for cfg in 2,4,1 2,4,2 2,4,4 2,2,1 2,2,2 2,2,4 1,4,1 1,4,2 1,4,4 1,2,2; do
    MTV_FORCE_WIN=$cfg python tools/profile_ops.py --iters 30 > $O/win_$cfg.txt 2>/dev/null
done
python - "$O" <<'PY'
import glob, os, re, sys
O = sys.argv[1]
tab = {}
for f in sorted(glob.glob(os.path.join(O, "*.txt"))):
    cfg = os.path.basename(f)[:-4]
    for l in open(f):
        m = re.match(r"\s*([0-9.]+) us .*?(conv3:\S+)\[(\S+) (k\d+) (t[^\]]*)\]", l)
        if m:
            tab.setdefault((m.group(2), m.group(3), m.group(4)), {})[cfg] = (float(m.group(1)), m.group(5))
tot_def = tot_best = 0.0
print(f"{'op':40s} {'default':>20s}   best forced")
for (op, shape, k), v in tab.items():
    if "default" not in v:
        continue
    d = v["default"]
    cand = {c: t for c, t in v.items() if c != "default" and ",80," in t[1]}
    if not cand:
        continue
    b = min(cand.items(), key=lambda kv: kv[1][0])
    tot_def += d[0]
    tot_best += min(d[0], b[1][0])
    print(f"{op + ' ' + shape + ' ' + k:40s} {d[0]:7.2f} {d[1]:>12s}   {b[1][0]:7.2f} {b[1][1]:>12s}   " + " ".join(f"{t[1][1:]}={t[0]:.1f}" for c, t in sorted(cand.items())))
print(f"# sum over the listed 3x3 launches: default {tot_def:.1f} us, best per op {tot_best:.1f} us (hipEvents around plain launches: includes the event overhead on both sides)")
PY
