#!/bin/bash
# A/B of k_conv_pw tiles (MT, NTW, multiplying waves) on the 1x1 convs of one sampler step: per-launch hipEvent table of tools/profile_ops.py under
# MTV_FORCE_PW, one process per tile, then the best tile per op.  Usage (GPU box): bash tools/pw_tile_ab.sh [outdir]
O=${1:-gpurun_out/pw_ab}; mkdir -p $O
cd "$(dirname "$0")/.."
python tools/profile_ops.py --iters 30 > $O/default.txt 2>/dev/null
for cfg in 2,1,1 1,1,1 2,1,6 1,1,6 2,1,4 1,1,4 2,1,2 1,1,2 1,2,1 2,2,1; do
    MTV_FORCE_PW=$cfg python tools/profile_ops.py --iters 30 > $O/pw_$cfg.txt 2>/dev/null
done
python - "$O" <<'PY'
import glob, os, re, sys
O = sys.argv[1]
tab = {}
for f in sorted(glob.glob(os.path.join(O, "*.txt"))):
    cfg = os.path.basename(f)[:-4]
    for l in open(f):
        m = re.match(r"\s*([0-9.]+) us .*?(conv1:\S+)\[(\S+) (k\d+) (t[^\]]*)\]", l)
        if m:
            tab.setdefault((m.group(2), m.group(3), m.group(4)), {})[cfg] = (float(m.group(1)), m.group(5))
tot_def = tot_best = 0.0
print(f"{'op':34s} {'default':>18s}   best forced")
for (op, shape, k), v in tab.items():
    if "default" not in v:
        continue
    d = v["default"]
    cand = {c: t for c, t in v.items() if c != "default" and ",96," in t[1]}
    b = min(cand.items(), key=lambda kv: kv[1][0]) if cand else ("-", d)
    tot_def += d[0]
    tot_best += min(d[0], b[1][0])
    print(f"{op + ' ' + shape + ' ' + k:34s} {d[0]:7.2f} {d[1]:>10s}   {b[1][0]:7.2f} {b[0]:>9s}   " + " ".join(f"{c[3:]}={t[0]:.1f}" for c, t in sorted(cand.items())))
print(f"# sum over the 1x1 launches: default {tot_def:.1f} us, best per op {tot_best:.1f} us (hipEvents around plain launches: includes the event overhead on both sides)")
PY
