#!/bin/bash
# Same-box alternating A/B of the sampler step under two environments.  Usage (GPU box):
#   bash tools/step_ab.sh <out file> <reps> "<env of a>" "<env of b>"      e.g.  bash tools/step_ab.sh gpurun_out/ab.txt 3 "" "MTV_ARENA_MB=256"
# One line per run: label, repeat, ms per step, steps/s, launches, conv roofline.frac.
O=$1; R=${2:-3}; A=$3; B=$4
cd "$(dirname "$0")/.."
: > $O
for r in $(seq 1 $R); do
    for lab in a b; do
        if [ $lab = a ]; then E=$A; else E=$B; fi
        env $E python bench.py --steps 250 --warmup 25 --no-cpu-baseline --no-autoencoder --no-res64 --batched-clips 0 2>/dev/null | tail -1 > /tmp/ab_line.json
        python - $lab $r >> $O <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print(sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"], d.get("launches_per_step"), d["roofline"]["frac"])
PY
    done
done
cat $O
