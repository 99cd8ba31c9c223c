#!/bin/bash
# Same-box round-robin A/B of the sampler step under N environments.  Usage (GPU box):
#   bash tools/multi_ab.sh <out file> <reps> "<env 1>" "<env 2>" ...      (an empty string = the default environment)
# One line per run: config number, repeat, ms per step, steps/s, launches, conv roofline.frac.
O=$1; R=$2; shift 2
cd "$(dirname "$0")/.."
: > $O
for r in $(seq 1 $R); do
    k=0
    for E in "$@"; do
        k=$((k + 1))
        env $E python bench.py --steps 250 --warmup 25 --no-cpu-baseline --no-autoencoder --no-res64 --batched-clips 0 2>/dev/null | tail -1 > /tmp/ab_line.json
        python - $k $r >> $O <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print(sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"], d.get("launches_per_step"), d["roofline"]["frac"])
PY
    done
done
cat $O
