#!/bin/bash
# Same-box alternating A/B of the default bench line incl. its informational workloads (8 clips batched, R = 64, autoencoder).
# Usage: bash tools/full_ab.sh <out> <reps> "<env a>" "<env b>";  one line per run: label, repeat, ms/step, steps/s, batched clip-steps/s, R=64 steps/s, AE decode ms, AE extract ms
O=$1; R=${2:-2}; A=$3; B=$4
cd "$(dirname "$0")/.."
: > $O
for r in $(seq 1 $R); do
    for lab in a b; do
        if [ $lab = a ]; then E=$A; else E=$B; fi
        env $E python bench.py --steps 250 --warmup 25 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab_line.json
        python - $lab $r >> $O <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
b = d.get("batched_info") or {}; r6 = d.get("res64_info") or {}; ae = d.get("autoencoder_info") or {}
print(sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"], b.get("clip_steps_per_s"), r6.get("steps_per_s"), ae.get("decode_from_sample_ms"), ae.get("extract_ms"))
PY
    done
done
cat $O
