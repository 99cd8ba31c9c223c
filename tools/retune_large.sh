#!/bin/bash
# Re-times only the conv shapes of at least 1024 tokens (the ones the split-bf16 kernels of conv_x3.hip are offered for), except
# the one-clip R = 32 step's (B1, L <= 2048: the headline plan stays as tuned), and keeps every other entry of the committed table: gpurun --timeout 1500 -- 'bash tools/retune_large.sh'.
# Output: gpurun_out/tune_gfx950.txt (copy it over moditalker_amd/csrc/tune_gfx950.txt).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export MTV_TUNE_CACHE=$PWD/gpurun_out/tune_raw.txt
python - <<'PY'
import re
keep = []
for line in open("moditalker_amd/csrc/tune_gfx950.txt"):
    m = re.match(r"B(\d+) L(\d+)/", line)
    if line.startswith("#") or (m and int(m.group(1)) * int(m.group(2)) >= 1024 and not (int(m.group(1)) == 1 and int(m.group(2)) <= 2048)):
        continue
    keep.append(line)
open("gpurun_out/tune_raw.txt", "w").writelines(keep)
print(len(keep), "entries kept")
PY
timeout 900 python bench.py --steps 20 --warmup 2 --ramp-steps 5 --no-cpu-baseline --batched-clips 8 > gpurun_out/retune_bench_b8.json
timeout 600 python bench.py --res 64 --steps 10 --warmup 2 --ramp-steps 5 --no-cpu-baseline --batched-clips 0 > gpurun_out/retune_bench_r64.json
timeout 300 python tools/ae_profile.py > gpurun_out/retune_ae_decode.txt
timeout 300 python tools/ae_profile.py --extract > gpurun_out/retune_ae_extract.txt
(echo "# conv shape -> measured best tile (MT NT NW KS XM); regenerate with tools/make_tune_table.sh on an MI355X"; sort -u $MTV_TUNE_CACHE) > gpurun_out/tune_gfx950.txt
wc -l gpurun_out/tune_gfx950.txt; grep -c " 48 1 0" gpurun_out/tune_gfx950.txt
