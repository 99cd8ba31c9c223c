#!/bin/bash
# Regenerates moditalker_amd/csrc/tune_gfx950.txt on an MI355X box (gpurun --timeout 900 -- 'bash tools/make_tune_table.sh'):
# every conv shape of the BASELINE configurations (B=1 and the batched B=8 at R=32, B=1 at R=64) and of the test suite's
# small models is timed cold, median of 5, over all valid tiles.  Output: gpurun_out/tune_gfx950.txt (copy it over the committed one).
cd "$(dirname "$0")/.."
export MTV_TUNE_CACHE=$PWD/gpurun_out/tune_raw.txt
rm -f $MTV_TUNE_CACHE; mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 2 --ramp-steps 5 --no-cpu-baseline --batched-clips 8 > /dev/null
timeout 600 python bench.py --res 64 --steps 10 --warmup 2 --ramp-steps 5 --no-cpu-baseline --batched-clips 0 > /dev/null
timeout 300 python tools/ae_profile.py > /dev/null                  # the autoencoder's GEMM shapes (decode, then extract)
timeout 300 python tools/ae_profile.py --extract > /dev/null
(echo "# conv shape -> measured best tile (MT NT NW KS XM); regenerate with tools/make_tune_table.sh on an MI355X"; sort -u $MTV_TUNE_CACHE) > gpurun_out/tune_gfx950.txt
wc -l gpurun_out/tune_gfx950.txt
