#!/bin/bash
# Regenerates every file under profiles/ for round NN on a box with one MI355X (run from the repo root, e.g.
#   gpurun --timeout 1500 -- "MTV_COMMIT=$(git rev-parse --short HEAD) bash tools/make_profiles.sh 05";  there is no .git on the GPU box: pass the commit in).
# Outputs go to gpurun_out/final/; copy them into profiles/ as the README there names them.
set -x
NN=${1:-01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final; rm -rf $O; mkdir -p $O
# (tiles come from the committed table moditalker_amd/csrc/tune_gfx950.txt: no tuning inside the profiled processes)
# 2. per-launch hipEvent table (op names / shapes / tiles; also the join key for per_op_rocprof.py)
timeout 120 python tools/profile_ops.py --iters 20 > $O/r${NN}_per_launch_hipevents.txt 2>$O/ops.err
# 3. (the bench line comes after the counter passes: it quotes profiles/pmc_traffic.json, which step 5b refreshes)
# 4. kernel trace + stats of the same command
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- \
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --batched-clips 0 --no-res64 > $O/kt.log 2>&1
KT=$(find $O/kt -name '*kernel_trace.csv' | head -1)
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/r${NN}_rocprofv3_kernel_stats.csv
python tools/per_op_rocprof.py $O/r${NN}_per_launch_hipevents.txt $KT $O/r${NN}_per_op_rocprof.txt > /dev/null 2>&1
# 5. PMC passes, each in its own run, kernel trace only (never combined with hip/hsa/sys trace domains), on the MEASURED path: the
#    hipGraph replay (round 2's image segfaulted there; round 3 found it working again).  A pass that fails or leaves no counter
#    file is repeated with MTV_EAGER=1 (the same launches as plain launches) and the summary says which one it was.
BARGS=""      # extra bench.py arguments of the pass (the R = 64 passes set --res 64)
pmc_pass() {   # $1 = output dir tag, rest = counters
    local tag=$1; shift
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$tag -o p -- \
        python bench.py $BARGS --steps 30 --warmup 8 --no-cpu-baseline --batched-clips 0 --no-autoencoder --no-res64 > $O/pmc_$tag.log 2>&1
    local rc=$?
    if [ $rc -ne 0 ] || [ -z "$(find $O/pmc_$tag -name '*counter_collection.csv' | head -1)" ]; then
        echo "pmc $tag on the graph path: rc=$rc -> eager"; rm -rf $O/pmc_$tag
        MTV_EAGER=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$tag -o p -- \
            python bench.py $BARGS --steps 30 --warmup 8 --no-cpu-baseline --batched-clips 0 --no-autoencoder --no-res64 > $O/pmc_$tag.log 2>&1
        echo "$tag eager" >> $O/pmc_modes.txt
    else
        echo "$tag graph" >> $O/pmc_modes.txt
    fi
}
for c in FETCH_SIZE WRITE_SIZE; do pmc_pass $c $c; done
PF=$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
PW=$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
NL=$(grep -m1 -oE '^# [0-9]+ launches' $O/r${NN}_per_launch_hipevents.txt | grep -oE '[0-9]+')
python tools/summarize_profile.py $KT --launches $NL ${PF:+--fetch $PF} ${PW:+--write $PW} > $O/r${NN}_step_summary.txt 2>&1
python tools/per_op_traffic.py $O/r${NN}_per_launch_hipevents.txt $PF $PW --out $O/r${NN}_per_op_traffic.txt > /dev/null 2>&1
# 5b. matrix-pipe / VALU utilisation counters (one more PMC pass, SQ + GRBM blocks only) -> MFMA utilisation per kernel family,
#     and the counter-based GB/s of the conv family (north_star: "rocprof counters reporting achieved HBM GB/s ... and MFMA utilisation")
pmc_pass SQ SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE
echo "pmc passes: $(tr '\n' ' ' < $O/pmc_modes.txt)"
PS=$(find $O/pmc_SQ -name '*counter_collection.csv' | head -1)
python tools/pmc_util.py $PS --launches $NL --fetch $PF --write $PW --trace $KT --json $O/pmc_util.json > $O/r${NN}_pmc_util.txt 2>&1
python tools/update_pmc_traffic.py $O/pmc_util.json "round $NN (tools/make_profiles.sh)" "${MTV_COMMIT:-$(git rev-parse --short HEAD 2>/dev/null)}"; cp profiles/pmc_traffic.json $O/pmc_traffic.json
# 3. the bench line (N=1, with cpu_baseline and batched_info), quoting the counters just collected
timeout 500 python bench.py --steps 250 --warmup 25 > $O/r${NN}_bench_n1.json 2>$O/bench.err
rm -rf $O/pmc_SQ
# 6. configs[3] (R=64), informational -- with its OWN counter passes (profiles/pmc_traffic.json is keyed by workload: "R64")
timeout 300 python bench.py --res 64 --steps 40 --warmup 10 --no-cpu-baseline --batched-clips 0 > /dev/null 2>&1
if [ -z "$MTV_PROFILES_SKIP_R64_PMC" ]; then
    BARGS="--res 64"
    timeout 200 python tools/profile_ops.py --res 64 --iters 10 > $O/r${NN}_per_launch_hipevents_res64.txt 2>>$O/ops.err
    NL6=$(grep -m1 -oE '^# [0-9]+ launches' $O/r${NN}_per_launch_hipevents_res64.txt | grep -oE '[0-9]+')
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt64 -o kt -- \
        python bench.py --res 64 --steps 40 --warmup 10 --no-cpu-baseline --batched-clips 0 > $O/kt64.log 2>&1
    KT6=$(find $O/kt64 -name '*kernel_trace.csv' | head -1)
    for c in FETCH_SIZE WRITE_SIZE; do pmc_pass ${c}_r64 $c; done
    pmc_pass SQ_r64 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE
    python tools/pmc_util.py "$(find $O/pmc_SQ_r64 -name '*counter_collection.csv' | head -1)" --launches $NL6 \
        --fetch "$(find $O/pmc_FETCH_SIZE_r64 -name '*counter_collection.csv' | head -1)" --write "$(find $O/pmc_WRITE_SIZE_r64 -name '*counter_collection.csv' | head -1)" \
        --trace $KT6 --json $O/pmc_util_r64.json > $O/r${NN}_pmc_util_res64.txt 2>&1
    python tools/update_pmc_traffic.py $O/pmc_util_r64.json "round $NN (tools/make_profiles.sh)" "${MTV_COMMIT:-$(git rev-parse --short HEAD 2>/dev/null)}" R64; cp profiles/pmc_traffic.json $O/pmc_traffic.json
    rm -rf $O/kt64 $O/pmc_SQ_r64 $O/pmc_FETCH_SIZE_r64 $O/pmc_WRITE_SIZE_r64
    BARGS=""
fi
# (with the CPU leg: BASELINE.md section 3 asks for >= 3 timed steps of the oracle at this geometry; bench.py stops it after 60 s)
timeout 600 python bench.py --res 64 --steps 150 --warmup 15 --batched-clips 0 > $O/r${NN}_bench_res64_n1.json 2>/dev/null
# 7. the autoencoder steps either side of the loop (informational): per-launch tables + rocprofv3 kernel stats of decode
timeout 300 python tools/ae_profile.py > $O/r${NN}_ae_decode_per_launch.txt 2>/dev/null
timeout 300 python tools/ae_profile.py --extract > $O/r${NN}_ae_extract_per_launch.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktae -o kt -- python tools/ae_profile.py > /dev/null 2>&1
cp "$(find $O/ktae -name '*kernel_stats.csv' | head -1)" $O/r${NN}_ae_decode_rocprofv3_kernel_stats.csv
# 8. the in-kernel phase anatomy of every conv and every attention launch of a step (diagnostic build with s_memtime
#    stamps: MTV_BUILD_STAMP=1 bash moditalker_amd/csrc/build.sh)
#    (a twin older than the library lacks its newest C-ABI symbols and fails to load: rebuild it first)
[ moditalker_amd/csrc/libmtv_hip_stamp.so -nt moditalker_amd/csrc/plan.hip ] || MTV_BUILD_STAMP=1 bash moditalker_amd/csrc/build.sh > $O/build_stamp.log 2>&1
MTV_LIB=$PWD/moditalker_amd/csrc/libmtv_hip_stamp.so MTV_STAMPS=1 timeout 300 python tools/stamps.py > $O/stamps_all.txt 2>$O/stamps.err
[ -s $O/stamps_all.txt ] || { echo "stamps.py produced nothing:"; tail -5 $O/stamps.err; }
sed -n '/^# attention/,$p' $O/stamps_all.txt > $O/r${NN}_attention_phase_stamps.txt
sed '/^# attention/,$d' $O/stamps_all.txt > $O/r${NN}_conv_phase_stamps.txt
# 9. micro-benchmarks: the launch chain, f32 MFMA vs VALU on one SIMD
timeout 120 tools/ubench/chain > $O/r${NN}_launch_chain_ubench.txt 2>&1
timeout 120 tools/ubench/mfma_valu > $O/r${NN}_mfma_valu_ubench.txt 2>&1
# 10. the kernels of the deep levels on their own (tools/ubench/deep_bench): correctness against plain CPU restatements, the chain
#     experiment (40 dependent convs with distinct weights vs the same chain on k_conv; 16 ops = weights that fit the Infinity Cache),
#     the fused attention + proj_out kernel; the -DMTV_DEEP_STAMP build adds the in-kernel phase anatomy
timeout 300 tools/ubench/deep_bench check > $O/r${NN}_deep_check.txt 2>&1
timeout 300 tools/ubench/deep_bench attn >> $O/r${NN}_deep_check.txt 2>&1
timeout 300 tools/ubench/deep_bench block >> $O/r${NN}_deep_check.txt 2>&1
if [ -x tools/ubench/deep_bench_stamp ]; then DB=tools/ubench/deep_bench_stamp; else DB=tools/ubench/deep_bench; fi
(for cfg in "40 4 2" "40 8 4" "16 4 2" "16 8 4"; do timeout 120 $DB chain $cfg 2>&1 | grep -v "final act"; done) > $O/r${NN}_deep_chain.txt
timeout 120 $DB attn time 2>&1 | grep -v "^ATTN" > $O/r${NN}_deep_attn.txt
# 10a. k_deep_block (csrc/block.hip): the whole attention block in one launch -- 15 shapes against the double-precision CPU block, the
#      graph-chain timing of the base model's shapes and the in-kernel stamps (stage 1 | hand-off | stage 2 | hand-off | attention | proj)
timeout 300 $DB block time > $O/r${NN}_deep_block.txt 2>&1
# 10b. the level-0/1 kernels in chains: k_conv tiles vs k_conv_win tiles (20 dependent 3x3 convs) and vs k_conv_pw tiles (20 qkv-shaped 1x1
#      launches), with stamps (the r04_conv_win_stamps.txt / r04_conv_pw_stamps.txt of round 4 are these commands run across kernel versions)
(timeout 120 $DB win 20 32 16 128; timeout 120 $DB win 20 16 8 256) > $O/r${NN}_conv_win_chain.txt 2>&1
(timeout 120 $DB pw 32 16 128 384; timeout 120 $DB pw 16 8 256 768; timeout 120 $DB pw 16 8 512 1536) > $O/r${NN}_conv_pw_chain.txt 2>&1
# 10c. round 6: one real level-0 stage (in1) as eight launches and as ONE persistent launch, same phase bodies, both checked against double CPU math
#      (tools/ubench/stage_bench.hip): us per stage of each form, the in-launch edges' anatomy, per-phase cycles
if [ -x tools/ubench/stage_bench ]; then
    (timeout 300 tools/ubench/stage_bench check | tail -9; timeout 200 tools/ubench/stage_bench time 12) > $O/r${NN}_stage_bench.txt 2>&1
fi
# 11. one page of numbers, written by a script from the files above (no hand-typed figures)
python tools/round_summary.py $NN $O > $O/r${NN}_summary.md 2>$O/summary.err
rm -rf $O/kt $O/ktae $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -25 $O/r${NN}_step_summary.txt
