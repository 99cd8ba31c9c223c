cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1b -o b -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_r1b_bench.json 2> $R/gpurun_out/prof_r1b.err
ls $R/gpurun_out/prof_r1b | head
