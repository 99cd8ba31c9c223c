export MTV_TUNE_CACHE=/tmp/tune.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['value'], d['ms_per_step'])"
MTV_EAGER=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MTV_EAGER=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_eager -o e -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
