export MTV_TUNE_CACHE=/tmp/tune.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 250 --warmup 25 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k:(v['ms_per_step'],v['launches']) for k,v in d['families'].items() if v['launches']>1})"
python tools/profile_ops.py > gpurun_out/ops_r1_v10.txt 2>&1
grep attn gpurun_out/ops_r1_v10.txt | sort -rn | awk 'NR%3==1' | head -8
