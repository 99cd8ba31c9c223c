timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/profile_ops.py > gpurun_out/ops_r1_v5.txt 2>&1
grep attn gpurun_out/ops_r1_v5.txt | sort -rn | awk 'NR%3==1' | head -16
