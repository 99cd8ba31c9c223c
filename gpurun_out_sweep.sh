for t in 1,1,1 1,1,4 1,2,1 1,2,2 1,4,1 1,4,4 1,4,16 2,4,1 2,4,8 4,4,1 4,4,2 4,4,8; do
  echo "== tile $t"; MTV_FORCE_TILE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_and_taps" 2>&1 | grep -E "passed|failed|taps off|Error|error" | cut -c1-300
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -30
