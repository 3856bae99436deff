#!/bin/bash
# soak: the tests that exercise the several-bodies form with a task in hand, repeated
n=${1:-6}; fail=0
for i in $(seq $n); do
  timeout 900 python -m pytest tests/test_gpu_groups.py tests/test_gpu_full_size.py tests/test_gpu_exact_cuts.py tests/test_gpu_sharded.py -x -q -m gpu > /tmp/soak.log 2>&1 || { fail=$((fail+1)); grep -E "timed out|assert|Error|FAILED" /tmp/soak.log | head -5; }
  grep -E "passed|failed" /tmp/soak.log | tail -1
done
echo "failures: $fail of $n"
