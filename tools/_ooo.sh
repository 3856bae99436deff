timeout 600 python -m pytest tests/test_gpu_edge_cases.py -q -x -k "two_workgroups or several_bodies" 2>&1 | grep -E "passed|failed|Error|assert" | head
for o in 0 1; do echo "OOO=$o"; AVBD_RESIDENT_OOO=$o timeout 300 python tools/_big.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"; done
