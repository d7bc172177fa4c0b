#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_pose_chain_gpu.py -q -s 2>&1 | grep -n "mutual=\|passed\|failed\|FAILED\|Error" | cut -c1-250
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/r05_gpu_tests_final.log" 2>&1
grep -n "passed\|failed\|FAILED" "$OUT/r05_gpu_tests_final.log" | tail -6
