#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/r05_gpu_tests_default2.log" 2>&1
grep -n "passed\|failed\|FAILED" "$OUT/r05_gpu_tests_default2.log" | tail -8
LCR_GEMM_SPLIT=0 timeout 1200 python -m pytest tests -m gpu -q > "$OUT/r05_gpu_tests_fp32.log" 2>&1
grep -n "passed\|failed\|FAILED" "$OUT/r05_gpu_tests_fp32.log" | tail -8
timeout 1500 bash tools/profile_round.sh > "$OUT/profile_round.log" 2>&1
tail -60 "$OUT/profile_round.log" | cut -c1-200
LCR_SCALE_DRY=1 LCR_SCALE_OUT=$ROOT/gpurun_out/r05/r05_scale_run_dry.json timeout 1200 bash tools/scale_run.sh 8 > "$OUT/scale_dry.log" 2>&1
tail -3 "$OUT/scale_dry.log" | cut -c1-600
