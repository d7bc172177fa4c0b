#!/bin/bash
# lane-per-query search kernel: exactness + timing against the wave form (gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
export LCR_RB_NO_CELL_ORDER=1
LCR_RS_LPQ=0 LCR_RB_CHECK=1 timeout 300 python tools/radius_bench.py > "$OUT/lpq_rb_wave.log" 2>&1
LCR_RS_LPQ=2 LCR_RB_CHECK=1 timeout 300 python tools/radius_bench.py > "$OUT/lpq_rb_stats.log" 2>&1
LCR_RS_LPQ=1 LCR_RB_CHECK=1 timeout 300 python tools/radius_bench.py > "$OUT/lpq_rb_lpq.log" 2>&1
grep -h "ordered\|CHECK\|lpq stats\|total" "$OUT/lpq_rb_wave.log" | head -20
echo ---- stats; grep -h "CHECK\|lpq stats\|check" "$OUT/lpq_rb_stats.log" | head -16
echo ---- lpq; grep -h "ordered\|CHECK\|total\|rror" "$OUT/lpq_rb_lpq.log" | head -24
LCR_RS_LPQ=1 LCR_RS_LPQ_ANY=1 LCR_PRE_SPLIT_SEARCHES=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py -q -x > "$OUT/lpq_tests.log" 2>&1
tail -5 "$OUT/lpq_tests.log"
