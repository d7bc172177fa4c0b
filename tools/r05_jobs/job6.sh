#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_transformer_gpu.py -q -s > "$OUT/job6_tests.log" 2>&1
grep -n "topk\|ThDRoFormer\|passed\|failed\|FAILED\|Error" "$OUT/job6_tests.log" | cut -c1-300 | tail -20
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
LCR_GEMM_BATCH_TILE=128 timeout 500 python tools/pair_bench.py --pairs-per-call 16 --pairs 192 --repeats 5 > "$OUT/pair_bench_tile128.log" 2>&1
tail -1 "$OUT/pair_bench_tile128.log" | cut -c1-700
