#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gemm_split_gpu.py tests/test_pose_gpu.py tests/test_matching_models_gpu.py tests/test_pipeline_gpu.py -q -s > "$OUT/job3_tests.log" 2>&1
grep -n "planted pair\|demo pair pose\|pose vs the reference\|passed\|failed\|FAILED\|dynamic range" "$OUT/job3_tests.log" | cut -c1-420
LCR_GEMM_BATCH_SHORT=0 timeout 600 python tools/pair_bench.py --pairs-per-call 8 16 --pairs 192 --repeats 5 > "$OUT/pair_bench_batchdeep.log" 2>&1
tail -1 "$OUT/pair_bench_batchdeep.log" | cut -c1-1500
timeout 600 python tools/pair_bench.py --pairs-per-call 1 8 16 32 --pairs 192 --repeats 5 > "$OUT/pair_bench_default.log" 2>&1
tail -1 "$OUT/pair_bench_default.log" | cut -c1-2500
timeout 900 python bench.py > "$OUT/bench_default.log" 2> "$OUT/bench_default.err"
tail -1 "$OUT/bench_default.log" | cut -c1-3000
