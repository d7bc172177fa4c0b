#!/bin/bash
# round-5 fuzz campaign over the final kernels (the search kernel's turn was refactored this round, the aggregation got its early exit)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 700 python -m pytest tests/test_pose_chain_gpu.py tests/test_pairs_batched_gpu.py tests/test_encoder_gpu.py tests/test_matching_models_gpu.py -q > "$OUT/fuzz_pretests.log" 2>&1
grep -n "passed\|failed\|FAILED" "$OUT/fuzz_pretests.log" | tail -6
timeout 620 python tools/fuzz_ops.py 80000 200000 --json "$OUT/r05_fuzz_ops.jsonl" --max-seconds 600 > "$OUT/fuzz_ops.log" 2>&1
tail -1 "$OUT/r05_fuzz_ops.jsonl" | cut -c1-400
timeout 420 python tools/fuzz_collate.py 30000 90000 --max-seconds 400 --json "$OUT/r05_fuzz_collate.jsonl" > "$OUT/fuzz_collate.log" 2>&1
timeout 320 python tools/fuzz_collate.py 90000 120000 dense --max-seconds 300 --json "$OUT/r05_fuzz_collate.jsonl" > "$OUT/fuzz_collate_dense.log" 2>&1
cat "$OUT/r05_fuzz_collate.jsonl" | cut -c1-400
timeout 320 python tools/fuzz_degenerate_gpu.py 30000 90000 --json "$OUT/r05_fuzz_degenerate.jsonl" --max-seconds 300 > "$OUT/fuzz_degenerate.log" 2>&1
tail -1 "$OUT/r05_fuzz_degenerate.jsonl" | cut -c1-400
timeout 500 python tools/pair_bench.py --pairs-per-call 8 16 32 --pairs 192 --repeats 5 > "$OUT/pair_bench_glue.log" 2>&1
tail -1 "$OUT/pair_bench_glue.log" | cut -c1-1200
