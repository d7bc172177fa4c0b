#!/bin/bash
# second fuzz campaign of round 6 (final kernels): longer runs, new seed ranges; plus the pair-model goldens under both GEMM forms
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
timeout 920 python tools/fuzz_ops.py 500000 900000 --json "$OUT/r06_fuzz_ops.jsonl" --max-seconds 900 > "$OUT/fuzz_ops2.log" 2>&1
timeout 620 python tools/fuzz_collate.py 600000 900000 --max-seconds 600 --json "$OUT/r06_fuzz_collate.jsonl" > "$OUT/fuzz_collate2.log" 2>&1
timeout 320 python tools/fuzz_collate.py 900000 990000 dense --max-seconds 300 --json "$OUT/r06_fuzz_collate.jsonl" > "$OUT/fuzz_collate_dense2.log" 2>&1
timeout 420 python tools/fuzz_degenerate_gpu.py 500000 900000 --json "$OUT/r06_fuzz_degenerate.jsonl" --max-seconds 400 > "$OUT/fuzz_degenerate2.log" 2>&1
timeout 620 python tools/fuzz_float_parity_gpu.py 3000 4000 --json "$OUT/r06_fuzz_float_parity.jsonl" --max-seconds 600 > "$OUT/fuzz_fp2.log" 2>&1
LCR_GEMM_SPLIT=0 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" > "$OUT/gpu_tests_fp32_mfma.log"
cat "$OUT/gpu_tests_fp32_mfma.log"
tail -n 1 "$OUT/r06_fuzz_ops.jsonl" "$OUT/r06_fuzz_collate.jsonl" "$OUT/r06_fuzz_degenerate.jsonl" "$OUT/r06_fuzz_float_parity.jsonl" | cut -c1-300
