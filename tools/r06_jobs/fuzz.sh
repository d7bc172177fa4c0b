#!/bin/bash
# fuzz campaign of round 6 on the round's final kernels (limit-1 arg-min path of the radius search, valid-first flag, top-1 candidate rule): new seed ranges
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
timeout 420 python tools/fuzz_ops.py 300000 500000 --json "$OUT/r06_fuzz_ops.jsonl" --max-seconds 400 > "$OUT/fuzz_ops.log" 2>&1
timeout 320 python tools/fuzz_collate.py 400000 600000 --max-seconds 300 --json "$OUT/r06_fuzz_collate.jsonl" > "$OUT/fuzz_collate.log" 2>&1
timeout 220 python tools/fuzz_degenerate_gpu.py 300000 500000 --json "$OUT/r06_fuzz_degenerate.jsonl" --max-seconds 200 > "$OUT/fuzz_degenerate.log" 2>&1
timeout 320 python tools/fuzz_float_parity_gpu.py 2000 3000 --json "$OUT/r06_fuzz_float_parity.jsonl" --max-seconds 300 > "$OUT/fuzz_fp.log" 2>&1
tail -n 1 "$OUT/r06_fuzz_ops.jsonl" "$OUT/r06_fuzz_collate.jsonl" "$OUT/r06_fuzz_degenerate.jsonl" "$OUT/r06_fuzz_float_parity.jsonl" | cut -c1-300
