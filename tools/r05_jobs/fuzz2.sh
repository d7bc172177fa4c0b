#!/bin/bash
# second fuzz campaign of round 5 (final kernels): longer runs, new seed ranges
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 920 python tools/fuzz_ops.py 100000 300000 --json "$OUT/r05_fuzz_ops.jsonl" --max-seconds 900 > "$OUT/fuzz_ops3.log" 2>&1
timeout 920 python tools/fuzz_collate.py 120000 300000 --max-seconds 900 --json "$OUT/r05_fuzz_collate.jsonl" > "$OUT/fuzz_collate3.log" 2>&1
timeout 620 python tools/fuzz_collate.py 300000 400000 dense --max-seconds 600 --json "$OUT/r05_fuzz_collate.jsonl" > "$OUT/fuzz_collate_dense3.log" 2>&1
timeout 620 python tools/fuzz_degenerate_gpu.py 60000 300000 --json "$OUT/r05_fuzz_degenerate.jsonl" --max-seconds 600 > "$OUT/fuzz_degenerate3.log" 2>&1
timeout 920 python tools/fuzz_float_parity_gpu.py 1100 2000 --json "$OUT/r05_fuzz_float_parity.jsonl" --max-seconds 900 > "$OUT/fuzz_fp3.log" 2>&1
tail -2 "$OUT/r05_fuzz_ops.jsonl" "$OUT/r05_fuzz_collate.jsonl" "$OUT/r05_fuzz_degenerate.jsonl" "$OUT/r05_fuzz_float_parity.jsonl" | cut -c1-260
