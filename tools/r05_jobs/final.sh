#!/bin/bash
# round-5 closing job: the profile set and the bench line on the final code, more random parity evidence
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python bench.py > "$OUT/bench_final.log" 2> "$OUT/bench_final.err"
tail -1 "$OUT/bench_final.log" | cut -c1-400
timeout 1500 bash tools/profile_round.sh > "$OUT/profile_round_final.log" 2>&1
tail -3 "$OUT/profile_round_final.log" | cut -c1-200
timeout 700 python tools/fuzz_float_parity_gpu.py 640 1100 --json "$OUT/r05_fuzz_float_parity.jsonl" --max-seconds 600 > "$OUT/fuzz_fp_final.log" 2>&1
tail -1 "$OUT/r05_fuzz_float_parity.jsonl" | cut -c1-300
timeout 620 python tools/fuzz_ops.py 90000 200000 --json "$OUT/r05_fuzz_ops.jsonl" --max-seconds 600 > "$OUT/fuzz_ops2.log" 2>&1
tail -1 "$OUT/r05_fuzz_ops.jsonl" | cut -c1-300
timeout 500 python tools/pair_bench.py --pairs-per-call 1 8 16 32 --pairs 192 --repeats 5 > "$OUT/pair_bench_final.log" 2>&1
tail -1 "$OUT/pair_bench_final.log" | cut -c1-300
