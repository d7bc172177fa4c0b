#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
for w in 2 3 4; do
  timeout 500 python tools/pair_bench.py --pairs-per-call 16 --pairs 192 --repeats 5 --workers $w > "$OUT/pair_bench_w$w.log" 2>&1
  echo "workers $w: $(tail -1 "$OUT/pair_bench_w$w.log" | python -c "import sys,json; l=json.loads(sys.stdin.read()); b=l['by_pairs_per_call']['16']; print(b['pairs_per_s'], b['passes_pairs_per_s'])")"
done
timeout 300 python -m pytest tests/test_pairs_batched_gpu.py tests/test_pose_gpu.py tests/test_matching_models_gpu.py -q 2>&1 | tail -2
