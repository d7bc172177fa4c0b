#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/r05_gpu_tests_final2.log" 2>&1
grep -n "passed\|failed\|FAILED" "$OUT/r05_gpu_tests_final2.log" | tail -5
timeout 600 python bench.py --no-cpu-baseline > "$OUT/bench_final2.log" 2>/dev/null
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r05/bench_final2.log") if x.startswith("{")][-1])
r=l["roofline"]; print(l["value"], l["ms_per_step"], r["frac"], r["aggregation"]["kernel_ms_per_step"], r["aggregation"]["frac"])
PY
