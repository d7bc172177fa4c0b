#!/bin/bash
# Profiles of the default bench workload, as committed under profiles/ (run on the GPU box through gpurun):
#   kernel trace + stats  -> tools/rocprof_summary.py -> gpurun_out/prof/kernel_summary.md (-> profiles/rNN_bench_kernel_summary.md)
#   PMC FETCH_SIZE / WRITE_SIZE in two separate passes -> tools/pmc_summary.py -> profiles/rNN_pmc_traffic.{md,json}
# rocprofv3 wants a writable TMPDIR and must not combine --pmc with the sys/hip traces.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o stats -- python "$ROOT/bench.py" --steps 50 --no-cpu-baseline > "$OUT/stats_bench.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" "$(find /tmp/prof_s -name "*.db" | head -1)" > "$OUT/kernel_summary.md"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o fetch -- python "$ROOT/bench.py" --steps 25 --no-cpu-baseline > "$OUT/fetch_bench.log" 2>&1
cp $(find /tmp/prof_f -name "fetch_counter_collection.csv" | head -1) "$OUT/"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o write -- python "$ROOT/bench.py" --steps 25 --no-cpu-baseline > "$OUT/write_bench.log" 2>&1
cp $(find /tmp/prof_w -name "write_counter_collection.csv" | head -1) "$OUT/"
cd "$ROOT"
python tools/pmc_summary.py "$OUT/fetch_counter_collection.csv" "$OUT/write_counter_collection.csv" "$OUT/pmc_traffic" > /dev/null
tail -1 "$OUT/stats_bench.log" | cut -c1-200
ls -la "$OUT"
