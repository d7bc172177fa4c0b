#!/bin/bash
# pair model (configs[4]) kernel trace + MFMA counters at 8 pairs per call on the round's FINAL code
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o pair -- python "$ROOT/tools/pair_bench.py" --pairs-per-call 8 --pairs 96 --repeats 2 > "$OUT/pair_trace2.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" "$(find /tmp/prof_p -name "*.db" | head -1)" \
  --note "rocprofv3 --kernel-trace --stats -- python tools/pair_bench.py --pairs-per-call 8 --pairs 96 --repeats 2 (warm-up passes included; three workers; final code of round 5)" > "$OUT/r05_pair_model_kernel_summary.md"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/prof_pm -o m -- python "$ROOT/tools/pair_bench.py" --pairs-per-call 8 --pairs 48 --repeats 1 > "$OUT/pair_pmc2.log" 2>&1
python "$ROOT/tools/pmc_mfma_summary.py" "$(find /tmp/prof_pm -name "m_counter_collection.csv" | head -1)" > "$OUT/r05_pmc_mfma_pair.md"
head -36 "$OUT/r05_pair_model_kernel_summary.md" | cut -c1-150
