#!/bin/bash
# Round-5 opening GPU job (gpurun): what round 4 ran out of GPU minutes for, plus this round's baselines.
#   1. full -m gpu suite with LCR_GEMM_SPLIT=1                                  -> r05_gpu_tests_split.log
#   2. pair model (configs[4]) kernel trace + MFMA counters at 8 pairs per call  -> r05_pair_model_kernel_summary.md, r05_pmc_mfma_pair.md
#   3. float-parity fuzz, split form seeds 0..300, fp32 form seeds 300..520      -> r05_fuzz_float_parity.jsonl
#   4. radius bench baseline (ten searches as single launches)                   -> r05_radius_bench_base.log
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p "$OUT"
cd "$ROOT"
LCR_GEMM_SPLIT=1 timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/r05_gpu_tests_split.log" 2>&1
tail -3 "$OUT/r05_gpu_tests_split.log"
timeout 300 python tools/radius_bench.py > "$OUT/r05_radius_bench_base.log" 2>&1
tail -25 "$OUT/r05_radius_bench_base.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o pair -- python "$ROOT/tools/pair_bench.py" --pairs-per-call 8 --pairs 96 --repeats 2 > "$OUT/pair_trace.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" "$(find /tmp/prof_p -name "*.db" | head -1)" \
  --note "rocprofv3 --kernel-trace --stats -- python tools/pair_bench.py --pairs-per-call 8 --pairs 96 --repeats 2 (warm-up passes included)" > "$OUT/r05_pair_model_kernel_summary.md"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/prof_pm -o m -- python "$ROOT/tools/pair_bench.py" --pairs-per-call 8 --pairs 48 --repeats 1 > "$OUT/pair_pmc.log" 2>&1
python "$ROOT/tools/pmc_mfma_summary.py" "$(find /tmp/prof_pm -name "m_counter_collection.csv" | head -1)" > "$OUT/r05_pmc_mfma_pair.md"
head -30 "$OUT/r05_pair_model_kernel_summary.md" | cut -c1-160
cd "$ROOT"
LCR_GEMM_SPLIT=1 timeout 800 python tools/fuzz_float_parity_gpu.py 0 300 --json "$OUT/r05_fuzz_float_parity.jsonl" --max-seconds 700 > "$OUT/fuzz_split.log" 2>&1
timeout 600 python tools/fuzz_float_parity_gpu.py 300 520 --json "$OUT/r05_fuzz_float_parity.jsonl" --max-seconds 500 > "$OUT/fuzz_fp32.log" 2>&1
cat "$OUT/r05_fuzz_float_parity.jsonl"
