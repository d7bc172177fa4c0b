#!/bin/bash
# Round profile set (run on the GPU box through gpurun; outputs under gpurun_out/prof, copy what is judged into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command          -> kernel_summary.md
#   2. PMC FETCH_SIZE / WRITE_SIZE of the bench command, separate passes       -> pmc_traffic.{md,json}
#   3. PMC MFMA / wave-state counters of the encoder alone                     -> pmc_mfma_encoder.md
#   4. rocprofv3 --kernel-trace --stats of the encoder alone                   -> encoder_alone_kernel_summary.md
#   5. PMC instruction counts per kernel family over a pass (10 searches)      -> pmc_insts.md
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o stats -- python "$ROOT/bench.py" --steps 50 --repeats 2 --no-cpu-baseline --no-blocks > "$OUT/stats_bench.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" "$(find /tmp/prof_s -name "*.db" | head -1)" --bench-log "$OUT/stats_bench.log" \
  --note "rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --repeats 2 --no-cpu-baseline --no-blocks (timed region + the clocked kernel-alone passes behind it)" > "$OUT/kernel_summary.md"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o fetch -- python "$ROOT/bench.py" --steps 25 --repeats 1 --no-cpu-baseline --no-blocks > "$OUT/fetch_bench.log" 2>&1
cp $(find /tmp/prof_f -name "fetch_counter_collection.csv" | head -1) "$OUT/"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o write -- python "$ROOT/bench.py" --steps 25 --repeats 1 --no-cpu-baseline --no-blocks > "$OUT/write_bench.log" 2>&1
cp $(find /tmp/prof_w -name "write_counter_collection.csv" | head -1) "$OUT/"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/prof_m -o m -- python "$ROOT/tools/enc_profile.py" 3 > "$OUT/mfma_enc.log" 2>&1
python "$ROOT/tools/pmc_mfma_summary.py" "$(find /tmp/prof_m -name "m_counter_collection.csv" | head -1)" > "$OUT/pmc_mfma_encoder.md"
rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o enc -- python "$ROOT/tools/enc_profile.py" 20 > "$OUT/enc_alone.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" "$(find /tmp/prof_e -name "*.db" | head -1)" \
  --note "rocprofv3 --kernel-trace --stats -- python tools/enc_profile.py 20: 20 encoder passes over the bench batch on one stream, nothing else on the GPU (what roofline.achieved / frac / avg_launch_us of the bench line measure live)" > "$OUT/encoder_alone_kernel_summary.md"
LCR_ENC_PROFILE_UPSAMPLING=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/prof_i -o i -- python "$ROOT/tools/enc_profile.py" 4 > "$OUT/insts_enc.log" 2>&1
python "$ROOT/tools/pmc_insts_summary.py" "$(find /tmp/prof_i -name "i_counter_collection.csv" | head -1)" 4 4 > "$OUT/pmc_insts.md"
cd "$ROOT"
python tools/pmc_summary.py "$OUT/fetch_counter_collection.csv" "$OUT/write_counter_collection.csv" "$OUT/pmc_traffic" > /dev/null
tail -1 "$OUT/stats_bench.log" | cut -c1-300
head -24 "$OUT/kernel_summary.md" | cut -c1-170
cat "$OUT/pmc_mfma_encoder.md" | head -16
head -14 "$OUT/pmc_traffic.md"
head -14 "$OUT/pmc_insts.md"
