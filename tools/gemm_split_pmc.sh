#!/bin/bash
# PMC counters of the split-bf16 GEMM variants vs the fp32 K-deep kernel on the encoder's deep shapes (gpurun; output under gpurun_out/split_pmc).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/split_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag -o p -- python "$ROOT/tools/gemm_split_bench.py" > "$OUT/run_$tag.log" 2>&1
  f=$(find /tmp/pmc_$tag -name "p_counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/$tag.csv"
done
python "$ROOT/tools/pmc_kernel.py" --match gemm "$OUT"/*.csv > "$OUT/summary.md"
cat "$OUT/summary.md"
