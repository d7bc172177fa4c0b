cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcg
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/p1 -o a -- python $R/tools/gemm_pmc_case.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d /tmp/p2 -o b -- python $R/tools/gemm_pmc_case.py > /dev/null 2>&1
cp $(find /tmp/p1 -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmcg/a.csv
cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmcg/b.csv
python $R/tools/pmc_by_grid.py $R/gpurun_out/pmcg/a.csv | grep gemm
python $R/tools/pmc_by_grid.py $R/gpurun_out/pmcg/b.csv | grep gemm
