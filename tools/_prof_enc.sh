cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_enc
rocprofv3 --kernel-trace --stats -d /tmp/pe -o enc -- python $R/tools/enc_profile.py 20 > $R/gpurun_out/prof_enc/log.txt 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/pe -name "*.db" | head -1)" > $R/gpurun_out/prof_enc/kernel_summary.md
head -40 $R/gpurun_out/prof_enc/kernel_summary.md
