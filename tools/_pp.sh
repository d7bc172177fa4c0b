cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/ppair -o pair -- python $R/tools/pair_bench.py --pairs-per-call 6 --pairs 120 > /dev/null 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/ppair -name "*.db" | head -1)" > $R/gpurun_out/pair_kernel_summary.md
head -36 $R/gpurun_out/pair_kernel_summary.md | cut -c1-150
