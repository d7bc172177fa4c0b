cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -3
python tools/pre_time.py 100
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pp -o pre -- python $GRAFT_REPO_ROOT/tools/pre_time.py 20 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$(find /tmp/pp -name "*.db" | head -1)" | head -14 | cut -c1-150
