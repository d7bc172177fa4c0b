set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_bench_multirank_gpu.py tests/test_pipeline_gpu.py tests/test_encoder_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; tail -c 2500 gpurun_out/r2/bench.json
LCR_BENCH_NO_KTIMER=1 timeout 600 python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/r2/bench_notimer.json 2>> gpurun_out/r2/bench.err; cut -c1-300 gpurun_out/r2/bench_notimer.json
