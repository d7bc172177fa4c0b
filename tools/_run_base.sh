set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/base
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/base/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/base/pytest.log
tail -3 gpurun_out/base/pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/base/bench.json 2> gpurun_out/base/bench.err; tail -c 1500 gpurun_out/base/bench.json
timeout 300 python tools/gemm_bench.py > gpurun_out/base/gemm.txt 2>&1
timeout 300 python tools/gemm_bench.py --square >> gpurun_out/base/gemm.txt 2>&1
timeout 300 python tools/shape_times.py > gpurun_out/base/shape_times.txt 2>&1
tail -40 gpurun_out/base/gemm.txt
