cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_integration_stub_gpu.py tests/test_bench_multirank_gpu.py -x -q 2>&1 | tail -4
echo "== 8 ranks on one device (gloo): nproc=$(nproc)"
( time LCR_BENCH_SINGLE_DEVICE=1 LCR_BENCH_RANK_TIMEOUT=900 timeout 1000 python bench.py --gpus 8 --steps 5 --warmup 1 --repeats 2 --no-cpu-baseline > gpurun_out/r6/bench_8rank.json 2> gpurun_out/r6/bench_8rank.err ) 2>&1 | tail -3
echo "rc=$?"; cut -c1-600 gpurun_out/r6/bench_8rank.json; tail -5 gpurun_out/r6/bench_8rank.err
echo "== 8 ranks via torch.distributed.run"
( time LCR_BENCH_SINGLE_DEVICE=1 timeout 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 5 --warmup 1 --repeats 2 --no-cpu-baseline > gpurun_out/r6/bench_8rank_torchrun.json 2> gpurun_out/r6/bench_8rank_torchrun.err ) 2>&1 | tail -3
grep -c '^{' gpurun_out/r6/bench_8rank_torchrun.json; grep '^{' gpurun_out/r6/bench_8rank_torchrun.json | cut -c1-300
