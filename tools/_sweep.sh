cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | $2: "; env $1 LCR_BENCH_NO_KTIMER=1 timeout 300 python bench.py --no-cpu-baseline --repeats 3 $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])"; }
run "A=1" ""
run "LCR_ENC_STREAMS=3" ""
run "A=1" "--pre-workers 3"
run "A=1" "--pre-workers 1"
run "A=1" "--depth 3"
run "LCR_ENC_STREAMS=3" "--pre-workers 3 --depth 3"
run "A=1" "--no-upsampling"
run "LCR_BENCH_BATCH=16" ""
