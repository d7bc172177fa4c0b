cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 300 python tools/shape_times.py > gpurun_out/r4/shape_times.txt 2>&1; grep -E "total" gpurun_out/r4/shape_times.txt
for v in "default" "pw3" "enc3" "enc1" "nk"; do
  case $v in
    default) E=""; A="";;
    pw3) E=""; A="--pre-workers 3";;
    enc3) E="LCR_ENC_STREAMS=3"; A="";;
    enc1) E=""; A="--single-encoder";;
    nk) E="LCR_BENCH_NO_KTIMER=1"; A="";;
  esac
  env $E LCR_PIPE_STATS=1 timeout 600 python bench.py --no-cpu-baseline --repeats 3 $A > gpurun_out/r4/bench_$v.json 2> gpurun_out/r4/bench_$v.err
  echo "$v: $(python -c "import json;d=json.load(open('gpurun_out/r4/bench_$v.json'));print(d['value'], d['ms_per_step'])") $(grep 'pipeline host' gpurun_out/r4/bench_$v.err | tail -1)"
done
