cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 300 python tools/shape_times.py > gpurun_out/r3/shape_times.txt 2>&1; grep -E "total|K= *(480|960|1920|3840)" gpurun_out/r3/shape_times.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3/bench.json 2> gpurun_out/r3/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/bench.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], 'gemm frac', r['frac'], 'alone', r['frac_alone'], 'bracketed', r['frac_event_bracketed'], 'whole', r['whole_step']['frac_of_fp32_mfma_peak'])
PY
