cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err; tail -3 gpurun_out/r5/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])
for k in ('kernel','achieved','frac','avg_launch_us','launches_timed','achieved_event_bracketed','achieved_alone','frac_alone','gflop_per_step','kernel_time_over_step_time'): print(' ',k, r.get(k))
print(' neighbor', {k:v for k,v in r['neighbor'].items() if k!='kernel'})
print(' agg', r['aggregation']); print(' whole', r['whole_step']); print(' secondary', r['secondary'])
PY
