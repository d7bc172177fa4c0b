#!/bin/bash
# One command for the day an N-GPU MI355X node is available (the reference's multi-process entry is utils/engine/base_tester.py:88):
#   1. bench.py --gpus {1,2,4,8} --no-cpu-baseline        -> the scans/s curve (weak scaling, descriptors all-gathered over RCCL in the timed region)
#   2. tools/loop_detection_run.py --frames 23201 --gpus N -> BASELINE configs[3]: KITTI 00-10 corpus sharded by contiguous frame ranges,
#                                                            one all_gather_into_tensor of [n_r,256], per-rank masked top-50
#   3. tools/pair_bench.py --gpus N                        -> BASELINE configs[4]: registration pairs, replicas + one all-reduce of the sums
# and checks in every multi-rank log that RCCL really formed an N-rank communicator (NCCL_DEBUG=INFO: "Init COMPLETE ... nranks N").
# Writes ONE JSON (default gpurun_out/scale_run.json; copy it to profiles/rNN_scale_run.json).
#
#   tools/scale_run.sh [MAX_GPUS]                  real run; MAX_GPUS defaults to the number of visible GPUs
#   LCR_SCALE_DRY=1 tools/scale_run.sh 8          dry run of every code path on ONE GPU (all ranks on device 0, gloo instead of RCCL:
#                                                  RCCL refuses two ranks on one device) — numbers are meaningless, the launch / binding /
#                                                  collective / output plumbing is what is exercised
# Host side: every rank pins itself to its GPU's NUMA share (bench.bind_rank) and the math libraries' pools are capped HERE so that
# 8 ranks x ~6 busy host threads fit their share: OMP/MKL threads = 4 per rank.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=${LCR_SCALE_OUT:-$ROOT/gpurun_out/scale_run.json}
LOGD=$(dirname "$OUT")/scale_logs
mkdir -p "$LOGD"
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4 MKL_NUM_THREADS=4
DRY=${LCR_SCALE_DRY:-0}
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
MAXG=${1:-$HAVE}
if [ "$DRY" = "1" ]; then
  export LCR_BENCH_SINGLE_DEVICE=1 LCR_BENCH_BACKEND=gloo
  STEPS=${LCR_SCALE_STEPS:-6}; REPEATS=1; FRAMES=${LCR_SCALE_FRAMES:-320}; PAIRS=${LCR_SCALE_PAIRS:-16}
else
  export NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT
  STEPS=${LCR_SCALE_STEPS:-200}; REPEATS=5; FRAMES=${LCR_SCALE_FRAMES:-23201}; PAIRS=${LCR_SCALE_PAIRS:-192}
  if [ "$MAXG" -gt "$HAVE" ]; then echo "scale_run: $MAXG GPUs asked, $HAVE visible" >&2; MAXG=$HAVE; fi
fi
TMO=${LCR_SCALE_TIMEOUT:-1500}
run() {  # name, n, command...  -> $LOGD/name.out (stdout: the JSON line), $LOGD/name.err
  local name=$1; shift
  timeout "$TMO" "$@" > "$LOGD/$name.out" 2> "$LOGD/$name.err"
  echo $? > "$LOGD/$name.rc"
}
NS=""
for n in 1 2 4 8; do [ "$n" -le "$MAXG" ] && NS="$NS $n"; done
for n in $NS; do
  run bench_$n python bench.py --gpus "$n" --steps "$STEPS" --warmup 3 --repeats "$REPEATS" --no-cpu-baseline
done
# BASELINE configs[3] as a STRONG-scaling run: the same fixed corpus at every N (wall time of descriptors + exchange + retrieval, per-rank
# retrieval time with the work-balanced query split of lcrnet_amd.retrieval.search_range); the gathered descriptors and the top-k rows must
# be the bytes N = 1 produced (sha1 in every line; the dry run also dumps the arrays and compares them value by value)
for n in $NS; do
  DUMP=""; [ "$DRY" = "1" ] && DUMP="--dump $LOGD/loop_$n.npz"
  run loop_$n python tools/loop_detection_run.py --frames "$FRAMES" --gpus "$n" $DUMP
done
run pairs_$MAXG python tools/pair_bench.py --gpus "$MAXG" --pairs "$PAIRS" --pairs-per-call 8 --repeats 3
python - "$OUT" "$LOGD" "$MAXG" "$DRY" $NS <<'PY'
import json, os, re, sys
out, logd, maxg, dry, ns = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] == "1", [int(x) for x in sys.argv[5:]]
def last_json(name):
    try:
        lines = [l for l in open(os.path.join(logd, name + ".out")).read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None
def rc(name):
    try:
        return int(open(os.path.join(logd, name + ".rc")).read())
    except Exception:
        return None
def rccl_ranks(name, n):
    """ranks RCCL reports for the communicators of this run: every 'Init COMPLETE' line of NCCL_DEBUG=INFO carries 'nranks N'"""
    try:
        err = open(os.path.join(logd, name + ".err")).read()
    except Exception:
        return None
    seen = [int(m) for m in re.findall(r"Init COMPLETE.*?nranks (\d+)", err)]
    return {"init_complete_lines": len(seen), "nranks_seen": sorted(set(seen)), "ok": bool(seen) and max(seen) == n}
rec = {"tool": "tools/scale_run.sh", "dry_run_single_device_gloo": dry, "max_gpus": maxg, "curve": [], "checks": {}}
base = None
for n in ns:
    j = last_json("bench_%d" % n)
    e = {"n_gpus": n, "rc": rc("bench_%d" % n), "value": j and j.get("value"), "unit": j and j.get("unit"), "ms_per_step": j and j.get("ms_per_step"),
         "parallelism": j and j.get("config", {}).get("parallelism"), "host": j and j.get("config", {}).get("host")}
    if n == 1 and j:
        base = j.get("value")
    if base and j:
        e["value_over_n_times_1gpu"] = round(j["value"] / (n * base), 4)       # informational; the driver computes its own
    if n > 1 and not dry:
        e["rccl"] = rccl_ranks("bench_%d" % n, n)
    rec["curve"].append(e)
lj, pj = last_json("loop_%d" % maxg), last_json("pairs_%d" % maxg)
rec["loop_detection"] = {"rc": rc("loop_%d" % maxg), "line": lj, "rccl": None if dry or maxg == 1 else rccl_ranks("loop_%d" % maxg, maxg)}
strong, l1 = [], last_json("loop_1")
for n in ns:
    j = last_json("loop_%d" % n)
    e = {"n_gpus": n, "rc": rc("loop_%d" % n), "frames": j and j.get("frames"), "wall_s": j and j.get("wall_s_descriptors_plus_retrieval"),
         "descriptor_s": j and j.get("descriptor_s"), "retrieval_ms_per_rank": j and j.get("retrieval_ms_per_rank"),
         "query_rows_per_rank": j and j.get("query_rows_per_rank"),
         "descriptors_identical_to_1gpu": bool(j and l1 and j.get("gathered_descriptors_sha1") == l1.get("gathered_descriptors_sha1")),
         "rows_identical_to_1gpu": bool(j and l1 and j.get("rows_sha1") == l1.get("rows_sha1"))}
    if l1 and j and l1.get("wall_s_descriptors_plus_retrieval"):
        e["speedup_over_1gpu"] = round(l1["wall_s_descriptors_plus_retrieval"] / j["wall_s_descriptors_plus_retrieval"], 3)
    if dry and n > 1:
        try:
            import numpy as np
            a, b = np.load(os.path.join(logd, "loop_1.npz")), np.load(os.path.join(logd, "loop_%d.npz" % n))
            e["dump_descriptors_max_abs_diff"] = float(np.abs(a["desc"] - b["desc"]).max())
            e["dump_rows_equal"] = bool(a["rows"].shape == b["rows"].shape and np.array_equal(a["rows"][..., :2], b["rows"][..., :2]))
        except Exception as ex:
            e["dump_error"] = str(ex)
    strong.append(e)
rec["strong_scaling_configs3"] = strong
rec["pairs"] = {"rc": rc("pairs_%d" % maxg), "line": pj, "rccl": None if dry or maxg == 1 else rccl_ranks("pairs_%d" % maxg, maxg)}
rec["checks"]["all_rc_zero"] = all(e["rc"] == 0 for e in rec["curve"]) and all(e["rc"] == 0 for e in strong) and rec["pairs"]["rc"] == 0
rec["checks"]["strong_scaling_rows_identical_to_1gpu"] = all(e["rows_identical_to_1gpu"] for e in strong)
rec["checks"]["strong_scaling_descriptors_identical_to_1gpu"] = all(e["descriptors_identical_to_1gpu"] for e in strong)
rec["checks"]["rccl_saw_all_ranks"] = None if dry else all((e.get("rccl") or {"ok": True})["ok"] for e in rec["curve"])
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({"scale_run": out, "curve": [(e["n_gpus"], e["value"]) for e in rec["curve"]], "checks": rec["checks"]}))
PY
