"""GPU: the N-rank code path of bench.py on ONE device (LCR_BENCH_SINGLE_DEVICE=1, gloo: RCCL refuses two ranks on one GPU): rank
processes started by bench.py itself, bound to their core ranges, rendezvous on 127.0.0.1, all-gather of the descriptors inside the
timed region, one JSON line from rank 0 with n_gpus = 2.  (the reference's multi-process entry is
utils/engine/base_tester.py:88.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_device():
    env = dict(os.environ, LCR_BENCH_SINGLE_DEVICE="1", LCR_BENCH_RANK_TIMEOUT="600")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # only rank 0 writes to stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["repeats"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["ms_per_step_min"] <= line["ms_per_step"] <= line["ms_per_step_max"]
    assert "gloo" in line["config"]["parallelism"]


def test_one_rank_rccl_group_runs_the_n_rank_path():
    """LCR_BENCH_FORCE_DIST=1: a one-rank RCCL communicator, and with it everything an N-rank run does around the pipeline — barriers,
    `all_gather_into_tensor` issued on the encoder's external HIP stream inside the timed region, the max-over-ranks all-reduce of the block
    times, every secondary block — against real RCCL on this single-GPU box (two ranks on one device are refused by RCCL)."""
    env = dict(os.environ, LCR_BENCH_FORCE_DIST="1", MASTER_PORT="29631")
    env.pop("LCR_BENCH_SINGLE_DEVICE", None)
    env.pop("LCR_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["forced_one_rank_process_group"] == "nccl" and line["n_gpus"] == 1 and line["value"] > 0
    assert "nccl" in line["config"]["parallelism"]
    blocks = ["with_h2d", "descriptor_only_7_searches"]
    # the A/B block times the form the headline does NOT use (default headline: split-bf16 K-deep GEMMs)
    blocks.append("split_bf16_gemm_ab" if os.environ.get("LCR_GEMM_SPLIT", "1") in ("", "0") else "true_fp32_gemm_ab")
    assert ("bf16 x 3 split" in line["dtype"]) == (os.environ.get("LCR_GEMM_SPLIT", "1") not in ("", "0"))
    for k in blocks:
        assert line[k]["value"] > 0
    assert line["with_h2d"]["descriptors_max_abs_diff_vs_resident"] == 0.0
