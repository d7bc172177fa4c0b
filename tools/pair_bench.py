#!/usr/bin/env python
"""Pairs/s of the full pair model (BASELINE configs[4]: registration pairs — encoder over the pair stack, 3D-RoFormer, vote
encoder, node / point matching, local-to-global registration) on one GPU; one JSON line.  Not the headline metric.

    python tools/pair_bench.py [--pairs-per-call P ...] [--pairs N] [--gpus G]

--gpus G: one process per GPU (spawned here, or by torch.distributed.run), the pairs dealt round-robin to the ranks, replicas only —
no collective on the data path; the slowest rank's time counts and the registration sums are reduced with one all-reduce
(lcrnet_amd.evaluation.registration_partial / registration_reduce, the reference's utils/utils/torch.py:16-34).

Pairs: the 15 combinations of the 6 committed KITTI demo scans (tests/golden/scans), cycled.  For every P the same pairs go
through PairPipeline(pairs_per_call=P): P = 1 is the reference's loop (one pair per forward, model_family/LCRNet.py:274-321; four
pairs in flight on four pinned host threads), P > 1 stacks P pairs per `LCRNet.forward_pairs` call.  The attention kernel's rate is
measured live with HIP events inside the library (KernelTimer): algorithmic flops 4 * Nq * Nk * 128 per attention problem (QK^T and
PV over 4 heads x 32) / launch time, against the 157.3 TFLOP/s fp32 MFMA peak."""
import argparse
import itertools
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FP32_PEAK_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs-per-call", type=int, nargs="+", default=[1, 8, 32])
    ap.add_argument("--pairs", type=int, default=192, help="pairs per timed pass (64 made a pass of 8-pair calls 0.15 s long: +-10 %% pass to pass)")
    ap.add_argument("--workers", type=int, default=0, help="host threads / streams with a call in flight; 0 = 4 at one pair per call, 5 when pairs are batched")
    ap.add_argument("--repeats", type=int, default=5, help="timed passes over the pairs; the median pass is reported (min / max beside it)")
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:          # launch the ranks ourselves (torchrun's environment contract)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                  env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                                           MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(args.gpus)]
        sys.exit(max(p.wait() for p in procs))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = 0 if os.environ.get("LCR_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", 0))
    from lcrnet_amd import functional as F
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.pipeline import PairPipeline
    from lcrnet_amd.weights import seeded_state_dict
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("LCR_BENCH_BACKEND", "gloo" if os.environ.get("LCR_BENCH_SINGLE_DEVICE") else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if (dist is not None and dist.get_backend() == "nccl") else None
    limits = [74, 68, 70, 67]
    cfg = make_cfg()
    cfg["neighbor_limits"] = limits
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
    m = m.to(dev)
    gold = os.path.join(ROOT, "tests", "golden", "scans")
    names = sorted(f[:-4] for f in os.listdir(gold) if f.endswith(".npy"))
    scans = {n: torch.from_numpy(np.load(os.path.join(gold, n + ".npy"))).to(dev) for n in names}
    combos = list(itertools.combinations(names, 2))
    work = []
    for i in range(args.pairs):
        a, b = combos[i % len(combos)]
        work.append((torch.cat([scans[a], scans[b]]), torch.tensor([len(scans[a]), len(scans[b])], dtype=torch.int64, device=dev)))
    n_total = len(work)
    work = work[rank::world]                                      # this rank's pairs
    results = {}
    for P in args.pairs_per_call:
        workers = args.workers or (4 if P == 1 else 5)
        with PairPipeline(m, neighbor_limits=limits, workers=workers, pairs_per_call=P) as pp:
            for _ in pp.run(work * max(2, workers)):               # warm-up: as many FULL untimed passes as workers (at least two) — the timed passes then see exactly the stack
                pass                                              # shapes the caching allocator already holds blocks for (a warm-up over a prefix, or one
                                                                  # pass only — two workers interleave differently the second time —
                                                                  # left the first timed pass growing the pool: round 3's 305 / 389 pairs/s minima)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            timer = F.KernelTimer({"attention"})
            F.set_timer(timer)
            dts, in_order, dev_allocs = [], [], []
            for _ in range(max(args.repeats, 1)):
                a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
                t0 = time.perf_counter()
                n_corr = 0
                for out in pp.run(work):
                    n_corr += out["corr_scores"].shape[0]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if dist is not None:                              # the slowest rank's time counts
                    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev or "cpu")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt[0])
                dts.append(dt)
                in_order.append(round(n_total / dt, 1))
                dev_allocs.append(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - a0)      # hipMalloc calls of the caching allocator during the pass
            F.set_timer(None)
            dts.sort()
            dt = dts[len(dts) // 2]
        att = timer.summary()["attention"]
        t_att = sum(t for t, _ in att)
        flops = sum(4.0 * meta[0] * meta[2] * meta[3] for _, meta in att)
        results[str(P)] = {"pairs_per_s": round(n_total / dt, 2), "pairs_per_s_min": round(n_total / dts[-1], 2), "pairs_per_s_max": round(n_total / dts[0], 2),
                           "timed_passes": len(dts), "passes_pairs_per_s": in_order, "device_allocs_per_pass": dev_allocs, "ms_per_pair": round(dt / n_total * 1e3, 3),
                           "attention_launches_per_pair": round(len(att) / (len(work) * len(dts)), 2),
                           "attention_us_per_launch": round(t_att / max(len(att), 1) * 1e6, 2),
                           "attention_tflops": round(flops / max(t_att, 1e-12) / 1e12, 3),
                           "attention_frac_of_fp32_mfma_peak": round(flops / max(t_att, 1e-12) / 1e12 / FP32_PEAK_TFLOPS, 4),
                           "mean_correspondences": round(n_corr / len(work), 1)}
    # the reference's per-pair output file (demo/demo.py:84-105) and its registration metrics on a few results (formats only: the
    # demo pairs carry no ground-truth pose here, identity stands in, and random weights give random poses)
    import shutil
    import tempfile
    from lcrnet_amd import evaluation as ev
    from lcrnet_amd import io_formats as io
    tmp = tempfile.mkdtemp(prefix="lcr_reg_")
    with PairPipeline(m, neighbor_limits=limits, workers=1, pairs_per_call=4) as pp:
        outs = list(pp.run(work[:4]))
    paths = [io.save_registration(tmp, 0, i, i + 1, o, np.eye(4, dtype=np.float32)) for i, o in enumerate(outs)]
    back = [io.load_registration(p) for p in paths]
    summary = ev.registration_reduce(ev.registration_partial([b["transform"] for b in back], [b["estimated_transform"] for b in back]),
                                     device=red_dev)             # N > 1: every rank's four pairs, one all-reduce of the sums
    summary = {k: (None if isinstance(v, float) and v != v else v) for k, v in summary.items()}       # NaN means (no accepted pair) -> null
    files = {"written": len(paths), "keys": len(back[0]), "bytes": sum(os.path.getsize(p) for p in paths), "registration_summary_vs_identity": summary}
    shutil.rmtree(tmp, ignore_errors=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    best = max(results, key=lambda k: results[k]["pairs_per_s"])
    print(json.dumps({"metric": "registration pairs/s (pair model end to end, %d GPU%s)" % (world, "s" if world > 1 else ""),
                      "value": results[best]["pairs_per_s"], "unit": "pairs/s", "n_gpus": world,
                      "pairs_per_call_best": int(best), "workers": args.workers or "4 at one pair per call, 5 when batched", "pairs": n_total,
                      "config": "15 combinations of the 6 KITTI demo scans (~17k pts each after 0.3 m voxels), limits [74,68,70,67], seeded random weights",
                      "by_pairs_per_call": results, "registration_files": files}))


if __name__ == "__main__":
    main()
