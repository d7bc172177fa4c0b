#!/usr/bin/env python
"""bench.py — scans/s of the per-scan hot path on MI355X (BASELINE.json metric, configs[1]).

One step = one batch of 8 synthetic 64-beam scans (~120 k points each, already resident in HBM) through the whole path:
0.3 m voxelisation (a-1) -> 3 grid subsamples + 10 radius searches (a-1/a-2/a-3) -> KPConv encoder (a-4..a-6) ->
NetVLAD (a-7) -> 8 unit-norm 256-D descriptors.  Weights: seeded random in the reference checkpoint layout (no checkpoint
in the containers).  N GPUs = N ranks, each with its own batch (scan-parallel, weak scaling), descriptors all-gathered
over RCCL inside the timed region (the exchange step of the retrieval).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = int(os.environ.get("LCR_BENCH_BATCH", "8"))   # BASELINE configs[1]: 8 scans per step (override only for experiments)
VOXEL, RADIUS, NUM_STAGES = 0.3, 1.275, 4
LIMITS = [64, 65, 74, 80]          # reference training/eval default (dataset_loop_detection.py:25,80)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3           # dense fp32 MFMA/vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-upsampling", action="store_true", help="skip the 3 decoder-only upsampling searches")
    ap.add_argument("--no-overlap", action="store_true", help="run pre-processing and encoder on one stream (no pipelining)")
    ap.add_argument("--no-thread", action="store_true", help="two streams but a single host thread")
    ap.add_argument("--pre-workers", type=int, default=2, help="host threads / streams pre-processing consecutive batches concurrently")
    ap.add_argument("--depth", type=int, default=2, help="batches pre-processed ahead of the encoder")
    ap.add_argument("--single-encoder", action="store_true", help="one encoder stream (default: consecutive batches alternate between two)")
    return ap.parse_args()


def make_batch(rank):
    import lcrnet_amd.synthetic as synthetic
    scans = [synthetic.synthetic_scan(rank * BATCH + i) for i in range(BATCH)]
    return scans



def cpu_baseline(scans, n_scans=2):
    """The same path on host cores: native ops from the compiled reference when oracle/_ref is present (else the C++
    restatement), encoder + NetVLAD from the torch fp32 restatement.  Bounded sample, rank 0 only."""
    from oracle import ops as oracle_ops
    from oracle import torch_ref
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.weights import seeded_state_dict
    impl = "ref" if oracle_ops.have_ref() else "oracle"
    m = create_model()
    sd = seeded_state_dict(m.state_dict(), 7351)
    threads = torch.get_num_threads()
    t_pre = t_enc = 0.0
    t0 = time.time()
    with torch.no_grad():
        for raw in scans[:n_scans]:
            t = time.time()
            p, l = oracle_ops.grid_subsample(raw, np.array([len(raw)]), VOXEL, impl=impl)
            st = oracle_ops.precompute_data_stack_mode(p, l, NUM_STAGES, VOXEL, RADIUS, LIMITS, impl=impl)
            t_pre += time.time() - t
            t = time.time()
            dd = {k: [torch.from_numpy(np.ascontiguousarray(x)) for x in v] for k, v in st.items()}
            feats = torch_ref.kp_encoder(sd, torch.ones(len(p), 1), dd)
            torch_ref.global_descriptor(sd, feats[-1])
            t_enc += time.time() - t
    dt = time.time() - t0
    return {"value": round(n_scans / dt, 4), "unit": "scans/s", "cores": threads,
            "kind": "reference" if impl == "ref" else "port",
            "sample": "%d of the %d scans of this batch, end to end; native ops: %s (1 thread, %.2f s/scan); encoder+NetVLAD: torch fp32 "
                      "restatement on %d threads (%.2f s/scan)" % (n_scans, BATCH, "reference C++ compiled from source (oracle/_ref)"
                                                                    if impl == "ref" else "oracle C++ restatement", t_pre / n_scans,
                                                                    threads, t_enc / n_scans)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LCR_BENCH_BACKEND", "nccl")         # "nccl" is RCCL on ROCm; "gloo" only for dry runs
        if os.environ.get("LCR_BENCH_SINGLE_DEVICE"):                  # dry run of the N>1 code path on a 1-GPU box
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from lcrnet_amd import functional as F
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.weights import seeded_state_dict
    model = create_model().eval()
    model.load_state_dict(seeded_state_dict(model.state_dict(), 7351))
    model = model.to(dev)

    scans = make_batch(rank)
    raw_pts = torch.from_numpy(np.concatenate(scans)).to(dev)
    raw_lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
    gathered = torch.empty((world * BATCH, 256), dtype=torch.float32, device=dev) if world > 1 else None

    from lcrnet_amd.pipeline import DescriptorPipeline
    pipe = DescriptorPipeline(model, VOXEL, RADIUS, NUM_STAGES, LIMITS, upsampling=not args.no_upsampling, raw_voxel=VOXEL,
                              overlap=not args.no_overlap, producer_thread=not args.no_thread, pre_workers=args.pre_workers, depth=args.depth)
    if not (args.single_encoder or args.no_overlap or args.no_thread):
        pipe.enable_dual_encoder(int(os.environ.get("LCR_ENC_STREAMS", "2")))

    def run_steps(n):
        """n steps = n batches, each fully processed (voxelise .. descriptors [+ all-gather]); the pre-processing of step
        k+1 overlaps the encoder of step k on a second stream (DescriptorPipeline)."""
        last = None
        dual = pipe.enc_streams is not None and not (args.no_overlap or args.no_thread)
        for item in pipe.run(((raw_pts, raw_lens) for _ in range(n)), sync_to_caller=not dual):
            desc, es = (item[0], item[2]) if dual else (item, None)
            if world > 1:
                if es is not None:
                    with torch.cuda.stream(es):              # the collective follows the encoder on ITS stream: nothing is parked
                        dist.all_gather_into_tensor(gathered, desc.contiguous())   # on the caller's queue
                else:
                    dist.all_gather_into_tensor(gathered, desc.contiguous())
            last = desc
        return last

    run_steps(8)                 # priming, untimed and not counted: allocator growth for the batches in flight, lazy code-object loads
    run_steps(args.warmup)
    stage_points = [sum(l) for l in pipe.preprocess(raw_pts, raw_lens)["lengths_host"]]
    # ---- timed region: exactly K steps between barrier + synchronize
    timer = F.KernelTimer({"kpconv_aggregate", "gemm"})
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    F.set_timer(timer)
    t0 = time.perf_counter()
    desc = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    F.set_timer(None)
    if os.environ.get("LCR_PIPE_STATS") and rank == 0:
        print("pipeline host threads:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in pipe.stats.items()}, file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    iso = None
    summ = timer.summary() if rank == 0 else None     # read the timed region's log before anything else is logged
    if rank == 0:
        # the same GEMM launches once more with nothing else on the GPU (one stream, outside the timed region): in the timed
        # region three streams share the CUs, so a launch's event-to-event duration includes time it spent waiting for them
        iso_timer = F.KernelTimer({"gemm"})
        dd_iso = pipe.preprocess(raw_pts, raw_lens)
        pipe.encode(dd_iso)
        torch.cuda.synchronize()
        F.set_timer(iso_timer)
        for _ in range(3):
            pipe.encode(dd_iso)
        torch.cuda.synchronize()
        F.set_timer(None)
        g = iso_timer.summary()["gemm"]
        iso = sum(2.0 * m[0] * m[1] * m[2] for _, m in g) / sum(t for t, _ in g) / 1e12
    if rank == 0:
        assert torch.isfinite(desc).all() and abs(float(desc.norm(dim=1).mean()) - 1.0) < 1e-3
        # ---- roofline of the dominant kernel family, measured live with HIP events on the launch stream.
        # Dominant by time = lcr::k_gemm_f32 (all tile variants; ~30 % of the step, profiles/): compute-bound on the fp32
        # matrix cores, so "achieved" = algorithmic flops (2*M*N*K per launch) / launch time vs the 157.3 TFLOP/s
        # dense fp32 MFMA peak.  HBM traffic per launch comes from the committed rocprofv3 PMC passes (profiles/*pmc*.json).
        gem, agg = summ["gemm"], summ["kpconv_aggregate"]
        t_gemm, t_agg = sum(t for t, _ in gem), sum(t for t, _ in agg)
        flops = sum(2.0 * m[0] * m[1] * m[2] for _, m in gem)

        def agg_bytes(m):   # indices + query/support xyz + support features + pos flags + the (M,15C) output + counts
            M, Ns, H, C, isz = m
            return M * H * isz + (M + Ns) * 12 + Ns * C * 4 + Ns + M * 15 * C * 4 + M * 4
        bytes_agg = sum(agg_bytes(m) for _, m in agg)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("k_gemm_f32", {}).get("traffic_bytes")
        roof = {"bound": "mfma", "kernel": "lcr::k_gemm_f32 (fp32 MFMA, %d launches/step)" % (len(gem) // max(args.steps, 1)),
                "achieved": round(flops / t_gemm / 1e12, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / t_gemm / 1e12 / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                "achieved_alone": round(iso, 2), "frac_alone": round(iso / FP32_PEAK_TFLOPS, 4),
                "avg_launch_us": round(t_gemm / max(len(gem), 1) * 1e6, 2),
                "gflop_per_launch": round(flops / max(len(gem), 1) / 1e9, 3),
                "share_of_step": {"gemm": round(t_gemm / dt, 3), "kpconv_aggregate": round(t_agg / dt, 3)},
                "secondary": {"kernel": "lcr::k_kpconv_aggregate", "bound": "hbm", "achieved": round(bytes_agg / t_agg / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_agg / t_agg / 1e9 / HBM_PEAK_GBS, 4),
                              "avg_launch_us": round(t_agg / max(len(agg), 1) * 1e6, 2)}}
        line = {
            "metric": "scans/s (120k-pt KITTI-shape scan -> 256-D descriptor)",
            "value": round(world * BATCH * args.steps / dt, 3), "unit": "scans/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: batch of 8 synthetic 64-beam scans (~120k pts, 0.3 m voxel -> ~16k pts), "
                                   "voxelise + 3 subsamples + %d radius searches + KPConv encoder + NetVLAD, seeded random weights"
                                   % (7 if args.no_upsampling else 10),
                       "scans_per_step_per_gpu": BATCH, "raw_points_per_scan": int(raw_pts.shape[0] // BATCH),
                       "stage_points_per_batch": stage_points, "neighbor_limits": LIMITS,
                       "streams": "1" if args.no_overlap else ("pre-processing stream (own host thread, 2 batches ahead) + %d encoder stream(s)"
                                                               % (1 if (args.single_encoder or args.no_thread) else 2)),
                       "parallelism": "scan-parallel x%d, all-gather of descriptors" % world},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(scans)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
