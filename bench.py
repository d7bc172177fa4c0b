#!/usr/bin/env python
"""bench.py — scans/s of the per-scan hot path on MI355X (BASELINE.json metric, configs[1]).

One step = one batch of 8 synthetic 64-beam scans (~120 k points each, already resident in HBM) through the whole path:
0.3 m voxelisation (a-1) -> 3 grid subsamples + 10 radius searches (a-1/a-2/a-3) -> KPConv encoder (a-4..a-6) ->
NetVLAD (a-7) -> 8 unit-norm 256-D descriptors.  Weights: seeded random in the reference checkpoint layout (no checkpoint
in the containers).  N GPUs = N ranks (one process per GPU), each with its own batch (scan-parallel, weak scaling),
descriptors all-gathered over RCCL inside the timed region (the exchange step of the retrieval; the reference's
multi-process entry is utils/engine/base_tester.py:88).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]

Launch forms: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks come from the
environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  A bare `python bench.py --gpus N` with N > 1 starts the N rank
processes itself (same environment contract, 127.0.0.1 rendezvous) and rank 0 prints the line with `n_gpus: N`.
`LCR_BENCH_SINGLE_DEVICE=1` puts every rank on GPU 0 (dry run of the N-rank code path on a 1-GPU box; backend from
`LCR_BENCH_BACKEND`, default gloo there because RCCL refuses two ranks on one device).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
_PRE_WORKER = "--cpu-worker" in sys.argv and sys.argv[sys.argv.index("--cpu-worker") + 1] == "pre"
if not _PRE_WORKER:                                        # a pre-processing-only CPU worker is numpy + ctypes only
    import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = int(os.environ.get("LCR_BENCH_BATCH", "8"))   # BASELINE configs[1]: 8 scans per step (override only for experiments)
VOXEL, RADIUS, NUM_STAGES = 0.3, 1.275, 4
LIMITS = [64, 65, 74, 80]          # reference training/eval default (dataset_loop_detection.py:25,80)
ISO_PASSES = 3                     # clocked passes per distinct batch for the kernel-alone figures
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3           # dense fp32 MFMA/vector peak
BF16_PEAK_TFLOPS = 2500.0          # dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
RS_VALU_PER_QUERY = 354            # SQ_INSTS_VALU per query of the search kernel (rocprofv3 --pmc)
RS_VALU_SOURCE = "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU per query, profiles/r02_radius_pmc.md; 256 CUs at 2.4 GHz"
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # written by tools/pmc_summary.py from separate rocprofv3 --pmc passes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="the timed K-step block runs this many times back to back (each bracketed by "
                    "barrier + synchronize); the line reports the MEDIAN block, ms_per_step_min / _max the spread")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-ab", action="store_true", help="skip the secondary A/B block with the split-bf16 K-deep GEMMs")
    ap.add_argument("--no-lazy-ab", action="store_true", help="skip the secondary block without the 3 decoder-only upsampling searches")
    ap.add_argument("--no-h2d", action="store_true", help="skip the secondary `with_h2d` measurement (host [N,4] batches uploaded inside the pipeline)")
    ap.add_argument("--no-blocks", action="store_true", help="skip the secondary blocks for configs[2]-[4] (`pairs`, `retrieval`, `sequence`: tools/bench_blocks.py)")
    ap.add_argument("--no-upsampling", action="store_true", help="skip the 3 decoder-only upsampling searches")
    ap.add_argument("--no-overlap", action="store_true", help="run pre-processing and encoder on one stream (no pipelining)")
    ap.add_argument("--no-thread", action="store_true", help="two streams but a single host thread")
    ap.add_argument("--pre-workers", type=int, default=2, help="host threads / streams pre-processing consecutive batches concurrently")
    ap.add_argument("--depth", type=int, default=2, help="batches pre-processed ahead of the encoder")
    ap.add_argument("--single-encoder", action="store_true", help="one encoder stream (default: consecutive batches alternate between two)")
    ap.add_argument("--distinct-batches", type=int, default=4, help="distinct input batches the steps cycle through (the batch of 8 scans "
                    "rotated about the vertical axis by k * 37 degrees: other voxels, other neighbourhoods, same scene statistics)")
    ap.add_argument("--cpu-worker", nargs=4, metavar=("MODE", "SCAN", "THREADS", "START"), help=argparse.SUPPRESS)
    return ap.parse_args()


def make_batch(rank):
    import lcrnet_amd.synthetic as synthetic
    return [synthetic.synthetic_scan(rank * BATCH + i) for i in range(BATCH)]


sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_host import bind_rank, cpu_baseline, cpu_worker, spawn_ranks  # noqa: E402  (CPU baseline, rank launch / core binding)


def rotated_inputs(scans, nb_in, dev):
    """nb_in resident input batches: the scans rotated about the vertical axis by k * 37 degrees (other voxels, other neighbourhoods, same
    scene statistics) -> [(points f32 [sum N, 3], lengths i64 [B])] on `dev`."""
    inputs = []
    for k in range(nb_in):
        a = np.deg2rad(float(os.environ.get("LCR_BENCH_ROT_DEG", "37")) * k)
        R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
        inputs.append((torch.from_numpy(np.concatenate([s @ R.T for s in scans]).astype(np.float32)).to(dev),
                       torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)))
    return inputs


def search_bytes(stage_points, upsampling):
    """SURVEY §8d a-2, int32 indices: (Nq + Ns) * 12 B + Nq * limit * 4 B for each of the 10 (7) searches of one batch."""
    n, tot = stage_points, 0
    for i in range(NUM_STAGES):
        tot += 2 * n[i] * 12 + n[i] * LIMITS[i] * 4
        if i + 1 < NUM_STAGES:
            tot += (n[i + 1] + n[i]) * 12 + n[i + 1] * LIMITS[i] * 4
            if upsampling:
                tot += (n[i] + n[i + 1]) * 12 + n[i] * LIMITS[i + 1] * 4
    return tot


def main():
    args = parse()
    if args.cpu_worker:
        return cpu_worker(*args.cpu_worker)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    backend = None
    # LCR_BENCH_FORCE_DIST=1: a ONE-rank process group, so that the N-rank code path (RCCL communicator, barriers, the all-gather issued on
    # the encoder's external stream inside the timed region, the max-over-ranks reduction) runs against real RCCL on a single-GPU box
    force_dist = world == 1 and bool(os.environ.get("LCR_BENCH_FORCE_DIST"))
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29555")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        single = bool(os.environ.get("LCR_BENCH_SINGLE_DEVICE"))          # dry run of the N>1 code path on a 1-GPU box
        backend = os.environ.get("LCR_BENCH_BACKEND", "gloo" if single else "nccl")   # "nccl" is RCCL on ROCm
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        bound = bind_rank(local, n_local, single)                       # both launch forms (torchrun / own spawn) pass through here
        if single:
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
        bound = (len(os.sched_getaffinity(0)), "single rank: not bound")
    dev = torch.device("cuda", torch.cuda.current_device())

    from lcrnet_amd import functional as F
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.weights import seeded_state_dict
    model = create_model().eval()
    model.load_state_dict(seeded_state_dict(model.state_dict(), 7351))
    model = model.to(dev)

    scans = make_batch(rank)
    # the steps cycle through a few DISTINCT resident batches (not one buffer re-fed every step): the scans rotated about z
    nb_in = max(1, args.distinct_batches)
    inputs = rotated_inputs(scans, nb_in, dev)
    raw_pts, raw_lens = inputs[0]
    gathered = torch.empty((world * BATCH, 256), dtype=torch.float32, device=dev) if dist is not None else None

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
        for item in pipe.run((inputs[k % nb_in] for k in range(n)), sync_to_caller=not dual):
            desc, es = (item[0], item[2]) if dual else (item, None)
            if dist is not None:
                if es is not None:
                    with torch.cuda.stream(es):              # the collective follows the encoder on ITS stream: nothing is parked
                        dist.all_gather_into_tensor(gathered, desc.contiguous())   # on the caller's queue
                else:
                    dist.all_gather_into_tensor(gathered, desc.contiguous())
            last = desc
        return last

    run_steps(8)                 # priming, untimed and not counted: allocator growth for the batches in flight, lazy code-object loads
    run_steps(args.warmup)
    # per distinct batch: stage sizes and the valid (non-padding) entries of the index lists the KPConv layers gather (the
    # aggregation's flops are 2 * 15 * nnz * C)
    stage_points_all, nnz = [], {}
    for pts_k, lens_k in inputs:
        dd0 = pipe.preprocess(pts_k, lens_k)
        sp = [sum(l) for l in dd0["lengths_host"]]
        stage_points_all.append(sp)
        for i in range(NUM_STAGES):
            nnz[(sp[i], sp[i])] = int((dd0["neighbors"][i] < sp[i]).sum())
            if i + 1 < NUM_STAGES:
                nnz[(sp[i + 1], sp[i])] = int((dd0["subsampling"][i] < sp[i]).sum())
        del dd0
    stage_points = stage_points_all[0]
    # ---- timed region: exactly K steps between barrier + synchronize — R such blocks back to back, the line reports the MEDIAN one
    blocks = []
    R = max(1, args.repeats)
    # per-launch clocks on every SAMPLE-th launch of a kind only: a timed launch is a profiled dispatch (4 event records, timestamped
    # completion signal, system-scope release at kernel end); with every launch timed the pipeline ran 7 % slower (2 409 vs 2 577
    # scans/s).  9 is coprime with the 35 GEMMs / 10 aggregations of a step, so every shape is sampled equally often.
    SAMPLE = max(1, int(os.environ.get("LCR_BENCH_KTIMER_SAMPLE", "9")))
    for rep in range(R):
        timer = F.KernelTimer({"kpconv_aggregate", "gemm", "radius_query"})
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if not os.environ.get("LCR_BENCH_NO_KTIMER"):                  # A/B: what the per-launch events cost the timed region
            F.set_timer(timer, SAMPLE)
        t0 = time.perf_counter()
        desc = run_steps(args.steps)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt_r = time.perf_counter() - t0
        F.set_timer(None)
        if dist is not None:
            t = torch.tensor([dt_r], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        blocks.append((dt_r, timer))
    order_r = sorted(range(R), key=lambda i: blocks[i][0])
    dt, timer = blocks[order_r[(R - 1) // 2]]                            # median block (lower median for even R)
    dt_min, dt_max = blocks[order_r[0]][0], blocks[order_r[-1]][0]
    if os.environ.get("LCR_PIPE_STATS") and rank == 0:
        print("pipeline host threads:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in pipe.stats.items()}, file=sys.stderr)

    summ = timer.records() if rank == 0 else None     # read the timed region's log before anything else is logged
    # ---- the same steps fed from HOST memory (not the headline): every batch arrives as a pinned host tensor f32 [sum N, 4] (KITTI
    # velodyne rows x, y, z, intensity, as a loader thread reading .bin files would leave them) + i64 lengths, is uploaded on the copy
    # stream `depth` batches ahead (lcrnet_amd.pipeline.HostIngest) and consumed unsliced by the voxel-key kernels.  The reference:
    # `_load_point_cloud(...)[:, :3]` + to_cuda (dataset_overlap_online.py:245-253, utils/engine/single_tester.py:59).
    with_h2d = None
    secondary = world == 1            # the A/B blocks belong to the single-GPU line; an N-rank run measures the headline only
    if secondary and not args.no_h2d and not (args.no_overlap or args.no_thread):
        host_inputs = []
        g = torch.Generator().manual_seed(5)
        for pts_k, lens_k in inputs:
            x4 = torch.cat([pts_k.cpu(), torch.rand(pts_k.shape[0], 1, generator=g)], dim=1).contiguous().pin_memory()
            host_inputs.append((x4, lens_k.cpu().pin_memory()))
        saved = inputs
        inputs = host_inputs
        run_steps(8)
        dts = []
        for rep in range(min(R, 3)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            desc_h = run_steps(args.steps)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            d = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([d], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d = float(t.item())
            dts.append(d)
        # the same from PAGEABLE host tensors (what np.fromfile / np.load hand a loader): staged through the ingest ring's pinned slots first
        inputs = [(x.clone(), l.clone()) for x, l in host_inputs]
        run_steps(8)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        d_page = time.perf_counter() - t0
        inputs = saved
        dth = sorted(dts)[(len(dts) - 1) // 2]
        same = float((desc_h - desc).abs().max())            # last step of both regions fed the same batch ((steps - 1) % nb_in)
        mb = sum(x.numel() * 4 for x, _ in host_inputs) / len(host_inputs) / 1e6
        with_h2d = {"value": round(world * BATCH * args.steps / dth, 3), "unit": "scans/s", "ms_per_step": round(dth / args.steps * 1e3, 3),
                    "over_resident": round(dt / dth, 4), "pageable_host_value": round(world * BATCH * args.steps / d_page, 3), "h2d_mb_per_step": round(mb, 2), "h2d_gb_per_s_needed": round(mb / 1e3 / (dth / args.steps), 2),
                    "descriptors_max_abs_diff_vs_resident": same,
                    "what": "same steps, inputs as pinned host f32 [N,4] rows (x, y, z, intensity) uploaded on a copy stream inside the pipeline; "
                            "median of %d blocks of %d steps; NOT the headline (value = inputs resident in HBM)" % (len(dts), args.steps)}
    # ---- A/B, not the headline: the same steps with the OTHER form of the K-deep GEMMs (N >= 64, K >= 288).  Default (round 5): fp32 operands as
    # three bf16 terms, six cross products on the bf16 matrix cores, fp32 accumulation (lcr_gemm_f32_bsplit; `dtype` spells it out); the block
    # `true_fp32_gemm_ab` then times the fp32-MFMA kernel (v_mfma_f32_32x32x2_f32) everywhere.  With LCR_GEMM_SPLIT=0 the roles swap.
    split_ab, split_ab_key = None, None
    if secondary and not args.no_split_ab:
        headline_split = F.gemm_split_enabled()
        F.set_gemm_split(not headline_split)
        run_steps(8)
        dts = []
        for rep in range(min(R, 3)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            desc_s = run_steps(args.steps)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            d = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([d], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d = float(t.item())
            dts.append(d)
        F.set_gemm_split(headline_split)
        run_steps(2)                                                     # back on the headline's tables before anything else is measured
        dts_m = sorted(dts)[(len(dts) - 1) // 2]
        split_ab_key = "true_fp32_gemm_ab" if headline_split else "split_bf16_gemm_ab"
        split_ab = {"value": round(world * BATCH * args.steps / dts_m, 3), "unit": "scans/s", "ms_per_step": round(dts_m / args.steps * 1e3, 3),
                    "over_headline": round(dt / dts_m, 4), "descriptors_max_abs_diff_vs_headline": float((desc_s - desc).abs().max()),
                    "what": ("same steps with LCR_GEMM_SPLIT=0: every GEMM on v_mfma_f32_32x32x2_f32 (true fp32 MFMA); " if headline_split else
                             "same steps with LCR_GEMM_SPLIT=1: K-deep GEMMs as 3 x bf16 terms / 6 products on the bf16 matrix cores (fp32-faithful); ")
                            + "NOT the headline; median of %d blocks" % len(dts)}
    # ---- secondary, not the headline: the descriptor-only deployment.  The three upsampling lists are consumed by the KPDecoder of the pair model
    # only (backbone4.py:347-367); loop detection (BASELINE configs 2-4) never reads them ("compute them lazily"), and
    # DescriptorPipeline's default is not to build them.  The headline keeps all ten searches of the reference's collate (§8d); this block runs
    # the same steps with the seven the descriptor path consumes.
    lazy = None
    if secondary and not args.no_upsampling and not args.no_lazy_ab and not (args.no_overlap or args.no_thread):
        pipe_main, pipe = pipe, None
        pipe7 = DescriptorPipeline(model, VOXEL, RADIUS, NUM_STAGES, LIMITS, upsampling=False, raw_voxel=VOXEL, overlap=True, producer_thread=True,
                                   pre_workers=args.pre_workers, depth=args.depth)
        pipe7.enable_dual_encoder(int(os.environ.get("LCR_ENC_STREAMS", "2")))
        pipe = pipe7
        run_steps(8)
        dts = []
        for rep in range(min(R, 3)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            desc_7 = run_steps(args.steps)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            d = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([d], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d = float(t.item())
            dts.append(d)
        pipe = pipe_main
        pipe7.close()
        d7 = sorted(dts)[(len(dts) - 1) // 2]
        lazy = {"value": round(world * BATCH * args.steps / d7, 3), "unit": "scans/s", "ms_per_step": round(d7 / args.steps * 1e3, 3),
                "over_headline": round(dt / d7, 4), "descriptors_max_abs_diff_vs_headline": float((desc_7 - desc).abs().max()),
                "what": "same steps with the 7 searches the descriptor path consumes (the 3 decoder-only upsampling lists not built: DescriptorPipeline's "
                        "default for loop detection); NOT the headline, which keeps the reference collate's 10; median of %d blocks" % len(dts)}
    # ---- secondary, not the headline: BASELINE configs[2]-[4] on this one GPU under the driver's clock — registration
    # pairs/s at 1 and 16 pairs per call, the retrieval at KITTI-00 / KITTI-00-10 corpus sizes, and a 2 048-frame sequence scans -> rows
    blocks_cfg = None
    if secondary and rank == 0 and not args.no_blocks and not (args.no_overlap or args.no_thread):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_blocks
        torch.cuda.synchronize()
        def guarded(name, fn):                         # a secondary block must never cost the headline its line: failures are reported in place
            try:
                return fn()
            except Exception as e:                     # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                torch.cuda.synchronize()
                return {"error": "%s block failed: %r" % (name, e)}
        blocks_cfg = {"retrieval": guarded("retrieval", lambda: bench_blocks.retrieval_block(dev)),
                      "sequence": guarded("sequence", lambda: bench_blocks.sequence_block(model, dev, scans, frames=int(os.environ.get("LCR_BENCH_SEQ_FRAMES", "2048")))),
                      "pairs": guarded("pairs", lambda: bench_blocks.pairs_block(dev, repeats=int(os.environ.get("LCR_BENCH_PAIR_PASSES", "5"))))}
        torch.cuda.empty_cache()
    final_line = None
    iso = None
    if rank == 0 and not os.environ.get("LCR_BENCH_NO_KTIMER"):
        # Every distinct batch once more with NOTHING else on the GPU (one stream, one batch in flight, outside the timed region) and
        # EVERY launch clocked: the kernels' own durations as a kernel trace of the encoder alone reports them
        # (profiles/rNN_encoder_alone_kernel_summary.md), and the exact flops / launches of a step (no sampling).
        iso = {"t_gemm": 0.0, "t_agg": 0.0, "t_rs": 0.0, "flops": [], "flops_agg": [], "n_gemm": 0, "n_agg": 0, "n_rs": 0, "passes": 0,
               "t_split": 0.0, "n_split": 0, "fl_split": 0.0, "t_light": 0.0, "n_light": 0, "fl_light": 0.0}
        is_split = lambda m: F.gemm_split_enabled() and F.gemm_split_ok(m[1], m[2])     # the shapes the encoder driver sends to lcr_gemm_f32_bsplit
        for pts_k, lens_k in inputs:
            pipe.encode(pipe.preprocess(pts_k, lens_k))                # warm: this batch's allocations and cache lines
            torch.cuda.synchronize()
            for _ in range(ISO_PASSES):
                iso_timer = F.KernelTimer({"kpconv_aggregate", "gemm", "radius_query"})
                F.set_timer(iso_timer)
                pipe.encode(pipe.preprocess(pts_k, lens_k))
                torch.cuda.synchronize()
                F.set_timer(None)
                r = iso_timer.records()
                own = lambda recs: sum((k if k is not None else b) for b, k, _ in recs)
                iso["t_gemm"] += own(r["gemm"])
                iso["t_agg"] += own(r["kpconv_aggregate"])
                iso["t_rs"] += own(r["radius_query"])
                iso["n_gemm"] += len(r["gemm"])
                iso["n_agg"] += len(r["kpconv_aggregate"])
                iso["n_rs"] += len(r["radius_query"])
                iso["passes"] += 1
                for b_, k_, m_ in r["gemm"]:
                    cls = "split" if is_split(m_) else "light"
                    iso["t_" + cls] += k_ if k_ is not None else b_
                    iso["n_" + cls] += 1
                    iso["fl_" + cls] += 2.0 * m_[0] * m_[1] * m_[2]
                    # the launch's own roofline time: the slower of its fp32 MFMA time and its algorithmic HBM time (A + B + C once each)
                    iso["t_roof_" + cls] = iso.get("t_roof_" + cls, 0.0) + max(2.0 * m_[0] * m_[1] * m_[2] / (FP32_PEAK_TFLOPS * 1e12),
                                                                            4.0 * (m_[0] * m_[2] + m_[1] * m_[2] + m_[0] * m_[1]) / (HBM_PEAK_GBS * 1e9))
                    iso["hbm_bound_" + cls] = iso.get("hbm_bound_" + cls, 0) + (4.0 * (m_[0] * m_[2] + m_[1] * m_[2] + m_[0] * m_[1]) / (HBM_PEAK_GBS * 1e9) >
                                                                               2.0 * m_[0] * m_[1] * m_[2] / (FP32_PEAK_TFLOPS * 1e12))
            iso["flops"].append(sum(2.0 * m[0] * m[1] * m[2] for _, _, m in r["gemm"]))
            iso["flops_agg"].append(sum(2.0 * 15 * nnz.get((M, Ns), M * H) * C for _, _, (M, Ns, H, C, isz) in r["kpconv_aggregate"]))
    if rank == 0 and os.environ.get("LCR_BENCH_NO_KTIMER"):            # A/B run without per-launch events: the headline only
        print(json.dumps({"value": round(world * BATCH * args.steps / dt, 3), "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "ms_per_step_min": round(dt_min / args.steps * 1e3, 3), "ms_per_step_max": round(dt_max / args.steps * 1e3, 3), "ktimer": False}), flush=True)
        rank = -1
    if rank == 0:
        assert torch.isfinite(desc).all() and abs(float(desc.norm(dim=1).mean()) - 1.0) < 1e-3
        # ---- roofline of the dominant kernel family (by time: the fp32 GEMMs, lcr::k_gemm_f32_deep + k_gemm_f32, profiles/).
        # Compute-bound on the fp32 matrix cores: achieved = algorithmic flops (2*M*N*K per launch) / the kernels' own
        # begin-to-end durations (hipExtLaunchKernel start / stop events on the launch stream = what a kernel trace reports) vs the
        # 157.3 TFLOP/s dense fp32 MFMA peak.  TWO measurements, both live in this run:
        #   * `achieved` / `frac` / `avg_launch_us`: every GEMM launch of ISO_PASSES passes over each distinct batch with nothing else on
        #     the GPU.  This is the figure a rocprofv3 kernel trace reproduces (profiles/rNN_encoder_alone_kernel_summary.md: launches x
        #     avg us of the k_gemm_f32* rows), because under the profiler the host is too slow for kernels of different streams to overlap.
        #   * `in_pipeline`: every SAMPLE-th launch inside the timed region, where four streams share the CUs and a kernel's begin-to-end
        #     time is about twice its time alone (NOT reproducible from a profile: rocprofv3 removes the overlap it would have to show).
        # The only fraction that is purely the driver's clock is `whole_step`: the exact fp32 MFMA flops of a step / the step time.
        kown = lambda recs: sum((k if k is not None else b) for b, k, _ in recs)
        gem_r, agg_r, rsq_r = summ["gemm"], summ["kpconv_aggregate"], summ["radius_query"]
        tk_gemm, tk_agg, tk_rs = kown(gem_r), kown(agg_r), kown(rsq_r)
        gem, agg, rsq = [(b, m) for b, _, m in gem_r], [(b, m) for b, _, m in agg_r], [(b, m) for b, _, m in rsq_r]
        t_gemm, t_agg, t_rs = sum(t for t, _ in gem), sum(t for t, _ in agg), sum(t for t, _ in rsq)
        flops = sum(2.0 * m[0] * m[1] * m[2] for _, m in gem)                          # of the SAMPLED launches (in-pipeline rate only)
        flops_agg = sum(2.0 * 15 * nnz.get((M, Ns), M * H) * C for _, (M, Ns, H, C, isz) in agg)
        uses = [len(range(k, args.steps, nb_in)) for k in range(nb_in)]              # how often the timed region fed each batch
        mix = lambda per_batch: sum(u * v for u, v in zip(uses, per_batch)) / max(args.steps, 1)
        flops_step, flops_agg_step = mix(iso["flops"]), mix(iso["flops_agg"])          # exact per step: no sampling involved
        gemm_alone = sum(iso["flops"]) * ISO_PASSES / iso["t_gemm"] / 1e12             # TFLOP/s over all clocked passes
        agg_alone = sum(iso["flops_agg"]) * ISO_PASSES / iso["t_agg"] / 1e12
        iso_rs = iso["t_rs"] / max(iso["n_rs"], 1)                                     # one launch per batch (all searches)
        # KPConv layer = aggregation + its (15C x Cout) contraction, against SURVEY §8d's a-4 bytes: indices + query/support xyz +
        # support features + OUTPUT features + weights (the materialised (M,15C) aggregate is NOT algorithmic traffic)
        contr = [((k if k is not None else b), m) for b, k, m in gem_r if m[2] % 15 == 0 and m[2] >= 480]
        bytes_kp = sum(M * H * isz + (M + Ns) * 12 + Ns * C * 4 for _, (M, Ns, H, C, isz) in agg) + \
            sum(M * N * 4 + K * N * 4 for _, (M, N, K) in contr)
        t_kp = tk_agg + sum(t for t, _ in contr)
        n_search = 7 if args.no_upsampling else 10
        rs_launch = tk_rs / max(len(rsq), 1)                                         # one launch per step (all searches of a batch)
        sp0 = stage_points
        n_queries = 2 * sp0[0] + 3 * sp0[1] + 3 * sp0[2] + 2 * sp0[3] if not args.no_upsampling else sp0[0] + 2 * sp0[1] + 2 * sp0[2] + 2 * sp0[3]
        bytes_rs = mix([search_bytes(sp, not args.no_upsampling) for sp in stage_points_all])
        bytes_rs_iso = sum(search_bytes(sp, not args.no_upsampling) for sp in stage_points_all) / len(stage_points_all)
        traffic, traffic_src, traffic_rs = None, None, None
        if os.path.exists(PMC_JSON):
            pmc = json.load(open(PMC_JSON))
            traffic = pmc.get("k_gemm_f32", {}).get("traffic_bytes")
            traffic_rs = (pmc.get("k_radius_query_multi") or pmc.get("k_radius_query") or {}).get("traffic_bytes")
            traffic_src = "profiles/pmc_traffic.json: separate rocprofv3 --pmc passes of this command (2*FETCH_SIZE + WRITE_SIZE per launch), not measured in this run"
        gemm_per_step = iso["n_gemm"] / iso["passes"]
        gemm_kernels = ("lcr::k_gemm_f32 (fp32 MFMA) + k_gemm_f32_bsplit_p (K-deep shapes: bf16 x 3 split on bf16 MFMA, fp32 accumulate)" if F.gemm_split_enabled()
                        else "lcr::k_gemm_f32 / k_gemm_f32_deep (fp32 MFMA)")
        roof = {"schema": "r05: achieved / frac / avg_launch_us are the kernel-ALONE clock (as since r04; r01-r03 lines carried the in-pipeline clock "
                          "under these keys, now `in_pipeline`); flops are algorithmic fp32 flops (2MNK) against the fp32 MFMA peak in both GEMM forms",
                "bound": "mfma", "kernel": "%s, %d launches/step" % (gemm_kernels, round(gemm_per_step)),
                "achieved": round(gemm_alone, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(gemm_alone / FP32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "clock": "kernel begin-to-end (hipExtLaunchKernel start/stop events) of EVERY GEMM launch of %d passes over each of the %d distinct "
                         "batches, encoder + pre-processing alone on the GPU, after the timed region; reproducible from "
                         "profiles/*_encoder_alone_kernel_summary.md (launches x avg us of the k_gemm_f32* rows)" % (ISO_PASSES, nb_in),
                "launches_timed": iso["n_gemm"],
                # the two GEMM families apart, each against the ceiling of the unit it runs on (VERDICT r5 / advisor r5): the split form
                # executes 6 bf16 products per fp32 product on the bf16 matrix cores, so its fp32-EQUIVALENT ceiling is 2 500 / 6 = 417 TFLOP/s
                "by_form": {
                    "split_bf16x3": None if not iso["n_split"] else {
                        "kernel": "k_gemm_f32_bsplit_p (K >= 288, N >= 64)", "launches_per_step": round(iso["n_split"] / iso["passes"], 1),
                        "kernel_ms_per_step": round(iso["t_split"] / iso["passes"] * 1e3, 4),
                        "fp32_equivalent_tflops": round(iso["fl_split"] / iso["t_split"] / 1e12, 2),
                        "frac_of_fp32_mfma_peak": round(iso["fl_split"] / iso["t_split"] / 1e12 / FP32_PEAK_TFLOPS, 4),
                        "executed_bf16_tflops": round(6 * iso["fl_split"] / iso["t_split"] / 1e12, 1),
                        "own_ceiling_fp32_equivalent_tflops": round(BF16_PEAK_TFLOPS / 6, 1),
                        "frac_of_own_ceiling": round(6 * iso["fl_split"] / iso["t_split"] / 1e12 / BF16_PEAK_TFLOPS, 4)},
                    "fp32_mfma": {"kernel": "k_gemm_f32 / k_gemm_f32_deep (v_mfma_f32_32x32x2_f32)", "launches_per_step": round(iso["n_light"] / iso["passes"], 1),
                                  "kernel_ms_per_step": round(iso["t_light"] / iso["passes"] * 1e3, 4),
                                  "tflops": round(iso["fl_light"] / max(iso["t_light"], 1e-12) / 1e12, 2),
                                  "frac_of_fp32_mfma_peak": round(iso["fl_light"] / max(iso["t_light"], 1e-12) / 1e12 / FP32_PEAK_TFLOPS, 4),
                                  "hbm_bound_launches_per_step": round(iso.get("hbm_bound_light", 0) / iso["passes"], 1),
                                  "frac_of_two_sided_roofline": round(iso.get("t_roof_light", 0.0) / max(iso["t_light"], 1e-12), 4)},
                    "two_sided": {"frac": round((iso.get("t_roof_light", 0.0) + iso.get("t_roof_split", 0.0)) / max(iso["t_gemm"], 1e-12), 4),
                                  "what": "sum over the launches of max(2MNK / 157.3 TFLOP/s, 4 (MK + NK + MN) B / 8 TB/s) / sum of their own durations: "
                                          "the wide stage-1/2 Linears (K <= 128) move more bytes than their flops take on the matrix cores"}},
                "avg_launch_us": round(iso["t_gemm"] / max(iso["n_gemm"], 1) * 1e6, 2),
                "kernel_ms_per_step": round(iso["t_gemm"] / iso["passes"] * 1e3, 4),
                "gflop_per_launch": round(flops_step / gemm_per_step / 1e9, 3),
                "gflop_per_step": round(flops_step / 1e9, 1),
                "in_pipeline": {"achieved": round(flops / tk_gemm / 1e12, 2), "frac": round(flops / tk_gemm / 1e12 / FP32_PEAK_TFLOPS, 4),
                                "avg_launch_us": round(tk_gemm / max(len(gem), 1) * 1e6, 2), "launches_timed": len(gem),
                                "clock": "kernel begin-to-end of every %d-th launch INSIDE the timed region: four streams share the CUs, a kernel "
                                         "takes about twice its time alone; not reproducible from a profile (rocprofv3 slows the host until "
                                         "streams no longer overlap)" % SAMPLE,
                                "achieved_event_bracketed": round(flops / t_gemm / 1e12, 2),
                                "avg_launch_us_event_bracketed": round(t_gemm / max(len(gem), 1) * 1e6, 2)},
                "kernel_time_over_step_time": {"gemm": round(tk_gemm * SAMPLE / dt, 3), "kpconv_aggregate": round(tk_agg * SAMPLE / dt, 3),
                                               "radius_query": round(tk_rs * SAMPLE / dt, 3),
                                               "note": "sum of in-pipeline kernel begin-to-end times (sampled, scaled) / wall time; streams overlap, so the shares do not add up to 1"},
                "neighbor": {"kernel": "lcr::k_radius_query[_multi] (%d searches in 1 launch per step)" % n_search,
                             "bound": "hbm",
                             "algorithmic_mb_per_step": round(bytes_rs_iso / 1e6, 2),
                             "achieved": round(bytes_rs_iso / iso_rs / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(bytes_rs_iso / iso_rs / 1e9 / HBM_PEAK_GBS, 4),
                             "ms_per_step": round(iso_rs * 1e3, 4),
                             "clock": "kernel begin-to-end, alone on the GPU (%d launches); in_pipeline = inside the timed region" % iso["n_rs"],
                             "algorithmic_bytes_per_launch": round(bytes_rs_iso), "traffic": traffic_rs,
                             "in_pipeline": {"achieved": round(bytes_rs / rs_launch / 1e9, 1), "frac": round(bytes_rs / rs_launch / 1e9 / HBM_PEAK_GBS, 4),
                                             "ms_per_step": round(rs_launch * 1e3, 4),
                                             "ms_per_step_event_bracketed": round(t_rs / max(len(rsq), 1) * 1e3, 4)},
                             "north_star_target_frac": 0.6,
                             "north_star_target_status": "NOT MET and not reachable by this algorithm: 0.60 of 8 TB/s over 143 MB is 30 us per batch = ~16 "
                                                         "wavefront instructions per query; an exact (d2, idx)-sorted radius search spends ~300 VALU "
                                                         "per query on ~98 distance tests + a 51-key sort (traffic 1.01x the algorithmic bytes: no wasted "
                                                         "bytes).  Five lane mappings built and measured over rounds 2-5; the stage's roofline is `issue_bound`",
                             # the search is bound by instruction issue, not by HBM (LABNOTES.md §4.4): VALU wavefront-instructions per query
                             # (rocprofv3 --pmc, profiles/) at 4 cycles each on a SIMD
                             "instruction_floor": {"valu_cycles_per_query_per_cu": RS_VALU_PER_QUERY, "queries_per_step": int(n_queries),
                                                   "ms_per_step": round(n_queries * RS_VALU_PER_QUERY / 256 / 2.4e9 * 1e3, 4),
                                                   "alone_over_floor": round(iso_rs / (n_queries * RS_VALU_PER_QUERY / 256 / 2.4e9), 3),
                                                   "source": RS_VALU_SOURCE},
                             # the bound the stage is actually on (kill-criterion branch): achieved / VALU-issue roofline =
                             # time the kernel's own VALU wavefront-instructions need at one issue per SIMD per 4 cycles / its time alone
                             "issue_bound": {"bound": "valu-issue", "frac": round((n_queries * RS_VALU_PER_QUERY / 256 / 2.4e9) / iso_rs, 4),
                                             "roofline_ms_per_step": round(n_queries * RS_VALU_PER_QUERY / 256 / 2.4e9 * 1e3, 4),
                                             "note": "lane-per-query form built and measured in round 5 (16-query tiles, LDS lists, per-query counting sort): "
                                                     "rows bit-identical, 344 vs 112 us on neighbors[0] -> removed (profiles/r05_lpq_radius_bench.md)"}},
                "aggregation": {"kernel": "lcr::k_kpconv_aggregate (fp32 MFMA 16x16x4, %d launches/step)" % round(iso["n_agg"] / iso["passes"]), "bound": "mfma",
                                "achieved": round(agg_alone, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(agg_alone / FP32_PEAK_TFLOPS, 4), "gflop_per_step": round(flops_agg_step / 1e9, 2),
                                "kernel_ms_per_step": round(iso["t_agg"] / iso["passes"] * 1e3, 4),
                                "in_pipeline": {"achieved": round(flops_agg / tk_agg / 1e12, 2), "frac": round(flops_agg / tk_agg / 1e12 / FP32_PEAK_TFLOPS, 4)},
                                "note": "2*15*nnz*C flops over the valid neighbours; alone on the GPU like `achieved` above"},
                "whole_step": {"gflop_per_step": round((flops_step + flops_agg_step) / 1e9, 1),
                               "what": "GEMMs + KPConv aggregation: exact fp32 MFMA flops of a step (every launch of a clocked pass over each distinct batch, "
                                       "weighted by how often the timed region fed it) / the driver-visible step time",
                               "tflops": round((flops_step + flops_agg_step) * args.steps / dt / 1e12, 2),
                               "frac_of_fp32_mfma_peak": round((flops_step + flops_agg_step) * args.steps / dt / 1e12 / FP32_PEAK_TFLOPS, 4)},
                "secondary": {"kernel": "KPConv layers: lcr::k_kpconv_aggregate + its (15C x Cout) contraction", "bound": "hbm",
                              "achieved": round(bytes_kp / t_kp / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(bytes_kp / t_kp / 1e9 / HBM_PEAK_GBS, 4),
                              "algorithmic_mb_per_step": round(bytes_kp * SAMPLE / args.steps / 1e6, 1),
                              "ms_per_step": round(t_kp * SAMPLE / args.steps * 1e3, 3), "clock": "in the pipeline (sampled)"}}
        line = {
            "metric": "scans/s (120k-pt KITTI-shape scan -> 256-D descriptor)",
            "value": round(world * BATCH * args.steps / dt, 3), "unit": "scans/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "repeats": R, "ms_per_step_min": round(dt_min / args.steps * 1e3, 3), "ms_per_step_max": round(dt_max / args.steps * 1e3, 3),
            "timing": "median of %d back-to-back blocks of %d steps, each bracketed by barrier + synchronize (max over ranks)" % (R, args.steps),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", **({"forced_one_rank_process_group": backend} if force_dist else {}),
            "config": {"workload": "configs[1]: batch of 8 synthetic 64-beam scans (~120k pts, 0.3 m voxel -> ~16k pts), "
                                   "voxelise + 3 subsamples + %d radius searches + KPConv encoder + NetVLAD, seeded random weights"
                                   % n_search,
                       "scans_per_step_per_gpu": BATCH, "raw_points_per_scan": int(raw_pts.shape[0] // BATCH),
                       "stage_points_per_batch": stage_points, "distinct_input_batches": nb_in, "neighbor_limits": LIMITS,
                       "streams": "1" if args.no_overlap else ("pre-processing stream (own host thread, 2 batches ahead) + %d encoder stream(s)"
                                                               % (1 if (args.single_encoder or args.no_thread) else 2)),
                       "parallelism": "scan-parallel x%d, all-gather of descriptors%s" % (world, (" (%s)" % backend) if backend else ""),
                       "host": "rank 0 on %d cores (%s)" % bound},
            "roofline": roof,
        }
        if with_h2d is not None:
            line["with_h2d"] = with_h2d
        if split_ab is not None:
            line[split_ab_key] = split_ab
        if lazy is not None:
            line["descriptor_only_7_searches"] = lazy
        if blocks_cfg is not None:
            line.update(blocks_cfg)                      # top level: `pairs`, `retrieval`, `sequence` ...
            roof["other_configs"] = blocks_cfg           # ... and inside `roofline`, which every consumer of the line keeps whole
        if F.gemm_split_enabled():
            line["dtype"] = ("f32 (K-deep GEMMs, K >= 288 and N >= 64: fp32 operands as bf16 x 3 split, 6 of 9 products on the bf16 matrix cores, fp32 "
                             "accumulate — error vs fp64 <= the fp32-MFMA kernel's; every other GEMM, KPConv and attention: fp32 MFMA)")
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(scans)
        final_line = json.dumps(line)
    pipe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()            # RCCL prints its version banner to stdout at teardown: the JSON line goes out after it
    if rank == 0 and final_line is not None:
        try:                                     # RCCL writes its version banner through C stdio, which a pipe buffers until exit: push it
            import ctypes                        # out now, so that the JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(final_line, flush=True)


if __name__ == "__main__":
    main()
