#!/usr/bin/env python
"""Which stage of the descriptor pipeline limits the step?  Times, on the bench workload (8 synthetic scans):
  pre       pre-processing only, one stream                     (GPU time of the latency-bound chain)
  pre-host  host time spent launching it (no final sync)
  enc       encoder + NetVLAD only on a fixed pre-computed batch, one stream
  enc-host  host time spent launching one encoder pass
  enc2      two encoder passes on two streams, launched from one host thread
    python tools/stage_bench.py [--iters 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import bench
    import lcrnet_amd
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.pipeline import DescriptorPipeline
    from lcrnet_amd.weights import seeded_state_dict
    dev = torch.device("cuda:0")
    scans = bench.make_batch(0)
    pts = torch.from_numpy(np.concatenate(scans)).to(dev)
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
    m = create_model()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
    m = m.eval().to(dev)
    pipe = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES,
                              neighbor_limits=bench.LIMITS, upsampling=True, raw_voxel=bench.VOXEL, overlap=False)

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3

    dd = pipe.preprocess(pts, lens)
    t, th = timed(lambda: pipe.preprocess(pts, lens), args.iters)
    print("pre   %.3f ms/batch (host launch time %.3f ms)" % (t, th))
    t, th = timed(lambda: pipe.encode(dd), args.iters)
    print("enc   %.3f ms/batch (host launch time %.3f ms)" % (t, th))
    s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def two():
        for k in range(2):
            with torch.cuda.stream(s[k]):
                pipe.encode(dd)
    t, th = timed(two, args.iters)
    print("enc2  %.3f ms/batch (two streams; host launch time %.3f ms/batch)" % (t / 2, th / 2))
    # host-side cost only: the same code path on scans thinned 50x (GPU work negligible) = pure Python + launch overhead
    tiny = [sc[::50] for sc in scans]
    tp = torch.from_numpy(np.concatenate(tiny)).to(dev)
    tl = torch.tensor([len(x) for x in tiny], dtype=torch.int64, device=dev)
    pipe3 = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES,
                               neighbor_limits=bench.LIMITS, upsampling=True, raw_voxel=bench.VOXEL, overlap=False)
    tdd = pipe3.preprocess(tp, tl)
    t, th = timed(lambda: pipe3.preprocess(tp, tl), args.iters)
    print("pre  on 50x thinned scans %.3f ms/batch (host %.3f)" % (t, th))
    t, th = timed(lambda: pipe3.encode(tdd), args.iters)
    print("enc  on 50x thinned scans %.3f ms/batch (host %.3f)" % (t, th))
    pipe2 = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES,
                               neighbor_limits=bench.LIMITS, upsampling=False, raw_voxel=bench.VOXEL, overlap=False)
    t, th = timed(lambda: pipe2.preprocess(pts, lens), args.iters)
    print("pre (no upsampling lists) %.3f ms/batch (host %.3f)" % (t, th))


if __name__ == "__main__":
    main()
