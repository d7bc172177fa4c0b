"""Latency of ONE raw scan -> 256-D descriptor (no pipelining, one stream, host-synchronised): the online loop-closing use of the
reference (one scan at a time, experiments/inference/infer_loop_detection_find_top1.py).  Prints ms per scan for batch sizes 1, 2, 4, 8."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lcrnet_amd.model_family import create_model  # noqa: E402
from lcrnet_amd.pipeline import DescriptorPipeline  # noqa: E402
from lcrnet_amd.weights import seeded_state_dict  # noqa: E402

dev = torch.device("cuda:0")
scans = bench.make_batch(0)
m = create_model()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
m = m.eval().to(dev)
pipe = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES, neighbor_limits=bench.LIMITS,
                          upsampling=False, raw_voxel=bench.VOXEL, overlap=False)
for B in (1, 2, 4, 8):
    pts = torch.from_numpy(np.concatenate(scans[:B])).to(dev)
    lens = torch.tensor([len(s) for s in scans[:B]], dtype=torch.int64, device=dev)
    for _ in range(5):
        pipe.encode(pipe.preprocess(pts, lens))
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        d = pipe.encode(pipe.preprocess(pts, lens))
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("batch %d: %.3f ms per call, %.3f ms per scan (%.0f scans/s, unpipelined)" % (B, dt * 1e3, dt * 1e3 / B, B / dt))
