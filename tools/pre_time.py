"""Isolated latency of the pre-processing chain of one bench batch (ms per batch, HIP-synchronised wall clock)."""
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

dev = torch.device("cuda:0")
scans = bench.make_batch(0)
pts = torch.from_numpy(np.concatenate(scans)).to(dev)
lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
m = create_model().eval().to(dev)
pipe = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES, neighbor_limits=bench.LIMITS,
                          upsampling=False, raw_voxel=bench.VOXEL, overlap=False)
for _ in range(5):
    pipe.preprocess(pts, lens)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
t0 = time.perf_counter()
for _ in range(n):
    pipe.preprocess(pts, lens)
torch.cuda.synchronize()
print("pre-processing: %.3f ms per batch" % ((time.perf_counter() - t0) / n * 1e3))
