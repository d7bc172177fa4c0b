"""Isolated per-shape timing of the encoder's GEMM and aggregate launches on the bench batch (library launch log, HIP events)."""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lcrnet_amd import functional as F  # noqa: E402
from lcrnet_amd.model_family import create_model  # noqa: E402
from lcrnet_amd.pipeline import DescriptorPipeline  # noqa: E402
from lcrnet_amd.weights import seeded_state_dict  # noqa: E402

dev = torch.device("cuda:0")
scans = bench.make_batch(0)
pts = torch.from_numpy(np.concatenate(scans)).to(dev)
lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
m = create_model()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
m = m.eval().to(dev)
pipe = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES, neighbor_limits=bench.LIMITS,
                          upsampling=False, raw_voxel=bench.VOXEL, overlap=False)
dd = pipe.preprocess(pts, lens)
for _ in range(3):
    pipe.encode(dd)
torch.cuda.synchronize()
t = F.KernelTimer({"gemm", "kpconv_aggregate", "kpconv_fused"})
F.set_timer(t)
n = 10
for _ in range(n):
    pipe.encode(dd)
torch.cuda.synchronize()
F.set_timer(None)
s = {k: [((kk if kk is not None else b), m) for b, kk, m in v] for k, v in t.records().items()}   # the kernels' own begin-to-end times
for name in ("kpconv_fused", "kpconv_aggregate", "gemm"):
    acc, cnt, order = defaultdict(float), defaultdict(int), []
    for sec, meta in s[name]:
        if meta not in acc:
            order.append(meta)
        acc[meta] += sec
        cnt[meta] += 1
    tot = 0
    for meta in order:
        us = acc[meta] / cnt[meta] * 1e6
        per = cnt[meta] / n
        tot += us * per
        if name == "gemm":
            M, N, K = meta
            print("gemm M=%6d N=%4d K=%4d  x%.0f  %7.1f us  %6.1f TF" % (M, N, K, per, us, 2.0 * M * N * K / us / 1e6))
        elif name == "kpconv_fused":
            print("fused KPConv M=%6d Ns=%6d H=%2d C=%3d  x%.0f  %7.1f us" % (meta[0], meta[1], meta[2], meta[3], per, us))
        else:
            print("aggregate M=%6d Ns=%6d H=%2d C=%3d  x%.0f  %7.1f us" % (meta[0], meta[1], meta[2], meta[3], per, us))
    print("%s total %.1f us per encode" % (name, tot))
