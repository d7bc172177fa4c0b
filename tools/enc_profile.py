"""Run only the encoder + NetVLAD on one pre-processed bench batch N times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

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
pts = torch.from_numpy(np.concatenate(scans)).to(dev)
lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
m = create_model()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
m = m.eval().to(dev)
pipe = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES, neighbor_limits=bench.LIMITS,
                          upsampling=False, raw_voxel=bench.VOXEL, overlap=False)
dd = pipe.preprocess(pts, lens)
torch.cuda.synchronize()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    pipe.encode(dd)
torch.cuda.synchronize()
