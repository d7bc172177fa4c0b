"""Run only the encoder + NetVLAD N times over the bench's distinct pre-processed batches, one stream, nothing else on the GPU (for
rocprofv3 --kernel-trace --stats: the kernel-alone durations that bench.py's roofline.achieved / frac / avg_launch_us measure live)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lcrnet_amd.model_family import create_model  # noqa: E402
from lcrnet_amd.pipeline import DescriptorPipeline  # noqa: E402
from lcrnet_amd.weights import seeded_state_dict  # noqa: E402

dev = torch.device("cuda:0")
inputs = bench.rotated_inputs(bench.make_batch(0), int(os.environ.get("LCR_ENC_PROFILE_BATCHES", "4")), dev)
m = create_model()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
m = m.eval().to(dev)
pipe = DescriptorPipeline(m, voxel_size=bench.VOXEL, radius=bench.RADIUS, num_stages=bench.NUM_STAGES, neighbor_limits=bench.LIMITS,
                          upsampling=os.environ.get("LCR_ENC_PROFILE_UPSAMPLING", "0") != "0", raw_voxel=bench.VOXEL, overlap=False)
dds = [pipe.preprocess(p, l) for p, l in inputs]
torch.cuda.synchronize()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    pipe.encode(dds[i % len(dds)])
torch.cuda.synchronize()
