"""Upper bound of the bench pipeline if the pre-processing were free: the encoder + NetVLAD alone over four pre-processed bench batches,
passes dealt round-robin onto S encoder streams (python tools/encoder_only_rate.py [passes] ; LCR_ENC_STREAMS=S)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lcrnet_amd.model_family import create_model  # noqa: E402
from lcrnet_amd.pipeline import DescriptorPipeline, distinct_queue_streams, release_streams  # noqa: E402
from lcrnet_amd.weights import seeded_state_dict  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = int(os.environ.get("LCR_ENC_STREAMS", "2"))
dev = torch.device("cuda:0")
scans = bench.make_batch(0)
m = create_model()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
m = m.eval().to(dev)
pipe = DescriptorPipeline(m, bench.VOXEL, bench.RADIUS, bench.NUM_STAGES, bench.LIMITS, upsampling=True, raw_voxel=bench.VOXEL, pre_workers=2, depth=2)
dds = []
for k in range(4):
    a = np.deg2rad(37.0 * k)
    R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    pts = torch.from_numpy(np.concatenate([s @ R.T for s in scans]).astype(np.float32)).to(dev)
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
    dds.append(pipe.preprocess(pts, lens))
torch.cuda.synchronize()
streams = distinct_queue_streams(dev, S) if S > 1 else [torch.cuda.current_stream(dev)]
res = {}
with torch.no_grad():
    for rep in range(3):
        for k in range(8):                                   # warm-up on every stream
            with torch.cuda.stream(streams[k % S]):
                pipe.encode(dds[k % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(passes):
            with torch.cuda.stream(streams[k % S]):
                pipe.encode(dds[k % 4])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["rep%d" % rep] = {"ms_per_pass": round(1e3 * dt / passes, 4), "scans_per_s": round(8 * passes / dt, 1)}
if S > 1:
    release_streams(streams)
print(json.dumps({"tool": "encoder_only_rate", "streams": S, "passes": passes, **res}))
