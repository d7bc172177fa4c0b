"""Inside the bench pipeline: how long an encoder pass occupies its stream (events around every pass) and how long the stream then
sits idle until its next pass starts.  Two encoder streams alternate, so a saturated pair shows pass time ~ 2 x step time and no idle."""
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
inputs = []
for k in range(4):
    a = np.deg2rad(37.0 * k)
    R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    inputs.append((torch.from_numpy(np.concatenate([s @ R.T for s in scans]).astype(np.float32)).to(dev),
                   torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)))
m = create_model()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
m = m.eval().to(dev)
pipe = DescriptorPipeline(m, bench.VOXEL, bench.RADIUS, bench.NUM_STAGES, bench.LIMITS, upsampling=True, raw_voxel=bench.VOXEL, pre_workers=2, depth=2)
pipe.enable_dual_encoder(int(os.environ.get("LCR_ENC_STREAMS", "2")))
evs = []
orig = pipe.encode


def enc(dd):
    s = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(s)
    t0 = time.perf_counter()
    out = orig(dd)
    host = time.perf_counter() - t0
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(s)
    evs.append((e0, e1, s.cuda_stream, host))
    return out


def run(n):
    for item in pipe.run((inputs[k % 4] for k in range(n)), sync_to_caller=False):
        pass
    torch.cuda.synchronize()


run(12)
pipe.encode = enc
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
run(N)
dt = time.perf_counter() - t0
print("step %.3f ms (%d steps)" % (dt / N * 1e3, N))
ev = evs[20:-4]
dur = [a.elapsed_time(b) for a, b, _, _ in ev]
host = [h * 1e3 for _, _, _, h in ev]
by = {}
for e in ev:
    by.setdefault(e[2], []).append(e)
idle = []
for s, lst in by.items():
    for p, q in zip(lst[:-1], lst[1:]):
        idle.append(p[1].elapsed_time(q[0]))
print("encoder pass on its stream: mean %.3f ms (min %.3f, max %.3f); host time to issue a pass: mean %.3f ms" % (np.mean(dur), min(dur), max(dur), np.mean(host)))
print("stream idle between its passes: mean %.3f ms (min %.3f, max %.3f)" % (np.mean(idle), min(idle), max(idle)))
print("pipeline stats:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in pipe.stats.items()})
