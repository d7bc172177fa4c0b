#!/usr/bin/env python
"""One pair per call: pairs/s per pass for a few host-side settings (interpreter switch interval, workers).  Same session A/B."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_blocks as bb  # noqa: E402
from lcrnet_amd.pipeline import PairPipeline  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m = bb.pair_model(dev)
work = bb.demo_pairs(dev, 192)


def rate(workers, passes=6, P=1):
    with PairPipeline(m, neighbor_limits=bb.PAIR_LIMITS, workers=workers, pairs_per_call=P) as pp:
        for _ in pp.run(work * 2):
            pass
        torch.cuda.synchronize()
        out = []
        for _ in range(passes):
            t0 = time.perf_counter()
            for _ in pp.run(work):
                pass
            torch.cuda.synchronize()
            out.append(round(len(work) / (time.perf_counter() - t0), 1))
    return out


print("keep_gil", os.environ.get("LCR_CTYPES_KEEP_GIL", "1"), flush=True)
for si in (5e-3,):
    sys.setswitchinterval(si)
    for w in (2, 3):
        r = rate(w)
        print("switchinterval %.0e workers=%d: %s  median %.1f min/median %.3f" % (si, w, r, sorted(r)[len(r) // 2], min(r) / sorted(r)[len(r) // 2]), flush=True)
sys.setswitchinterval(5e-3)
print("workers=1:", rate(1, 4), flush=True)
print("P=16 workers=3:", rate(3, 4, 16), flush=True)
