#!/usr/bin/env python
"""Where ONE registration pair per call spends its host time (165 -> 124 pairs/s between rounds 4 and 5).
    python tools/pair_host_profile.py
Prints: pairs/s with 1 / 2 / 3 workers at one pair per call (same session), host-issue time vs device time of a call, and the
cProfile top of a single-worker pass (host functions by cumulative time)."""
import cProfile
import io
import os
import pstats
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
work = bb.demo_pairs(dev, 96)


def rate(workers, passes=4):
    with PairPipeline(m, neighbor_limits=bb.PAIR_LIMITS, workers=workers, pairs_per_call=1) as pp:
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


for w in (1, 2, 3, 2, 1):
    print("workers=%d pairs/s per pass: %s" % (w, rate(w)), flush=True)

with PairPipeline(m, neighbor_limits=bb.PAIR_LIMITS, workers=1, pairs_per_call=1) as pp:
    for _ in pp.run(work[:32]):
        pass
    torch.cuda.synchronize()
    # host issue time (no sync) vs device time (event bracket) of single calls
    iss, devt = [], []
    for p, l in work[:32]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        pp.one(p, l)
        iss.append(time.perf_counter() - t0)
        e1.record()
        torch.cuda.synchronize()
        devt.append(e0.elapsed_time(e1) * 1e-3)
    print("one call alone: host time to return %.2f ms (median), device span %.2f ms" % (sorted(iss)[16] * 1e3, sorted(devt)[16] * 1e3), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in pp.run(work[:48]):
        pass
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
