"""Where a group of P registration pairs spends its time: host wall time to ISSUE each stage of LCRNet.forward_pairs (no sync) and
the stage's time with a device synchronisation after it (single stream, nothing else on the GPU)."""
import itertools
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcrnet_amd import functional as F  # noqa: E402
from lcrnet_amd.config import make_cfg  # noqa: E402
from lcrnet_amd.data import precompute_batch  # noqa: E402
from lcrnet_amd.model_family import LCRNet  # noqa: E402
from lcrnet_amd.weights import seeded_state_dict  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
limits = [74, 68, 70, 67]
cfg = make_cfg()
cfg["neighbor_limits"] = limits
m = LCRNet(cfg).eval()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
m = m.to(dev)
gold = os.path.join(ROOT, "tests", "golden", "scans")
names = sorted(f[:-4] for f in os.listdir(gold) if f.endswith(".npy"))
scans = {n: torch.from_numpy(np.load(os.path.join(gold, n + ".npy"))).to(dev) for n in names}
combos = list(itertools.combinations(names, 2))[:P]
points = torch.cat([torch.cat([scans[a], scans[b]]) for a, b in combos]).contiguous()
lengths = torch.tensor([len(scans[x]) for ab in combos for x in ab], dtype=torch.int64, device=dev)

acc = {}


def timed(name, fn, sync):
    t0 = time.perf_counter()
    r = fn()
    if sync:
        torch.cuda.synchronize()
    acc.setdefault(name, []).append(time.perf_counter() - t0)
    return r


def wrap(obj, attr, name, sync):
    target = getattr(obj, attr)
    if isinstance(target, torch.nn.Module):                 # a child module: time its forward
        obj, attr, target = target, "forward", target.forward
    orig = target
    object.__setattr__(obj, attr, lambda *a, **k: timed(name, lambda: orig(*a, **k), sync))
    return obj, attr, orig


def run_once(sync):
    dd = timed("collate", lambda: precompute_batch(points, lengths, 4, 0.3, 1.275, limits, upsampling=True), sync)
    del dd["segment_lengths"]
    dd["features"] = torch.ones(points.shape[0], 1, device=dev)
    dd["lengths_c_host"] = dd["lengths_host"][-1]
    with torch.no_grad():
        return timed("forward_pairs (total)", lambda: m.forward_pairs(dd), sync)


for sync in (False, True):
    acc.clear()
    saved = []
    for obj, attr, name in ((m, "encoder", "encoder"), (m, "transformer", "transformer"), (m.netvlad, "describe", "netvlad"), (m, "vote_encoder", "vote_encoder"),
                            (m, "kpdecoder", "kpdecoder"), (m, "_dense_matching_group", "dense matching (group)"),
                            (m, "_local_global_registration", "  of which LGR (per pair, summed)"),
                            (F, "log_optimal_transport", "  of which optimal transport (node + patch)"),
                            (F, "point_to_node_partition", "  of which partition (per cloud, summed)"), (F, "top1_matching", "  of which top-1 matching")):
        saved.append(wrap(obj, attr, name, sync))
    for _ in range(3):
        run_once(sync)
    acc.clear()
    n = 5
    for _ in range(n):
        run_once(sync)
    torch.cuda.synchronize()
    print("---- %s, P = %d pairs per call, ms per call (per pair)" % ("stage + synchronize" if sync else "host issue time only", P))
    for k, v in acc.items():
        tot = sum(v) / n * 1e3
        print("%-46s %8.3f  (%.3f)" % (k, tot, tot / P))
    for obj, attr, orig in saved:
        object.__setattr__(obj, attr, orig)
