"""CPU: the N>1 path of the registration evaluation (BASELINE configs[4]: pairs dealt to the ranks, per-rank sums, one all-reduce)
on gloo, world_size 2: the reduced summary equals registration_summary over all pairs in one process."""
import json
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pairs(n=37, seed=5):
    """(gt, est) with small, large and failing errors mixed (some pairs rejected by the 5 deg / 2 m rule)."""
    rng = np.random.default_rng(seed)
    gts, ests = [], []
    for i in range(n):
        def rigid(angle_deg, t):
            ax = rng.standard_normal(3)
            ax /= np.linalg.norm(ax)
            a = np.deg2rad(angle_deg)
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = R, t
            return T
        gt = rigid(rng.uniform(0, 180), rng.uniform(-20, 20, 3))
        err = rigid(rng.choice([0.3, 2.0, 4.9, 7.0, 40.0]), rng.standard_normal(3) * rng.choice([0.05, 0.8, 3.0]))
        gts.append(gt.astype(np.float32))
        ests.append((err @ gt).astype(np.float32))
    return gts, ests


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lcrnet_amd import evaluation as ev
    gts, ests = _pairs()
    mine = range(rank, len(gts), world)                            # pairs dealt round-robin, like tools/pair_bench.py --gpus N
    summary = ev.registration_reduce(ev.registration_partial([gts[i] for i in mine], [ests[i] for i in mine]))
    json.dump(summary, open(os.path.join(out_dir, f"r{rank}.json"), "w"))
    dist.destroy_process_group()


def test_sharded_registration_summary_equals_single_process(tmp_path):
    from lcrnet_amd import evaluation as ev
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    gts, ests = _pairs()
    want = ev.registration_summary(gts, ests)
    assert 0 < want["accepted"] < want["pairs"]                     # the case exercises both branches
    for r in range(world):
        got = json.load(open(tmp_path / f"r{r}.json"))
        assert got["pairs"] == want["pairs"] and got["accepted"] == want["accepted"]
        for k in ("RR", "RRE", "RTE", "Rx", "Ry", "Rz"):
            assert abs(got[k] - want[k]) < 1e-9, k


def test_reduce_without_process_group_is_the_local_summary():
    from lcrnet_amd import evaluation as ev
    gts, ests = _pairs(11, 3)
    a, b = ev.registration_reduce(ev.registration_partial(gts, ests)), ev.registration_summary(gts, ests)
    assert a["pairs"] == b["pairs"] and a["accepted"] == b["accepted"]
    for k in ("RR", "RRE", "RTE", "Rx", "Ry", "Rz"):
        assert (np.isnan(a[k]) and np.isnan(b[k])) or abs(a[k] - b[k]) < 1e-9
    none = ev.registration_reduce(ev.registration_partial([], []))
    assert none["pairs"] == 0 and none["RR"] == 0.0 and np.isnan(none["RRE"])
