"""Whole native collate (raw scans -> 0.3 m voxels -> 3 subsamples -> 10 searches, one native call) vs the C++ oracle on random
stacks of decimated synthetic scans under random rigid motions (large translations stress the fp32 voxel arithmetic):
    python tools/fuzz_collate.py FIRST_SEED LAST_SEED [dense] [--json FILE] [--max-seconds S]
Odd seeds go in as KITTI velodyne rows f32[N,4] (x, y, z + a junk fourth column): the strided raw-scan ingest of round 4."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import ops as O
import lcrnet_amd.synthetic as synthetic
from lcrnet_amd.data import precompute_batch_native
t0 = time.time(); bad = []; n = 0
MAXS = float(sys.argv[sys.argv.index("--max-seconds") + 1]) if "--max-seconds" in sys.argv else 900.0
base = {i: synthetic.synthetic_scan(i) for i in range(6)}
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    dense = len(sys.argv) > 3 and sys.argv[3] == "dense"          # fuller scans, up to 8 clouds per stack
    B = int(rng.integers(1, 9 if dense else 5))
    clouds = []
    for _ in range(B):
        s = base[int(rng.integers(0, 6))][:: int(rng.integers(1, 6) if dense else rng.integers(4, 40))]
        a = rng.uniform(0, 2 * np.pi)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
        t = (rng.normal(0, [300, 300, 20])).astype(np.float32)
        if rng.random() < 0.3:
            t = np.round(t / 0.3) * 0.3                       # translations commensurate with the voxel: points on cell faces
        clouds.append((s @ R.T + t).astype(np.float32))
    xyz = np.concatenate(clouds); lens = np.array([len(c) for c in clouds], dtype=np.int64)
    limits = [int(rng.integers(8, 40))] * 4
    try:
        p0, l0 = O.grid_subsample(xyz, lens, 0.3)
        want = O.precompute_data_stack_mode(p0, l0, 4, 0.3, 1.275, limits)
        rows = np.concatenate([xyz, rng.standard_normal((len(xyz), 1)).astype(np.float32) * 1e3], axis=1) if seed % 2 else xyz
        got = precompute_batch_native(torch.from_numpy(np.ascontiguousarray(rows)).cuda(), torch.from_numpy(lens).cuda(), 4, 0.3, 1.275, limits, raw_voxel=0.3)
        torch.cuda.synchronize()
        ok = True
        for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
            for x, y in zip(want[key], got[key]):
                yy = y.cpu().numpy()
                if key in ("neighbors", "subsampling", "upsampling"):
                    w = min(x.shape[1], yy.shape[1])
                    ok &= np.array_equal(x[:, :w].astype(np.int64), yy[:, :w].astype(np.int64)) and (yy[:, w:] == yy.max()).all() if yy.size else True
                elif key == "points":
                    ok &= np.array_equal(x.view(np.uint32), yy.view(np.uint32))
                else:
                    ok &= np.array_equal(x, yy)
        n += 1
        if not ok:
            bad.append(seed); print("MISMATCH seed", seed)
    except Exception as e:
        bad.append(seed); print("EXC seed", seed, repr(e)[:300])
    if time.time() - t0 > MAXS: break
rec = {"tool": "fuzz_collate", "mode": "dense" if (len(sys.argv) > 3 and sys.argv[3] == "dense") else "sparse", "first": int(sys.argv[1]), "last_done": seed,
       "configs_exact": n - len(bad), "failures": bad, "odd_seeds_as_xyzi_rows": True, "seconds": round(time.time() - t0, 1)}
print("collate fuzz: " + json.dumps(rec))
if "--json" in sys.argv:
    open(sys.argv[sys.argv.index("--json") + 1], "a").write(json.dumps(rec) + "\n")
