"""CPU, build container only: the oracle (oracle/lcr_oracle.cpp) against the REFERENCE's own compiled C++ (oracle/_ref/libref_ops.so) on
the random clouds of the op fuzz plus the degenerate classes it found (planes / lines / points at coordinates that land one cell below the
voxel origin, far offsets): grid subsample bit-exact incl. order, radius search equal as sets per row (the reference's kd-tree orders
equal-distance ties differently).
    python tools/fuzz_oracle_vs_ref.py FIRST LAST [--json FILE] [--max-seconds S]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ops  # noqa: E402

def below_origin(v, rng):
    """a coordinate c with floor(c * fl(1/v)) * v > c in fp32 (none exist for power-of-two voxels)"""
    v32, inv = np.float32(v), np.float32(1.0) / np.float32(v)
    for _ in range(4000):
        k = int(rng.integers(-1500, 1500))
        for c in (np.float32(k * v), np.nextafter(np.float32(k * v), np.float32(1e9)), np.nextafter(np.float32(k * v), np.float32(-1e9))):
            if np.float32(np.floor(np.float32(c * inv)) * v32) > c:
                return c
    return None


def make_case(seed):
    """(xyz f32[N,3], lens i64[B], voxel) of one fuzz case"""
    rng = np.random.default_rng(900 + seed)
    voxel = float(rng.choice([0.25, 0.3, 0.5, 0.6, 0.77, 1.3]))
    clouds = []
    for _ in range(int(rng.integers(1, 6))):
        m = int(rng.integers(1, 1500))
        kind = int(rng.integers(0, 8))
        if kind == 0:
            p = rng.random((m, 3)) * rng.uniform(1, 60, 3)
        elif kind == 1:
            p = np.round(rng.random((m, 3)) * 40) * 0.25
        elif kind == 2:
            p = np.concatenate([rng.random((m, 2)) * 50, np.zeros((m, 1))], 1)
        elif kind == 3:
            p = np.outer(rng.random(m) * 80, rng.standard_normal(3))
        elif kind == 4:
            p = rng.standard_normal((m, 3)) * 0.3 + rng.integers(0, 5, (m, 1)) * 7.0
        elif kind == 5:
            p = rng.standard_normal((m, 3)) * 15
        else:                                            # constant coordinates below the voxel origin on 1..3 axes
            p = rng.random((m, 3)) * 30
            for ax in rng.permutation(3)[: int(rng.integers(1, 4))]:
                c = below_origin(voxel, rng)
                if c is not None:
                    p[:, ax] = c
            clouds.append(p.astype(np.float32))
            continue
        clouds.append((p + rng.normal(0, rng.choice([0, 200, 5000]), 3)).astype(np.float32))
    xyz = np.concatenate(clouds)
    lens = np.array([len(c) for c in clouds], dtype=np.int64)
    return xyz, lens, voxel, rng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int)
    ap.add_argument("last", type=int)
    ap.add_argument("--json", default=None)
    ap.add_argument("--max-seconds", type=float, default=0.0)
    args = ap.parse_args()
    if not ops.have_ref():
        sys.exit("oracle/_ref/libref_ops.so not built (needs /root/reference)")
    t0, bad, n, seed = time.time(), [], 0, args.first - 1
    for seed in range(args.first, args.last):
        xyz, lens, voxel, rng = make_case(seed)
        try:
            a, al = ops.grid_subsample(xyz, lens, voxel)
            b, bl = ops.grid_subsample(xyz, lens, voxel, impl="ref")
            assert np.array_equal(al, bl), "subsample lengths"
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "subsample points / order"
            radius = voxel * float(rng.choice([2.5, 4.25]))
            ra = ops.radius_search(a, xyz, al, lens, radius, -1)
            rb = ops.radius_search(a, xyz, al, lens, radius, -1, impl="ref")
            assert ra.shape == rb.shape and np.array_equal(np.sort(ra, 1), np.sort(rb, 1)), "cross search sets"
            sa = ops.radius_search(a, a, al, al, radius, -1)
            sb = ops.radius_search(a, a, al, al, radius, -1, impl="ref")
            assert sa.shape == sb.shape and np.array_equal(np.sort(sa, 1), np.sort(sb, 1)), "self search sets"
            n += 1
        except AssertionError as e:
            bad.append({"seed": seed, "assert": str(e)})
            print("FAIL", bad[-1], flush=True)
        if args.max_seconds and time.time() - t0 > args.max_seconds:
            break
    rec = {"tool": "fuzz_oracle_vs_ref", "first": args.first, "last_done": seed, "cases_equal": n, "failures": bad, "seconds": round(time.time() - t0, 1)}
    print("oracle vs compiled reference: " + json.dumps(rec))
    if args.json:
        with open(args.json, "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
