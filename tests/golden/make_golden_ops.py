"""Generate the golden vectors for the native ops (a-1 grid subsample, a-2 radius search, a-3 precompute).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_ops.py

Source of truth = the REFERENCE's own C++ (utils/extensions/cpu/{grid_subsampling,radius_neighbors}) compiled from
/root/reference by oracle/Makefile into oracle/_ref/libref_ops.so and driven through oracle.ops(impl='ref').
Inputs: the six KITTI scans shipped with the reference demo (demo/data_demo/*.npy, xyz columns; they are data files,
re-saved here as float32 [N,3]) and two seeded synthetic 64-beam scans (lcr-net_amd/synthetic.py, voxelised at 0.3 m by
the reference op).

What is stored (tests/golden/ops_golden.npz + scans/*.npy):
  * per scan, per stage: SHA-256 of the reference's stage points (bit pattern) and the stage lengths;
  * per scan, per index tensor (4 neighbours, 3 subsampling, 3 upsampling; limits [74,68,70,67] = the reference demo's
    calibrated values):
      - sha_raw   : SHA-256 of the reference output as returned (nanoflann order inside equal-d² runs),
      - sha_canon : SHA-256 after re-ordering every row by (d², index) — the canonical tie order
                    (cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:125-208) that the build targets,
      - n_rows_permuted : rows where raw != canonical (tie permutations only),
      - n_rows_set_diff : rows whose SET of kept neighbours differs between raw and canonical after the limit cut
                          (a tie run straddling the cut) — must be reported, expected 0 on these inputs,
      - counts_sha / max_count : uncapped in-radius count per row (SHA) and the reference's output width;
  * for one small cropped cloud (2 048 points of scan 003854, pair-stacked with 1 500 points of 000958) the full tensors.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ops  # noqa: E402
import lcrnet_amd.synthetic as synthetic  # noqa: E402

REF_DEMO = "/root/reference/demo/data_demo"
OUT = os.path.dirname(os.path.abspath(__file__))
LIMITS = [74, 68, 70, 67]
NUM_STAGES, VOXEL, RADIUS = 4, 0.3, 1.275


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def d2_f32(q, s):
    """((dx*dx)+dy*dy)+dz*dz in fp32, element-wise (numpy float32 ops round once each, no FMA)."""
    d = (q - s).astype(np.float32)
    sq = (d * d).astype(np.float32)
    return ((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32)


def canonicalise(idx, q, s):
    """Re-order each row's valid entries by (d², index); pads (== len(s)) stay at the end."""
    ns = s.shape[0]
    s_pad = np.concatenate([s, np.full((1, 3), 1e9, np.float32)], 0)
    d2 = d2_f32(q[:, None, :], s_pad[idx])
    d2 = np.where(idx == ns, np.float32(np.inf), d2)
    order = np.lexsort((idx, d2), axis=1)
    return np.take_along_axis(idx, order, axis=1)


def one_search(q, s, ql, sl, radius, limit):
    full = ops.radius_search(q, s, ql, sl, radius, -1, impl="ref")          # reference op, full width
    ns = s.shape[0]
    counts = (full != ns).sum(1).astype(np.int32)
    canon_full = canonicalise(full, q, s)
    raw = np.ascontiguousarray(full[:, :limit])
    canon = np.ascontiguousarray(canon_full[:, :limit])
    if raw.shape[1] < limit:   # reference returns fewer columns than the limit when max count < limit
        pass
    perm = int((raw != canon).any(1).sum())
    setdiff = int((np.sort(raw, 1) != np.sort(canon, 1)).any(1).sum())
    return raw, canon, counts, int(full.shape[1]), perm, setdiff


def stack_record(points, lengths, tag, store, full=False):
    ref = ops.precompute_data_stack_mode(points, lengths, NUM_STAGES, VOXEL, RADIUS, [-1] * NUM_STAGES, impl="ref")
    pts, lens = ref["points"], ref["lengths"]
    for i in range(NUM_STAGES):
        store[f"{tag}/points{i}_sha"] = sha(pts[i])
        store[f"{tag}/lengths{i}"] = lens[i]
        if full:
            store[f"{tag}/points{i}"] = pts[i]
    r = RADIUS
    for i in range(NUM_STAGES):
        specs = [("neighbors", pts[i], pts[i], lens[i], lens[i], r, LIMITS[i])]
        if i < NUM_STAGES - 1:
            specs.append(("subsampling", pts[i + 1], pts[i], lens[i + 1], lens[i], r, LIMITS[i]))
            specs.append(("upsampling", pts[i], pts[i + 1], lens[i], lens[i + 1], r * 2, LIMITS[i + 1]))
        for name, q, s, ql, sl, rad, lim in specs:
            raw, canon, counts, width, perm, setdiff = one_search(q, s, ql, sl, rad, lim)
            k = f"{tag}/{name}{i}"
            store[k + "_sha_raw"] = sha(raw)
            store[k + "_sha_canon"] = sha(canon)
            store[k + "_shape"] = np.array(canon.shape)
            store[k + "_counts_sha"] = sha(counts)
            store[k + "_max_count"] = width
            store[k + "_n_rows_permuted"] = perm
            store[k + "_n_rows_set_diff"] = setdiff
            if full:
                store[k] = canon.astype(np.int32)
                store[k + "_raw"] = raw.astype(np.int32)
                store[k + "_counts"] = counts
            print(f"{k}: shape {canon.shape} width {width} permuted rows {perm} set-diff rows {setdiff}")
        r *= 2


def main():
    assert ops.have_ref(), "build oracle/_ref first: make -C oracle ref"
    store = {}
    scans = {}
    for f in sorted(os.listdir(REF_DEMO)):
        xyz = np.ascontiguousarray(np.load(os.path.join(REF_DEMO, f))[:, :3].astype(np.float32))
        name = f[:-4]
        np.save(os.path.join(OUT, "scans", name + ".npy"), xyz)
        scans[name] = xyz
    for seed in (0, 1):
        raw = synthetic.synthetic_scan(seed)
        ds, dl = ops.grid_subsample(raw, np.array([len(raw)]), VOXEL, impl="ref")
        store[f"syn{seed}/raw_sha"] = sha(raw)
        store[f"syn{seed}/raw_n"] = len(raw)
        store[f"syn{seed}/voxel03_sha"] = sha(ds)
        store[f"syn{seed}/voxel03_n"] = len(ds)
        scans[f"syn{seed}"] = ds
    for name, xyz in scans.items():
        stack_record(xyz, np.array([len(xyz)], dtype=np.int64), name, store)
    # the reference demo pair, stacked [ref, src] as registration_collate_fn_stack_mode does (data.py:110-113)
    pair = np.concatenate([scans["003854"], scans["000958"]], 0)
    stack_record(pair, np.array([len(scans["003854"]), len(scans["000958"])], dtype=np.int64), "pair_003854_000958", store)
    # small stacked case with full tensors
    small = np.concatenate([scans["003854"][:2048], scans["000958"][:1500]], 0)
    store["small/points_in"] = small
    stack_record(small, np.array([2048, 1500], dtype=np.int64), "small", store, full=True)
    store["names"] = np.array(sorted(scans.keys()))
    np.savez_compressed(os.path.join(OUT, "ops_golden.npz"), **store)
    print("wrote", os.path.join(OUT, "ops_golden.npz"))


if __name__ == "__main__":
    main()
