"""Golden vectors for the registration metrics (f-3 / BASELINE config 5) from the IMPORTED reference (build container only).

    python tests/golden/make_golden_registration.py

Runs, unmodified, utils/utils/registration.py: compute_registration_error (:97-113) (-> compute_relative_rotation_error :13-28,
compute_relative_rotation_error_rpy :50-80, compute_relative_translation_error :82-93) and the acceptance / averaging rule of
experiments/registration/eval.py:222-236 restated over the reference's own SummaryBoard (utils/utils/summary_board.py) on
seeded transform pairs: small perturbations (accepted), large ones (rejected), 180-degree turns (wrapped Euler differences),
gimbal lock (pitch = 90 deg), identity.  Output: tests/golden/registration_golden.npz = inputs + expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
REF = "/root/reference"


def rot(axis, deg):
    a = np.asarray(axis, dtype=np.float64)
    a /= np.linalg.norm(a)
    t = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K


def make_pairs(seed=0, n=64):
    rng = np.random.default_rng(seed)
    gts, ests = [], []
    for i in range(n):
        R = rot(rng.standard_normal(3), rng.uniform(0, 180))
        if i % 8 == 1:
            R = rot([0, 1, 0], 90.0) @ rot([0, 0, 1], rng.uniform(-30, 30))      # gimbal lock: R00 = R10 = 0
        if i % 8 == 2:
            R = rot([0, 0, 1], 179.5)                                             # yaw next to the +-180 wrap
        t = rng.uniform(-20, 20, 3)
        scale = [0.5, 3.0, 12.0, 60.0][i % 4]                                     # degrees of perturbation
        dR = rot(rng.standard_normal(3), rng.uniform(0, scale))
        if i % 8 == 2:
            dR = rot([0, 0, 1], 1.0)                                              # crosses the wrap: est yaw = -179.5
        dt = rng.standard_normal(3) * [0.05, 0.5, 1.5, 4.0][(i // 4) % 4]
        G, E = np.eye(4), np.eye(4)
        G[:3, :3], G[:3, 3] = R, t
        E[:3, :3], E[:3, 3] = dR @ R, t + dt
        if i == 0:
            E = G.copy()                                                          # identical: acos(1)
        gts.append(G)
        ests.append(E.astype(np.float32).astype(np.float64) if i % 2 else E)     # half of the estimates carry fp32 rounding like the model's
    return np.stack(gts), np.stack(ests)


def main():
    import make_golden_model as mgm
    mgm.install_stubs()
    sys.path.insert(0, REF)
    from utils.utils.registration import compute_registration_error
    from utils.utils.summary_board import SummaryBoard
    gts, ests = make_pairs()
    errs = np.array([compute_registration_error(g, e) for g, e in zip(gts, ests)], dtype=np.float64)
    meter = SummaryBoard(names=["recall", "rre", "rte", "rx", "ry", "rz"])
    for rre, rte, rx, ry, rz in errs:                      # eval.py:222-236
        accepted = rre < 5.0 and rte < 2.0
        if accepted:
            for k, v in zip(("rre", "rte", "rx", "ry", "rz"), (rre, rte, rx, ry, rz)):
                meter.update(k, v)
        meter.update("recall", float(accepted))
    summary = np.array([meter.mean(k) for k in ("recall", "rre", "rte", "rx", "ry", "rz")])
    print("pairs", len(errs), "RR %.4f RRE %.4f RTE %.4f Rx %.4f Ry %.4f Rz %.4f" % tuple(summary))
    np.savez_compressed(os.path.join(HERE, "registration_golden.npz"), gt=gts, est=ests, errors=errs, summary=summary)


if __name__ == "__main__":
    main()
