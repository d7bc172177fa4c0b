"""Synthetic 64-beam LiDAR scans (BASELINE.json configs[1]; SURVEY.md §8d "Config 2").

A spinning 64-beam sensor (elevation -24.8°..+2.0°, 1900 azimuth steps -> 121 600 rays, ~120 k returns) at
height 1.73 m is ray-cast against a ground plane plus seeded random axis-aligned boxes (building façades along
a street canyon, parked-car sized boxes, poles) and volumetric vegetation blobs.  Range is clipped at 80 m, returns get 2 cm range noise.  The scene
density is tuned so that 0.3 m voxelisation leaves 16 k ± 2 k points with KITTI-like stage statistics.
There is no equivalent in the reference (it reads KITTI .bin/.npy files); this only provides inputs of the
shape BASELINE.json names.  Pure numpy, deterministic in ``seed``.
"""
import numpy as np

N_BEAMS = 64
N_AZIMUTH = 1900
ELEV_MIN_DEG, ELEV_MAX_DEG = -24.8, 2.0
SENSOR_HEIGHT = 1.73
MAX_RANGE = 80.0


def _scene_boxes(rng):
    """Axis-aligned boxes [xmin,ymin,zmin,xmax,ymax,zmax] in the sensor frame (ground at z=-1.73)."""
    boxes = []
    g = -SENSOR_HEIGHT
    # street canyon: façades on both sides of the x axis with gaps, at |y| in [7, 14] m
    for side in (-1.0, 1.0):
        x = -90.0
        while x < 90.0:
            w = rng.uniform(8.0, 25.0)
            gap = rng.uniform(0.0, 14.0)
            d = rng.uniform(9.0, 26.0)
            depth = rng.uniform(6.0, 15.0)
            h = rng.uniform(4.0, 12.0)
            y0, y1 = (d, d + depth) if side > 0 else (-d - depth, -d)
            boxes.append([x, y0, g, x + w, y1, g + h])
            x += w + gap
    # cross-street blockers ahead / behind
    for sx in (-1.0, 1.0):
        d = rng.uniform(50.0, 75.0)
        boxes.append([sx * d if sx > 0 else -d - 10.0, -40.0, g, sx * d + 10.0 if sx > 0 else -d, 40.0, g + rng.uniform(5.0, 15.0)])
    # parked cars / small objects
    for _ in range(int(rng.integers(14, 24))):
        cx = rng.uniform(-45.0, 45.0)
        cy = rng.choice([-1.0, 1.0]) * rng.uniform(3.0, 6.5)
        l, w, h = rng.uniform(3.5, 5.0), rng.uniform(1.6, 2.0), rng.uniform(1.4, 2.0)
        boxes.append([cx - l / 2, cy - w / 2, g, cx + l / 2, cy + w / 2, g + h])
    # poles / trunks
    for _ in range(int(rng.integers(10, 20))):
        cx, cy = rng.uniform(-50, 50), rng.choice([-1.0, 1.0]) * rng.uniform(5.5, 7.0)
        r = rng.uniform(0.1, 0.3)
        boxes.append([cx - r, cy - r, g, cx + r, cy + r, g + rng.uniform(3.0, 8.0)])
    solid = np.asarray(boxes, dtype=np.float64)
    # vegetation: tree crowns / hedges, volumetric returns (the ray stops at a random depth inside)
    fuzzy = []
    for _ in range(int(rng.integers(30, 45))):
        cx, cy = rng.uniform(-60, 60), rng.choice([-1.0, 1.0]) * rng.uniform(5.0, 24.0)
        sx, sy, sz = rng.uniform(2.5, 6.0), rng.uniform(2.5, 6.0), rng.uniform(2.0, 5.0)
        z0 = g + rng.choice([0.0, rng.uniform(1.5, 3.0)])
        fuzzy.append([cx - sx / 2, cy - sy / 2, z0, cx + sx / 2, cy + sy / 2, z0 + sz])
    return solid, np.asarray(fuzzy, dtype=np.float64)


def synthetic_scan(seed: int, n_azimuth: int = N_AZIMUTH) -> np.ndarray:
    """float32 [N,3] raw scan (N ≈ 120 k), rays in (beam-major) acquisition order."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(ELEV_MAX_DEG, ELEV_MIN_DEG, N_BEAMS))
    azim = np.linspace(0.0, 2.0 * np.pi, n_azimuth, endpoint=False) + rng.uniform(0, 2 * np.pi)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    dirs = np.stack([ce * np.cos(azim)[None, :], ce * np.sin(azim)[None, :], np.broadcast_to(se, (N_BEAMS, n_azimuth))], -1)
    dirs = dirs.reshape(-1, 3)
    t_hit = np.full(dirs.shape[0], np.inf)
    # ground plane z = -h
    dz = dirs[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(dz < -1e-6, -SENSOR_HEIGHT / dz, np.inf)
    t_hit = np.minimum(t_hit, tg)
    # boxes (slab test, origin at 0)
    solid, fuzzy = _scene_boxes(rng)
    inv = 1.0 / np.where(np.abs(dirs) < 1e-12, 1e-12, dirs)
    for k, b in enumerate(np.concatenate([solid, fuzzy], 0)):
        t0 = b[None, :3] * inv
        t1 = b[None, 3:] * inv
        tn = np.minimum(t0, t1).max(axis=1)
        tf = np.maximum(t0, t1).min(axis=1)
        hit = (tf >= np.maximum(tn, 0.0)) & (tn > 0.5)
        if k >= len(solid):  # vegetation: 70 % of the rays are stopped, at a uniform depth inside the volume
            u = rng.random(dirs.shape[0])
            hit &= rng.random(dirs.shape[0]) < 0.7
            tn = tn + u * (tf - tn)
        t_hit = np.where(hit, np.minimum(t_hit, tn), t_hit)
    ok = np.isfinite(t_hit) & (t_hit < MAX_RANGE) & (t_hit > 1.5)
    t = t_hit[ok] + rng.normal(0.0, 0.02, size=int(ok.sum()))
    pts = dirs[ok] * t[:, None]
    return np.ascontiguousarray(pts.astype(np.float32))
