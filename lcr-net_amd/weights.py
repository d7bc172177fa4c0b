"""Checkpoint layout + deterministic seeded weights.

Checkpoints are read exactly like the reference: ``torch.load(path, map_location='cpu', weights_only=True)['model']``
then ``load_state_dict(strict=False)`` with the DDP ``module.`` prefix handled (utils/engine/base_tester.py:111-122).

There is no trained checkpoint in the build/GPU containers, so tests and bench use *seeded* weights: every tensor of a
state dict is filled from a CPU ``torch.Generator`` seeded by (seed, crc32(key)) — a pure function of the key name and
shape, so the reference model (when generating goldens) and this package's model (when testing) get identical values
without shipping an 88 MB file.  KPConv kernel points — buffers that live in the checkpoint (kpconv/kpconv.py:64-65) —
are the 15-point disposition scaled by the layer radius, z-rotated and jittered like kernel_points.py:426-455, but from
the same seeded generator.
"""
import os
import zlib

import numpy as np
import torch

_DISPOSITION = os.path.join(os.path.dirname(os.path.abspath(__file__)), "modules", "kpconv", "dispositions",
                            "k_015_center_3D.npy")


def base_kernel_points():
    """float64 [15,3] unit-sphere disposition (value copy of the reference's k_015_center_3D.ply data)."""
    return np.load(_DISPOSITION)


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def seeded_tensor(key, shape, dtype, seed, radius=None):
    g = _gen(seed, key)
    shape = tuple(shape)
    leaf = key.split(".")[-1]
    if dtype in (torch.int64, torch.int32):
        return torch.zeros(shape, dtype=dtype)                      # num_batches_tracked
    if leaf == "kernel_points":
        assert radius is not None, "kernel_points need the layer radius"
        kp = torch.from_numpy(base_kernel_points()).double()
        theta = float(torch.rand((), generator=g, dtype=torch.float64)) * 2 * np.pi
        c, s = np.cos(theta), np.sin(theta)
        R = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
        kp = kp + 0.01 * torch.randn(kp.shape, generator=g, dtype=torch.float64)
        return ((radius * kp) @ R).float()
    if leaf == "running_var":
        return (0.5 + torch.rand(shape, generator=g)).to(dtype)
    if leaf == "running_mean":
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "alpha":
        return torch.ones(shape, dtype=dtype)
    if len(shape) <= 1:
        if leaf == "weight":                                        # GroupNorm / LayerNorm / BatchNorm scale
            return (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)    # biases
    if leaf == "weights" and len(shape) == 3 and "KPConv" in key:   # (K, Cin, Cout)
        fan = shape[0] * shape[1] / 3.0
    elif leaf == "cluster_weights":
        fan = 1024.0 / 64.0                                         # sharp soft-assignments (trained-model-like), not uniform ones
    elif leaf == "hidden1_weights":
        fan = 1.0 / 16.0                                            # O(1) pre-BN outputs so the descriptor is data-, not bias-dominated
    elif leaf == "cluster_weights2":
        fan = 1024.0
    elif leaf == "gating_weights":
        fan = float(shape[0])
    elif leaf == "ctr_reg":
        fan = float(shape[-1])
    else:                                                           # nn.Linear weight (out, in)
        fan = float(shape[-1])
    return (torch.randn(shape, generator=g) / np.sqrt(fan)).to(dtype)


def kpconv_radius_of(key, init_radius):
    """Radius argument of the KPConv layer that owns ``key`` (backbone4.py:15-58, :102-118)."""
    parts = key.split(".")
    name = [p for p in parts if p.startswith("encoder")][-1]       # encoder2_1, encoder6_2, ...
    table = {"encoder1_1": 1, "encoder1_2": 1, "encoder2_1": 1, "encoder2_2": 2, "encoder2_3": 2, "encoder3_1": 2,
             "encoder3_2": 4, "encoder3_3": 4, "encoder4_1": 4, "encoder4_2": 8, "encoder4_3": 8,
             "encoder6_1": 8, "encoder6_2": 16, "encoder6_3": 16}
    return init_radius * table[name]


def seeded_state_dict(reference_sd, seed, init_radius=4.25 * 0.3):
    """New state dict with the keys/shapes/dtypes of ``reference_sd`` and seeded values."""
    out = {}
    for k in sorted(reference_sd.keys()):
        v = reference_sd[k]
        r = kpconv_radius_of(k, init_radius) if k.endswith("kernel_points") else None
        out[k] = seeded_tensor(k, v.shape, v.dtype, seed, radius=r)
    return out


def load_snapshot(model, path, strict=False):
    """Reference checkpoint reader (base_tester.py:111-122): ``{'epoch','iteration','model'}`` tar."""
    state = torch.load(path, map_location="cpu", weights_only=True)
    sd = state["model"] if "model" in state else state
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)
