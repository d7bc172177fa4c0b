"""grid_subsample — drop-in for experiments/lcrnet/modules/ops/grid_subsample.py:7-22 on the GPU.

Reference: ``ext.grid_subsampling(points, lengths, voxel_size) -> (s_points, s_lengths)``
(utils/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62).  Bit-exact, including the output order.
"""
import ctypes

import torch

from ... import _lib


def grid_subsample_device(points, lengths, voxel_size, key_bits_hint=0):
    """Sync-free form: returns (out_xyz [N,3] capacity buffer, out_len i64[B] on device, status).
    points: f32 [N, C], C >= 3 with x, y, z first — a KITTI velodyne scan [N,4] (x, y, z, intensity) is consumed unsliced
    (the reference slices `[:, :3]` on the host, dataset_overlap_online.py:245); the output is always [M,3].
    key_bits_hint > 0 promises voxel-key bits + cloud-id bits <= hint (fewer radix passes); a broken promise sets
    LCR_STATUS_KEY_OVERFLOW in `status` and the caller must retry with 0."""
    _lib.require_cuda(points)
    if points.dtype != torch.float32:
        raise RuntimeError("points must be a float tensor")
    if lengths.dtype != torch.int64:
        raise RuntimeError("lengths must be an long tensor")
    if not points.is_contiguous():
        raise RuntimeError("points must be contiguous")
    if not lengths.is_contiguous():
        raise RuntimeError("lengths must be contiguous")
    dev = points.device
    lengths = lengths.to(dev, non_blocking=True)
    B, n = lengths.numel(), points.shape[0]
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.lcr_grid_subsample_ws_bytes(n, B, ctypes.byref(nbytes)), "lcr_grid_subsample_ws_bytes")
    ws = _lib.workspace(nbytes.value, dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
    out_len = torch.empty((B,), dtype=torch.int64, device=dev)
    if points.dim() != 2 or points.shape[1] < 3:
        raise RuntimeError("points must be [N, C] with C >= 3 (x, y, z first)")
    _lib.check(L.lcr_grid_subsample_rows(_lib.ptr(points), int(points.shape[1]), _lib.ptr(lengths), B, n, float(voxel_size), int(key_bits_hint),
                                         _lib.ptr(out), _lib.ptr(out_len), _lib.ptr(status), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(dev)), "lcr_grid_subsample")
    return out, out_len, status


MAX_CLOUDS = 64      # clouds per native call (GS_MAX_B in csrc/grid_subsample.hip); longer stacks are processed in groups


def grid_subsample(points, lengths, voxel_size):
    """Grid subsampling in stack mode (GPU).  Returns (s_points (M,3), s_lengths (B,)) like the reference."""
    if lengths.numel() > MAX_CLOUDS:                          # clouds are independent: groups of MAX_CLOUDS, results concatenated
        lens, parts, plens, o = lengths.tolist(), [], [], 0
        for g in range(0, len(lens), MAX_CLOUDS):
            n_g = sum(lens[g:g + MAX_CLOUDS])
            p, l = grid_subsample(points[o:o + n_g], lengths[g:g + MAX_CLOUDS].contiguous(), voxel_size)
            parts.append(p)
            plens.append(l)
            o += n_g
        if o != points.shape[0]:
            raise RuntimeError("lcr_grid_subsample: lengths do not match the point tensor")
        return torch.cat(parts), torch.cat(plens)
    out, out_len, status = grid_subsample_device(points, lengths, voxel_size)
    m = int(out_len.sum().item())   # host sync: the output shape is data dependent
    st = int(status.item())
    if st:
        raise RuntimeError("lcr_grid_subsample: device status 0x%x (voxel key overflow / length mismatch)" % st)
    return out[:m], out_len
