"""Drop-in for the reference's pybind11 extension `utils.ext` (utils/extensions/pybind.cpp:7-24): the three functions the
reference looks up by name — `grid_subsampling`, `radius_neighbors`, `radius_filter` — bound to the C ABI of include/lcr_hip.h
with ctypes only (no torch C++ API, no pybind).  Copy next to the reference's `utils/__init__.py` as `utils/ext.py`, set
LCR_HIP_LIB to the built library, and `experiments/lcrnet/modules/ops/*.py` run unchanged — on the GPU.

This file is the text of INTEGRATION.md's stub, kept importable so that tests/test_integration_stub_gpu.py exercises exactly what a
maintainer would paste.  It is NOT part of the product package (which binds the same C ABI in lcr-net_amd/_lib.py)."""
import ctypes
import os

import torch

_L = ctypes.CDLL(os.environ.get("LCR_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lcr-net_amd", "liblcr_hip.so")))
_vp, _i64, _f, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
_L.lcr_last_error.restype = ctypes.c_char_p
_L.lcr_grid_subsample_ws_bytes.argtypes = [_i64, _int, ctypes.POINTER(ctypes.c_size_t)]
_L.lcr_grid_subsample.argtypes = [_vp, _vp, _int, _i64, _f, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]
_L.lcr_radius_search_ws_bytes.argtypes = [_i64, _i64, _int, ctypes.POINTER(ctypes.c_size_t)]
_L.lcr_radius_search.argtypes = [_vp, _vp, _vp, _vp, _int, _i64, _i64, _f, _int, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]
_L.lcr_greedy_nms_ws_bytes.argtypes = [_i64, ctypes.POINTER(ctypes.c_size_t)]
_L.lcr_greedy_nms.argtypes = [_vp, _vp, _int, _i64, _f, _vp, _vp, _vp, _vp]
_p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
_stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc):
    if rc:
        raise RuntimeError((_L.lcr_last_error() or b"lcr error").decode())


def _ws(fn, *a):
    n = ctypes.c_size_t(0)
    _check(fn(*a, ctypes.byref(n)))
    return torch.empty(max(n.value, 256), dtype=torch.uint8, device="cuda")


def grid_subsampling(points, lengths, voxel_size):          # utils/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62
    points, lengths = points.cuda().contiguous(), lengths.cuda().contiguous()
    n, B = points.shape[0], lengths.numel()
    ws = _ws(_L.lcr_grid_subsample_ws_bytes, n, B)
    out, out_len = torch.empty_like(points), torch.empty_like(lengths)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    _check(_L.lcr_grid_subsample(_p(points), _p(lengths), B, n, voxel_size, _p(out), _p(out_len), _p(status), _p(ws), ws.numel(), _stream()))
    if int(status):
        raise RuntimeError("grid_subsampling: device status %d" % int(status))
    return out[: int(out_len.sum())], out_len               # same (s_points, s_lengths) as the reference, on the GPU


def radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius):   # cpu/radius_neighbors/radius_neighbors.cpp:5-68
    q, s = q_points.cuda().contiguous(), s_points.cuda().contiguous()
    ql, sl = q_lengths.cuda().contiguous(), s_lengths.cuda().contiguous()
    nq, ns, B = q.shape[0], s.shape[0], ql.numel()
    ws = _ws(_L.lcr_radius_search_ws_bytes, nq, ns, B)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(nq, dtype=torch.int32, device="cuda")
    _check(_L.lcr_radius_search(_p(q), _p(s), _p(ql), _p(sl), B, nq, ns, radius, 0, None, None, _p(cnt), _p(status), _p(ws), ws.numel(), _stream()))
    width = int(cnt.max()) if nq else 0                     # the reference's data-dependent output width (host sync, like the CPU op)
    out = torch.empty((nq, width), dtype=torch.int64, device="cuda")
    if width:
        _check(_L.lcr_radius_search(_p(q), _p(s), _p(ql), _p(sl), B, nq, ns, radius, width, _p(out), None, None, _p(status), _p(ws), ws.numel(), _stream()))
    if int(status):
        raise RuntimeError("radius_neighbors: lengths do not match the point tensors (status %d)" % int(status))
    return out


def radius_filter(nodes_dict, length_dict, radius):         # cpu/radius_filter/radius_filter.cpp:3-36 (dead code in the reference: vote.py:91)
    """Greedy radius NMS per cloud: node i is kept iff every node kept before it is farther than `radius`.  -> ([mask per cloud],
    [kept count per cloud]) like the reference's nested vectors.  Runs lcr_greedy_nms (the exact parallel replay the vote encoder uses);
    its distance is nn.PairwiseDistance's (|a - b + 1e-6|, vote.py:13-70) where this op takes the plain norm: they differ only for
    pairs within 2e-6 m of the radius."""
    pts = nodes_dict.cuda().float().contiguous()
    lens = torch.as_tensor(length_dict, dtype=torch.int64).reshape(-1).cuda().contiguous()
    n, B = pts.shape[0], lens.numel()
    ws = _ws(_L.lcr_greedy_nms_ws_bytes, n)
    keep = torch.empty(n, dtype=torch.uint8, device="cuda")
    kept = torch.empty(B, dtype=torch.int64, device="cuda")
    _check(_L.lcr_greedy_nms(_p(pts), _p(lens), B, n, radius, _p(keep), _p(kept), _p(ws), _stream()))
    masks = list(torch.split(keep.bool(), lens.tolist()))
    return masks, [k for k in kept]
