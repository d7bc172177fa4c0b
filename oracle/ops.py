"""ORACLE — test infrastructure only.

ctypes front-end to the C++ restatement (``liblcr_oracle.so``) and to the compiled reference
(``_ref/libref_ops.so``, optional).  Signatures mirror the reference's Python operator API:
  grid_subsample  : /root/reference/experiments/lcrnet/modules/ops/grid_subsample.py:7-22
  radius_search   : /root/reference/experiments/lcrnet/modules/ops/radius_search.py:7-27
  precompute_data_stack_mode : /root/reference/experiments/lcrnet/data.py:10-74
All inputs/outputs are numpy arrays (float32 [N,3], int64 [B]).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "_build", "liblcr_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref_ops.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(ref: bool = True) -> None:
    """Compile the restatement (always) and the reference checker (only where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/utils/extensions"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        L = ctypes.CDLL(_ORACLE_SO)
        L.oracle_grid_subsample.restype = ctypes.c_int64
        L.oracle_grid_subsample.argtypes = [_f32p, _i64p, ctypes.c_int, ctypes.c_float, _f32p, _i64p]
        L.oracle_radius_count.restype = ctypes.c_int64
        L.oracle_radius_count.argtypes = [_f32p, _f32p, _i64p, _i64p, ctypes.c_int, ctypes.c_float, _i32p]
        L.oracle_radius_search.restype = ctypes.c_int
        L.oracle_radius_search.argtypes = [_f32p, _f32p, _i64p, _i64p, ctypes.c_int, ctypes.c_float,
                                           ctypes.c_int64, _i64p, _i32p]
        L.oracle_hashmap_order.restype = ctypes.c_int64
        L.oracle_hashmap_order.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64, _i64p]
        _lib = L
    return _lib


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = ctypes.CDLL(_REF_SO)
        L.ref_grid_subsample.restype = ctypes.c_int64
        L.ref_grid_subsample.argtypes = [_f32p, _i64p, ctypes.c_int, ctypes.c_float, _f32p, _i64p]
        L.ref_radius_neighbors.restype = ctypes.c_int64
        L.ref_radius_neighbors.argtypes = [_f32p, _f32p, _i64p, _i64p, ctypes.c_int, ctypes.c_float, _i64p,
                                           ctypes.c_int64]
        _ref = L
    return _ref


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def grid_subsample(points, lengths, voxel_size, impl="oracle"):
    points, pp = _f32(points)
    lengths, lp = _i64(lengths)
    out = np.empty((max(points.shape[0], 1), 3), dtype=np.float32)
    out_len = np.empty(lengths.shape[0], dtype=np.int64)
    fn = lib().oracle_grid_subsample if impl == "oracle" else ref().ref_grid_subsample
    m = fn(pp, lp, len(lengths), ctypes.c_float(voxel_size), out.ctypes.data_as(_f32p), out_len.ctypes.data_as(_i64p))
    return out[:m].copy(), out_len


def hashmap_order(keys):
    """order[j] = insertion rank of the j-th element in std::unordered_map iteration order (distinct keys)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    order = np.empty(keys.shape[0], dtype=np.int64)
    n = lib().oracle_hashmap_order(keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), keys.shape[0],
                                   order.ctypes.data_as(_i64p))
    assert n == keys.shape[0], "keys must be distinct"
    return order


def radius_count(q_points, s_points, q_lengths, s_lengths, radius):
    q, qp = _f32(q_points)
    s, sp = _f32(s_points)
    ql, qlp = _i64(q_lengths)
    sl, slp = _i64(s_lengths)
    counts = np.empty(q.shape[0], dtype=np.int32)
    mx = lib().oracle_radius_count(qp, sp, qlp, slp, len(ql), ctypes.c_float(radius), counts.ctypes.data_as(_i32p))
    return counts, int(mx)


def radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, impl="oracle", return_counts=False, ref_width=False):
    """int64 [Nq, k]: k = neighbor_limit if > 0 else the max in-radius count.  ref_width=True: the reference's shape for
    neighbor_limit > 0 too, k = min(neighbor_limit, max count) (radius_search.py:25-26 slices a matrix only as wide as the densest
    ball); the default keeps neighbor_limit columns (the extra ones are pure padding) like the native fixed-width lists."""
    q, qp = _f32(q_points)
    s, sp = _f32(s_points)
    ql, qlp = _i64(q_lengths)
    sl, slp = _i64(s_lengths)
    nq = q.shape[0]
    if impl == "ref":
        w = ref().ref_radius_neighbors(qp, sp, qlp, slp, len(ql), ctypes.c_float(radius), None, 0)
        out = np.empty((nq, w), dtype=np.int64)
        r = ref().ref_radius_neighbors(qp, sp, qlp, slp, len(ql), ctypes.c_float(radius), out.ctypes.data_as(_i64p), w)
        assert r == w
        if neighbor_limit > 0:
            out = np.ascontiguousarray(out[:, :neighbor_limit])
        return out
    counts = np.empty(nq, dtype=np.int32)
    if neighbor_limit > 0:
        width = int(neighbor_limit)
    else:
        # the grid search fills counts too; do a throw-away pass of width 0 to learn the max count
        lib().oracle_radius_search(qp, sp, qlp, slp, len(ql), ctypes.c_float(radius), 0, None, counts.ctypes.data_as(_i32p))
        width = int(counts.max()) if nq else 0
    out = np.empty((nq, width), dtype=np.int64)
    lib().oracle_radius_search(qp, sp, qlp, slp, len(ql), ctypes.c_float(radius), width,
                               out.ctypes.data_as(_i64p), counts.ctypes.data_as(_i32p))
    if ref_width and nq and int(counts.max()) < width:
        out = np.ascontiguousarray(out[:, :int(counts.max())])
    return (out, counts) if return_counts else out


def precompute_data_stack_mode(points, lengths, num_stages, voxel_size, radius, neighbor_limits, impl="oracle"):
    """Restates data.py:10-74 (3 subsamples with voxel doubling from 2*voxel... see note, 4+3+3 radius searches)."""
    assert num_stages == len(neighbor_limits)
    points_list, lengths_list, neighbors_list, subsampling_list, upsampling_list = [], [], [], [], []
    points = np.ascontiguousarray(points, dtype=np.float32)
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    for i in range(num_stages):
        if i > 0:
            points, lengths = grid_subsample(points, lengths, voxel_size, impl=impl)
        points_list.append(points)
        lengths_list.append(lengths)
        voxel_size *= 2
    for i in range(num_stages):
        cur_p, cur_l = points_list[i], lengths_list[i]
        neighbors_list.append(radius_search(cur_p, cur_p, cur_l, cur_l, radius, neighbor_limits[i], impl=impl))
        if i < num_stages - 1:
            sub_p, sub_l = points_list[i + 1], lengths_list[i + 1]
            subsampling_list.append(radius_search(sub_p, cur_p, sub_l, cur_l, radius, neighbor_limits[i], impl=impl))
            upsampling_list.append(radius_search(cur_p, sub_p, cur_l, sub_l, radius * 2, neighbor_limits[i + 1], impl=impl))
        radius *= 2
    return {"points": points_list, "lengths": lengths_list, "neighbors": neighbors_list,
            "subsampling": subsampling_list, "upsampling": upsampling_list}
