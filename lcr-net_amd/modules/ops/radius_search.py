"""radius_search — drop-in for experiments/lcrnet/modules/ops/radius_search.py:7-27 on the GPU.

Reference: ``ext.radius_neighbors(q, s, q_lengths, s_lengths, radius)`` then ``[:, :neighbor_limit]`` (a
non-contiguous view; here the result is always contiguous).  Checks mirror the TORCH_CHECKs of
utils/extensions/cpu/radius_neighbors/radius_neighbors.cpp:12-23 (dtype / contiguity), with "CUDA" for "CPU".
"""
import ctypes

import torch

from ... import _lib


def _check(q_points, s_points, q_lengths, s_lengths):
    _lib.require_cuda(q_points, s_points)
    for name, t in (("q_points", q_points), ("s_points", s_points)):
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be a float tensor" % name)
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % name)
    for name, t in (("q_lengths", q_lengths), ("s_lengths", s_lengths)):
        if t.dtype != torch.int64:
            raise RuntimeError("%s must be an long tensor" % name)
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % name)
    if q_lengths.numel() != s_lengths.numel():
        raise RuntimeError("q_lengths and s_lengths must have the same batch size")


MAX_CLOUDS = 64      # clouds per native call (GRID_MAX_B in csrc/radius_search.hip); longer stacks are processed in groups


def _call(q_points, s_points, q_lengths, s_lengths, radius, limit, want64, want32, want_cnt):
    dev = q_points.device
    q_lengths = q_lengths.to(dev, non_blocking=True)
    s_lengths = s_lengths.to(dev, non_blocking=True)
    B = q_lengths.numel()
    if B > MAX_CLOUDS:
        return _call_grouped(q_points, s_points, q_lengths, s_lengths, radius, limit, want64, want32, want_cnt)
    nq, ns = q_points.shape[0], s_points.shape[0]
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.lcr_radius_search_ws_bytes(nq, ns, B, ctypes.byref(nbytes)), "lcr_radius_search_ws_bytes")
    ws = _lib.workspace(nbytes.value, dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    out64 = torch.empty((nq, limit), dtype=torch.int64, device=dev) if (want64 and limit > 0) else None
    out32 = torch.empty((nq, limit), dtype=torch.int32, device=dev) if (want32 and limit > 0) else None
    cnt = torch.zeros((nq,), dtype=torch.int32, device=dev) if want_cnt else None      # rows beyond sum(q_lengths) are never written
    _lib.check(L.lcr_radius_search(_lib.ptr(q_points), _lib.ptr(s_points), _lib.ptr(q_lengths), _lib.ptr(s_lengths), B,
                                   nq, ns, float(radius), int(limit), _lib.ptr(out64), _lib.ptr(out32), _lib.ptr(cnt),
                                   _lib.ptr(status), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "lcr_radius_search")
    return out64, out32, cnt, status


def _call_grouped(q_points, s_points, q_lengths, s_lengths, radius, limit, want64, want32, want_cnt):
    """Stacks of more than MAX_CLOUDS clouds: clouds are independent, so the stack is cut into groups of MAX_CLOUDS (one host read
    of the lengths), each group searched on its own rows, and the group-local indices moved to the stack's numbering (a group's pad
    value — its own support count — becomes the stack's)."""
    ql, sl = q_lengths.tolist(), s_lengths.tolist()
    ns_total = s_points.shape[0]
    outs64, outs32, cnts, status = [], [], [], torch.zeros(1, dtype=torch.int32, device=q_points.device)
    qo = so = 0
    for g in range(0, len(ql), MAX_CLOUDS):
        nq_g, ns_g = sum(ql[g:g + MAX_CLOUDS]), sum(sl[g:g + MAX_CLOUDS])
        o64, o32, c, st = _call(q_points[qo:qo + nq_g], s_points[so:so + ns_g], q_lengths[g:g + MAX_CLOUDS].contiguous(),
                                s_lengths[g:g + MAX_CLOUDS].contiguous(), radius, limit, want64, want32, want_cnt)
        for o, lst in ((o64, outs64), (o32, outs32)):
            if o is not None:
                lst.append(torch.where(o == ns_g, torch.full_like(o, ns_total), o + so))
        if c is not None:
            cnts.append(c)
        status = status | st
        qo, so = qo + nq_g, so + ns_g
    if qo != q_points.shape[0] or so != ns_total:
        status = status | 1                                   # lengths do not add up to the rows (LCR_STATUS_LEN_MISMATCH)
    cat = lambda lst: torch.cat(lst) if lst else None
    return cat(outs64), cat(outs32), cat(cnts), status


def radius_count(q_points, s_points, q_lengths, s_lengths, radius):
    """int32 [Nq] uncapped in-radius counts (all that calibrate_neighbors_stack_mode needs, data.py:423)."""
    _check(q_points, s_points, q_lengths, s_lengths)
    return _call(q_points, s_points, q_lengths, s_lengths, radius, 0, False, False, True)[2]


def radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, dtype=torch.int64, check=True):
    r"""Neighbours of ``q_points`` in ``s_points`` within ``radius`` (stack mode), on the GPU.

    check=True (the drop-in behaviour): blocking like the reference op; the device status word is read back and a lengths / rows
    mismatch raises RuntimeError instead of returning truncated lists, and the result has the reference's shape
    (N, min(neighbor_limit, max in-radius count)) — radius_search.py:25-26 slices ``[:, :limit]`` off a matrix that is only as wide
    as the densest neighbourhood.  check=False (stream-ordered callers inside the model): no host synchronisation, always
    ``neighbor_limit`` columns (the extra ones are pure padding).
    Rows are ascending in (d², index) and padded with M = s_points.shape[0]; the result is contiguous.
    """
    pending = radius_search_deferred(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, dtype)
    return finish_deferred([pending])[0] if check else pending[0]


def radius_search_deferred(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, dtype=torch.int64):
    """The launches of `radius_search` without its read-back: (indices [N, limit], counts, status word, limit).  `finish_deferred`
    turns a list of these into the reference-shaped results with ONE host synchronisation for all of them (a collate runs ten
    searches).  neighbor_limit <= 0 (full width) needs the widest neighbourhood first and therefore synchronises here; its count
    pass is reused for the result."""
    _check(q_points, s_points, q_lengths, s_lengths)
    limit = int(neighbor_limit)
    want64 = dtype == torch.int64
    if limit <= 0:
        cnt = radius_count(q_points, s_points, q_lengths, s_lengths, radius)
        limit = int(cnt.max().item()) if cnt.numel() else 0   # host sync: output width is data dependent
        if limit == 0:
            empty = torch.empty((q_points.shape[0], 0), dtype=dtype, device=q_points.device)
            return empty, cnt, torch.zeros(1, dtype=torch.int32, device=q_points.device), 0
        out64, out32, _, status = _call(q_points, s_points, q_lengths, s_lengths, radius, limit, want64, not want64, False)
        return (out64 if want64 else out32), cnt, status, limit
    out64, out32, cnt, status = _call(q_points, s_points, q_lengths, s_lengths, radius, limit, want64, not want64, True)
    return (out64 if want64 else out32), cnt, status, limit


def finish_deferred(pending):
    """[(indices, counts, status, limit), ...] -> [indices cut to (N, min(limit, widest neighbourhood))], raising on a lengths /
    rows mismatch like the reference op's TORCH_CHECKs.  One read-back for the whole list."""
    if not pending:
        return []
    dev = pending[0][0].device
    words = []
    for out, cnt, status, limit in pending:
        mx = cnt.max().to(torch.int32).reshape(1) if cnt.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
        words += [status.reshape(1), mx]
    host = torch.cat(words).tolist()
    res = []
    for k, (out, cnt, status, limit) in enumerate(pending):
        st, width = host[2 * k], host[2 * k + 1]
        if st != 0:
            raise RuntimeError("radius_search: lengths do not match the point tensors (status %d)" % st)
        res.append(out[:, :width].contiguous() if width < limit else out)
    return res


class SupportGrid:
    """A uniform grid over stacked support clouds, built once and queried by several query sets (the 10 searches of
    precompute_data_stack_mode need only 4 grids: neighbors[i], subsampling[i] and upsampling[i-1] share support and radius).
    Sync-free: lengths stay on the device, `s_points` may be a capacity buffer whose first sum(s_lengths) rows are valid."""

    def __init__(self, s_points, s_lengths, radius):
        _lib.require_cuda(s_points, s_lengths)
        assert s_points.dtype == torch.float32 and s_points.is_contiguous() and s_lengths.dtype == torch.int64
        self.s_points, self.s_lengths, self.radius = s_points, s_lengths, float(radius)
        self.B, self.ns_cap = s_lengths.numel(), s_points.shape[0]
        dev = s_points.device
        L = _lib.lib()
        nbytes = ctypes.c_size_t(0)
        _lib.check(L.lcr_support_grid_ws_bytes(self.ns_cap, self.B, ctypes.byref(nbytes)), "lcr_support_grid_ws_bytes")
        self.ws = _lib.workspace(nbytes.value, dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.lcr_support_grid_build(_lib.ptr(s_points), _lib.ptr(s_lengths), self.B, self.ns_cap, self.radius,
                                            _lib.ptr(self.status), _lib.ptr(self.ws), self.ws.numel(), _lib.stream_ptr(dev)),
                   "lcr_support_grid_build")

    def order(self):
        """int32 [ns_cap]: stacked support rows in cell-sorted order (first sum(s_lengths) entries valid)."""
        out = torch.empty((self.ns_cap,), dtype=torch.int32, device=self.s_points.device)
        _lib.check(_lib.lib().lcr_support_grid_order(_lib.ptr(self.ws), self.ns_cap, self.B, _lib.ptr(out), _lib.stream_ptr(out.device)),
                   "lcr_support_grid_order")
        return out

    def query(self, q_points, q_lengths, neighbor_limit, dtype=torch.int32, want_counts=False, q_order=None):
        """[nq_cap, limit] indices (rows beyond sum(q_lengths) are left unwritten) and optionally the in-radius counts.
        q_order (int32 [nq], a permutation of the query rows, e.g. the query set's own `order()`): processing order only —
        the result is the same, spatially coherent wavefronts re-use their candidate cells from cache."""
        assert q_points.dtype == torch.float32 and q_points.is_contiguous() and q_lengths.numel() == self.B
        dev = q_points.device
        nq = q_points.shape[0]
        limit = int(neighbor_limit)
        out = torch.empty((nq, limit), dtype=dtype, device=dev) if limit > 0 else None
        cnt = torch.empty((nq,), dtype=torch.int32, device=dev) if (want_counts or limit == 0) else None
        o64, o32 = (out, None) if dtype == torch.int64 else (None, out)
        assert q_order is None or (q_order.dtype == torch.int32 and q_order.is_contiguous() and q_order.numel() >= nq)
        _lib.check(_lib.lib().lcr_radius_query_ordered(_lib.ptr(q_points), _lib.ptr(q_lengths), self.B, nq, _lib.ptr(self.ws), self.ns_cap,
                                                       self.radius, limit, _lib.ptr(o64), _lib.ptr(o32), _lib.ptr(cnt), _lib.ptr(q_order),
                                                       _lib.stream_ptr(dev)), "lcr_radius_query")
        return (out, cnt) if (want_counts or limit == 0) else out
