"""Stack-mode data pipeline on the GPU — experiments/lcrnet/data.py:10-74 (`precompute_data_stack_mode`) and the
single-scan / pair collates (:77-127, :350-406) without the DataLoader worker processes: the whole batch is prepared by
HIP kernels on the stream that then runs the encoder.

Two entry points:
  * precompute_data_stack_mode(...)  — the reference signature and return layout (int64 indices, exact shapes), built from
    the drop-in ops; every op synchronises once to size its output (like the reference, it is a blocking call).
  * precompute_batch(...)            — the throughput path: capacity buffers + device-side lengths, the 4 support grids
    shared by the 10 searches (7 if the decoder-only upsampling lists are skipped), int32 indices, ONE host sync at the end.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib
from .modules.ops import SupportGrid, finish_deferred, grid_subsample, grid_subsample_device, radius_search, radius_search_deferred

STATUS_KEY_OVERFLOW = 1
STATUS_LEN_MISMATCH = 2


def precompute_data_stack_mode(points, lengths, num_stages, voxel_size, radius, neighbor_limits):
    """Drop-in for data.py:10-74 on device tensors (same loop structure: voxel and radius double per stage)."""
    assert num_stages == len(neighbor_limits)
    points_list, lengths_list, neighbors_list, subsampling_list, upsampling_list = [], [], [], [], []
    lengths = lengths.to(points.device)
    for i in range(num_stages):
        if i > 0:
            points, lengths = grid_subsample(points, lengths, voxel_size=voxel_size)
            points = points.contiguous()
        points_list.append(points)
        lengths_list.append(lengths)
        voxel_size *= 2
    pending, slots = [], []                                  # the ten searches are issued back to back, read back ONCE
    for i in range(num_stages):
        cur_p, cur_l = points_list[i], lengths_list[i]
        pending.append(radius_search_deferred(cur_p, cur_p, cur_l, cur_l, radius, neighbor_limits[i]))
        slots.append(neighbors_list)
        if i < num_stages - 1:
            sub_p, sub_l = points_list[i + 1], lengths_list[i + 1]
            pending.append(radius_search_deferred(sub_p, cur_p, sub_l, cur_l, radius, neighbor_limits[i]))
            slots.append(subsampling_list)
            pending.append(radius_search_deferred(cur_p, sub_p, cur_l, sub_l, radius * 2, neighbor_limits[i + 1]))
            slots.append(upsampling_list)
        radius *= 2
    for lst, idx in zip(slots, finish_deferred(pending)):
        lst.append(idx)
    return {"points": points_list, "lengths": lengths_list, "neighbors": neighbors_list,
            "subsampling": subsampling_list, "upsampling": upsampling_list}


MAX_STAGES = 8


class PrecomputeLayout(ctypes.Structure):
    """Mirror of LcrPrecomputeLayout (include/lcr_hip.h)."""
    _fields_ = [("num_stages", ctypes.c_int), ("B", ctypes.c_int), ("upsampling", ctypes.c_int), ("n_raw", ctypes.c_int64),
                ("limits", ctypes.c_int * MAX_STAGES), ("cap", ctypes.c_int64 * MAX_STAGES),
                ("off_points", ctypes.c_size_t * MAX_STAGES), ("off_lengths", ctypes.c_size_t * MAX_STAGES),
                ("off_order", ctypes.c_size_t * MAX_STAGES), ("off_neighbors", ctypes.c_size_t * MAX_STAGES),
                ("off_subsampling", ctypes.c_size_t * MAX_STAGES), ("off_upsampling", ctypes.c_size_t * MAX_STAGES),
                ("out_bytes", ctypes.c_size_t), ("ws_bytes", ctypes.c_size_t)]


class PrecomputedArena:
    """Result of one native pre-processing call before any tensor view exists: the output arena, its layout and the host copy
    of the per-stage lengths.  `views()` builds the data dictionary.  The split lets a producer thread hand the arena over
    with almost no interpreter work (the ~25 tensor views cost more host time than issuing the 140 launches)."""
    __slots__ = ("out", "lay", "lengths_host", "points", "lengths", "raw", "upsampling", "B", "S")

    def __init__(self, out, lay, lengths_host, points, lengths, raw, upsampling):
        self.out, self.lay, self.lengths_host, self.points, self.lengths = out, lay, lengths_host, points, lengths
        self.raw, self.upsampling, self.B, self.S = raw, upsampling, lay.B, lay.num_stages

    def views(self):
        out, lay, S, B = self.out, self.lay, self.S, self.B
        tot = [sum(l) for l in self.lengths_host]

        def view(off, rows, cols, dtype, esize):
            return out[off:off + rows * cols * esize].view(dtype).view(rows, cols)

        first = 0 if self.raw else 1
        pts = ([] if self.raw else [self.points]) + [view(lay.off_points[i], tot[i], 3, torch.float32, 4) for i in range(first, S)]
        lens = ([] if self.raw else [self.lengths]) + [out[lay.off_lengths[i]:lay.off_lengths[i] + 8 * B].view(torch.int64) for i in range(first, S)]
        orders = [out[lay.off_order[i]:lay.off_order[i] + 4 * tot[i]].view(torch.int32) for i in range(S)]
        neighbors = [view(lay.off_neighbors[i], tot[i], lay.limits[i], torch.int32, 4) for i in range(S)]
        subsampling = [view(lay.off_subsampling[i], tot[i + 1], lay.limits[i], torch.int32, 4) for i in range(S - 1)]
        # upsampling == "nearest" (layout flag 2): one column per row — all the decoder reads
        upsamp = [view(lay.off_upsampling[i], tot[i], 1 if self.upsampling == 2 else lay.limits[i + 1], torch.int32, 4) for i in range(S - 1)] if self.upsampling else []
        # lists_valid_first: every row came out of a radius search (valid entries first, padding behind) — the native encoder driver may
        # then stop reading a row at its first padded chunk (LCR_ENC_LISTS_VALID_FIRST); a hand-built dictionary without the key gets full scans
        return {"order": orders, "points": pts, "lengths": lens, "neighbors": neighbors, "subsampling": subsampling, "upsampling": upsamp,
                "lengths_host": self.lengths_host, "segment_lengths": lens, "lists_valid_first": True}


def _ups_flag(upsampling):
    """True / 1: the reference collate's full upsampling rows; "nearest" / 2: column 0 only (what KPDecoder reads); False / 0: none."""
    return 2 if upsampling in ("nearest", 2) else int(bool(upsampling))


_layout_cache = {}
_ws_cache = threading.local()


def _layout_for(n0, B, S, neighbor_limits, upsampling, n_raw):
    key = (n0, B, S, tuple(int(x) for x in neighbor_limits), _ups_flag(upsampling), n_raw)
    lay = _layout_cache.get(key)
    if lay is None:
        lay = PrecomputeLayout()
        lim = (ctypes.c_int * S)(*key[3])
        _lib.check(_lib.lib().lcr_precompute_layout(n0, B, S, ctypes.cast(lim, ctypes.c_void_p), _ups_flag(upsampling), n_raw,
                                                    ctypes.addressof(lay)), "lcr_precompute_layout")
        if len(_layout_cache) > 64:
            _layout_cache.clear()
        _layout_cache[key] = lay
    return lay


def _workspace(nbytes, dev, stream):
    """Scratch of the native call, reused by a (thread, stream): it is dead when the call returns (the call ends with a stream
    synchronisation), so consecutive calls on the same stream can share it."""
    cache = getattr(_ws_cache, "ws", None)
    if cache is None:
        cache = _ws_cache.ws = {}
    key = (dev, stream.value)
    ws = cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = cache[key] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    return ws


def precompute_batch_arena(points, lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling=True, key_bits_hint=32,
                           raw_voxel=None, capacity=None):
    """The native pre-processing call (csrc/precompute.hip) without the tensor views: returns a PrecomputedArena."""
    _lib.require_cuda(points, lengths)
    assert points.dtype == torch.float32 and points.is_contiguous() and lengths.dtype == torch.int64 and num_stages == len(neighbor_limits)
    dev = points.device
    lengths = lengths.contiguous()
    B, S = lengths.numel(), num_stages
    raw = raw_voxel is not None
    n_raw = points.shape[0] if raw else 0
    row_floats = int(points.shape[1]) if points.dim() == 2 else 3      # raw mode takes [N, C >= 3] rows (x, y, z first; 4 = KITTI xyzi)
    if points.dim() != 2 or row_floats < 3 or (row_floats != 3 and not raw):
        raise RuntimeError("points must be [N,3] (or [N, C >= 3] raw scans with raw_voxel set)")
    n0 = (int(capacity) if capacity else max(n_raw // 4, min(n_raw, 65536))) if raw else points.shape[0]
    lay = _layout_for(n0, B, S, neighbor_limits, upsampling, n_raw)
    stream = _lib.stream_ptr(dev)
    out = torch.empty(max(lay.out_bytes, 256), dtype=torch.uint8, device=dev)
    ws = _workspace(lay.ws_bytes, dev, stream)
    lens_host = (ctypes.c_int64 * (S * B))()
    status = ctypes.c_uint32(0)
    _lib.check(_lib.lib().lcr_precompute_batch_rows(_lib.ptr(points), row_floats, _lib.ptr(lengths), ctypes.addressof(lay), float(voxel_size),
                                                    float(radius), float(raw_voxel) if raw else 0.0, int(key_bits_hint), _lib.ptr(out), out.numel(),
                                                    _lib.ptr(ws), ws.numel(), ctypes.addressof(lens_host), ctypes.addressof(status), stream),
               "lcr_precompute_batch")
    st = status.value
    if st & STATUS_KEY_OVERFLOW and key_bits_hint:
        return precompute_batch_arena(points, lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling, 0, raw_voxel, capacity)
    if raw and st & STATUS_LEN_MISMATCH and n0 < n_raw:
        return precompute_batch_arena(points, lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling, key_bits_hint,
                                      raw_voxel, n_raw)             # the voxel count exceeded the guessed capacity
    if st:
        raise RuntimeError("precompute_batch: device status 0x%x" % st)
    flat = list(lens_host)
    return PrecomputedArena(out, lay, [flat[i * B:(i + 1) * B] for i in range(S)], points, lengths, raw, _ups_flag(upsampling))


def precompute_batch_native(points, lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling=True, key_bits_hint=32,
                            raw_voxel=None, capacity=None):
    """precompute_batch as ONE native call (csrc/precompute.hip): same dictionary, int32 indices.  The ~140 launches and the
    length read-back are issued by C++ with the interpreter lock released throughout.

    raw_voxel: `points` are RAW scans, voxelised at this size inside the same call (no extra host round trip); `capacity` is
    the row capacity assumed for the voxelised stack (default: a quarter of the raw points; a too small guess is detected on
    the device and the call is repeated with the safe bound)."""
    return precompute_batch_arena(points, lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling, key_bits_hint,
                                  raw_voxel, capacity).views()


def precompute_batch(points, lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling=True,
                     index_dtype=torch.int32, key_bits_hint=32, native=True):
    """Same five lists as precompute_data_stack_mode for a stack of B clouds, plus:
         'lengths_host'    : list of python lists (per stage),
         'segment_lengths' : per-stage device int64 lengths (GroupNorm segments = clouds; regroup for pair semantics).
    points: device f32 [N,3] (exact rows), lengths: device i64 [B].  One host synchronisation."""
    assert num_stages == len(neighbor_limits)
    dev = points.device
    lengths = lengths.to(dev)
    if native and index_dtype == torch.int32:
        return precompute_batch_native(points.contiguous(), lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling,
                                       key_bits_hint)
    pts, lens, statuses = [points.contiguous()], [lengths], []
    v = voxel_size
    for i in range(1, num_stages):
        v *= 2
        p, l, st = grid_subsample_device(pts[-1], lens[-1], v, key_bits_hint=key_bits_hint)
        pts.append(p)
        lens.append(l)
        statuses.append(st)
    grids = []
    r = radius
    for i in range(num_stages):
        grids.append(SupportGrid(pts[i], lens[i], r))
        r *= 2
    orders = [g.order() for g in grids]          # spatially coherent processing order of every stage's points
    neighbors, subsampling, upsamp = [], [], []
    for i in range(num_stages):
        neighbors.append(grids[i].query(pts[i], lens[i], neighbor_limits[i], dtype=index_dtype))
        if i < num_stages - 1:
            subsampling.append(grids[i].query(pts[i + 1], lens[i + 1], neighbor_limits[i], dtype=index_dtype))
            if upsampling:
                upsamp.append(grids[i + 1].query(pts[i], lens[i], 1 if _ups_flag(upsampling) == 2 else neighbor_limits[i + 1], dtype=index_dtype))
    host = torch.stack(lens + [torch.cat([s.long() for s in statuses] + [g.status.long() for g in grids]).sum().expand(lengths.numel())]).cpu()
    status = int(host[-1][0])
    if status & STATUS_KEY_OVERFLOW and key_bits_hint:
        return precompute_batch(points, lengths, num_stages, voxel_size, radius, neighbor_limits, upsampling, index_dtype, 0, native)
    if status:
        raise RuntimeError("precompute_batch: device status 0x%x" % status)
    lengths_host = [host[i].tolist() for i in range(num_stages)]
    tot = [sum(l) for l in lengths_host]
    pts = [pts[i][:tot[i]] for i in range(num_stages)]
    neighbors = [neighbors[i][:tot[i]] for i in range(num_stages)]
    subsampling = [subsampling[i][:tot[i + 1]] for i in range(num_stages - 1)]
    upsamp = [upsamp[i][:tot[i]] for i in range(len(upsamp))]
    orders = [orders[i][:tot[i]] for i in range(num_stages)]
    return {"order": orders, "points": pts, "lengths": lens, "neighbors": neighbors, "subsampling": subsampling, "upsampling": upsamp,
            "lengths_host": lengths_host, "segment_lengths": lens, "lists_valid_first": True}


def _merge_samples(data_dicts):
    """Lists of per-sample values keyed like the samples (numpy arrays become tensors)."""
    merged = {}
    for sample in data_dicts:
        for key, value in sample.items():
            merged.setdefault(key, []).append(torch.from_numpy(value) if isinstance(value, np.ndarray) else value)
    return merged


def _finish_collate(merged, feats, points_list, batch_size, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data, device):
    lengths = torch.tensor([int(p.shape[0]) for p in points_list], dtype=torch.int64)
    points = torch.cat(points_list, dim=0)
    if batch_size == 1:                                   # the reference unwraps single-sample batches
        merged = {k: v[0] for k, v in merged.items()}
    merged["features"] = feats
    if precompute_data:
        dev = torch.device(device)
        merged["features"] = feats.to(dev)
        merged.update(precompute_data_stack_mode(points.to(dev, torch.float32).contiguous(), lengths.to(dev), num_stages, voxel_size,
                                                 search_radius, neighbor_limits))
    else:
        merged["points"], merged["lengths"] = points, lengths
    merged["batch_size"] = batch_size
    return merged


def registration_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data=True,
                                       device="cuda"):
    """Pair collate of experiments/lcrnet/data.py:77-127: clouds stacked [ref_1..ref_B, src_1..src_B], features concatenated
    the same way, every other key a per-sample list (unwrapped for B = 1), `batch_size` added.  With precompute_data the five
    lists of precompute_data_stack_mode are computed on `device` (the reference does it in DataLoader workers on the CPU) and
    `features` is moved there; without it `points` / `lengths` stay on the host like the reference's."""
    merged = _merge_samples(data_dicts)
    feats = torch.cat(merged.pop("ref_feats") + merged.pop("src_feats"), dim=0)
    points_list = merged.pop("ref_points") + merged.pop("src_points")
    return _finish_collate(merged, feats, points_list, len(data_dicts), num_stages, voxel_size, search_radius, neighbor_limits,
                           precompute_data, device)


def test_loop_detection_collate_fn_stack_mode_online(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                                     precompute_data=True, device="cuda"):
    """Single-scan collate of data.py:350-406 (loop-detection inference): `anc_points` of every sample stacked, `features` =
    the FIRST sample's `anc_feats` (as the reference does — it is only ever used with batch size 1)."""
    merged = _merge_samples(data_dicts)
    feats = merged.pop("anc_feats")[0]
    points_list = merged.pop("anc_points")
    return _finish_collate(merged, feats, points_list, len(data_dicts), num_stages, voxel_size, search_radius, neighbor_limits,
                           precompute_data, device)


test_loop_detection_collate_fn_stack_mode_online.__test__ = False     # a collate of the reference's name, not a pytest case


# ---- training-time collates (experiments/lcrnet/data.py:130-348), mirrored by name ------------------------------------------------
# Host logic only (stacking order, which keys are popped / unwrapped); the collate proper is the same device-side
# precompute_data_stack_mode.  They exist so that the reference's trainval_* scripts can keep their DataLoader code when the
# pre-processing moves to the GPU; the training loop itself (losses, optimiser) is out of scope.
def _precompute_or_keep(merged, feats, points, lengths, batch_size, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data,
                        device):
    merged["features"] = feats
    if precompute_data:
        dev = torch.device(device)
        merged["features"] = feats.to(dev)
        merged.update(precompute_data_stack_mode(points.to(dev, torch.float32).contiguous(), lengths.to(dev), num_stages, voxel_size,
                                                 search_radius, neighbor_limits))
    else:
        merged["points"], merged["lengths"] = points, lengths
    merged["batch_size"] = batch_size
    return merged


def train_loop_detection_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data=True):
    """data.py:130-170 (pre-extracted features): `features` = [pos.., anc.., neg..] feats, `lengths` likewise; nothing is
    precomputed whatever `precompute_data` says (the reference ignores it here), every other key stays a per-sample list."""
    merged = _merge_samples(data_dicts)
    feats = torch.cat(merged.pop("pos_feats") + merged.pop("anc_feats") + merged.pop("neg_feats"), dim=0)
    lengths = torch.cat(merged.pop("pos_lengths") + merged.pop("anc_lengths") + merged.pop("neg_lengths"), dim=0)
    merged["features"], merged["lengths"], merged["batch_size"] = feats, lengths, len(data_dicts)
    return merged


def train_loop_detection_collate_fn_stack_mode_online(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                                      precompute_data=True, device="cuda"):
    """data.py:173-235 (online triplets): clouds stacked [pos_1..pos_B, anc_1..anc_B, neg_1..neg_B]; when the first negative is
    empty (`lengths[2B] == 0`) the negatives' LENGTHS are dropped (their zero points add nothing to the stack); `transform` is
    unwrapped for B = 1."""
    B = len(data_dicts)
    merged = _merge_samples(data_dicts)
    lengths = torch.cat(merged.pop("pos_lengths") + merged.pop("anc_lengths") + merged.pop("neg_lengths"), dim=0)
    if lengths[2 * B] == 0:
        lengths = lengths[:2 * B]
    feats = torch.cat(merged.pop("pos_feats") + merged.pop("anc_feats") + merged.pop("neg_feats"), dim=0)
    points = torch.cat(merged.pop("pos_points") + merged.pop("anc_points") + merged.pop("neg_points"), dim=0)
    if B == 1 and "transform" in merged:
        merged["transform"] = merged["transform"][0]
    return _precompute_or_keep(merged, feats, points, lengths, B, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data, device)


def train_loop_detection_collate_fn_stack_mode_halfonline(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                                          precompute_data=True, device="cuda"):
    """data.py:238-288 (anchors online, positives / negatives pre-extracted): the stack holds the anchors only; `lengths_c` /
    `feats_c` carry the pre-extracted [pos.., neg..] coarse features."""
    B = len(data_dicts)
    merged = _merge_samples(data_dicts)
    lengths = torch.cat(merged.pop("anc_lengths"), dim=0)
    feats = torch.cat(merged.pop("anc_feats"), dim=0)
    points = torch.cat(merged.pop("anc_points"), dim=0)
    lengths_c = torch.cat(merged.pop("pos_lengths") + merged.pop("neg_lengths"), dim=0)
    feats_c = torch.cat(merged.pop("pos_feats") + merged.pop("neg_feats"), dim=0)
    out = _precompute_or_keep(merged, feats, points, lengths, B, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data, device)
    out["lengths_c"], out["feats_c"] = lengths_c, feats_c
    return out


def all_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data=True, device="cuda"):
    """data.py:290-348 (feature pre-extraction over a whole sequence): a sample flagged `pass` is handed back as it is; otherwise
    the `anc_points` of the samples are stacked, `features` = the first sample's `anc_feats`, and EVERY other key is unwrapped for
    B = 1 (not only `transform`)."""
    B = len(data_dicts)
    if data_dicts[0]["pass"] == True:   # noqa: E712  (the reference compares with ==)
        return data_dicts[0]
    merged = _merge_samples(data_dicts)
    feats = merged.pop("anc_feats")
    points_list = merged.pop("anc_points")
    lengths = torch.LongTensor([p.shape[0] for p in points_list])
    points = torch.cat(points_list, dim=0)
    if B == 1:
        merged = {k: v[0] for k, v in merged.items()}
    return _precompute_or_keep(merged, feats[0], points, lengths, B, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data, device)


def voxelize_raw_scans(points, lengths, voxel_size, key_bits_hint=32):
    """Raw-scan ingest (replaces the offline Open3D voxel_down_sample(0.3) of data/Kitti/downsample_pcd.py:29
    with the a-1 kernel): stacked raw scans -> stacked voxel barycentres; returns (points, lengths_dev, lengths_host)."""
    out, out_len, status = grid_subsample_device(points, lengths, voxel_size, key_bits_hint=key_bits_hint)
    host = torch.cat([out_len, status.long()]).cpu()
    st = int(host[-1])
    if st & STATUS_KEY_OVERFLOW and key_bits_hint:
        return voxelize_raw_scans(points, lengths, voxel_size, 0)
    if st:
        raise RuntimeError("voxelize_raw_scans: device status 0x%x" % st)
    lh = host[:-1].tolist()
    return out[:sum(lh)], out_len, lh


def calibrate_neighbors_stack_mode(clouds, num_stages, voxel_size, search_radius, keep_ratio=0.8, sample_threshold=2000):
    """Neighbour-limit calibration of data.py:408-433 from device-side counts: `clouds` is an iterable of dataset items, each either
    an f32[N,3] device tensor (a single cloud) or a `(points, lengths)` pair (a stack of clouds, e.g. a registration pair
    [ref, src] — the reference feeds dataset items through the collate with limit = hist_n, and the histogram is checked against
    `sample_threshold` after every ITEM).  Only per-row in-radius counts are needed: count-only searches."""
    import numpy as np
    from .modules.ops import grid_subsample, radius_count
    hist_n = int(np.ceil(4 / 3 * np.pi * (search_radius / voxel_size + 1) ** 3))
    hists = np.zeros((num_stages, hist_n), dtype=np.int64)
    for item in clouds:
        if isinstance(item, (tuple, list)):
            pts, lens = item[0], item[1].to(item[0].device)
        else:
            pts, lens = item, torch.tensor([item.shape[0]], dtype=torch.int64, device=item.device)
        v, r = voxel_size, search_radius
        for i in range(num_stages):
            if i > 0:
                pts, lens = grid_subsample(pts, lens, v)
                pts = pts.contiguous()
            v *= 2
            c = radius_count(pts, pts, lens, lens, r).clamp(max=hist_n - 1).cpu().numpy()   # width is capped at hist_n there
            hists[i] += np.bincount(c, minlength=hist_n)[:hist_n]
            r *= 2
        if hists.sum(1).min() > sample_threshold:
            break
    cum = np.cumsum(hists.T, axis=0)
    return np.sum(cum < (keep_ratio * cum[hist_n - 1, :]), axis=0)
