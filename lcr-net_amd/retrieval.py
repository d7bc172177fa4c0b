"""Descriptor retrieval for loop detection (a-9) and its scan-parallel, multi-GPU form.

Reference: experiments/loop_detection/eval_loop_detection_overlap_dataset.py:148-260 — per query frame i >= 101 a faiss
IndexIVFFlat(nlist=1) over descriptors [0, i-100), k=50, squared L2 (`eval_one_epoch` :183-214); Recall@1 via
`compute_topN` (:29-62).  Here: one masked exhaustive search on the GPU (lcr_retrieval_topk), and for N GPUs each rank owns a
contiguous range of frames, computes their descriptors, all-gathers the [n_r,256] blocks over RCCL (the only collective
on the path — it has no counterpart in the reference, which runs retrieval offline on one CPU) and searches its own rows.
"""
import ctypes

import torch

from . import _lib

K_DEFAULT, EXCLUDE_DEFAULT, START_DEFAULT = 50, 100, 101


def retrieval_topk(queries, q0, database, k=K_DEFAULT, exclude=EXCLUDE_DEFAULT):
    """queries [Q,D] = frames q0..q0+Q-1, database [C,D] = frames 0..C-1 -> (idx int32 [Q,k], d2 f32 [Q,k]) on the GPU."""
    _lib.require_cuda(queries, database)
    assert queries.dtype == torch.float32 and database.dtype == torch.float32
    queries, database = queries.contiguous(), database.contiguous()
    Q, D = queries.shape
    C = database.shape[0]
    dev = queries.device
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.lcr_retrieval_ws_bytes(Q, C, ctypes.byref(nbytes)), "lcr_retrieval_ws_bytes")
    ws = _lib.workspace(nbytes.value, dev)
    idx = torch.empty((Q, k), dtype=torch.int32, device=dev)
    d2 = torch.empty((Q, k), dtype=torch.float32, device=dev)
    _lib.check(L.lcr_retrieval_topk(_lib.ptr(queries), Q, int(q0), _lib.ptr(database), C, D, int(k), int(exclude), _lib.ptr(idx),
                                    _lib.ptr(d2), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "lcr_retrieval_topk")
    return idx, d2


def shard_range(n_frames, world, rank):
    """Contiguous frame range [lo, hi) of `rank` (keeps the i-100 temporal mask a plain global-index compare)."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def search_range(n_frames, world, rank, start=START_DEFAULT, exclude=EXCLUDE_DEFAULT):
    """Query rows [q_lo, q_hi) that `rank` SEARCHES (the descriptors stay owned by shard_range's frame ranges).  The mask j < i - exclude is
    causal: query i costs (i - exclude) columns, so with the queries split like the frames rank 0 would have almost nothing to do and the
    last rank 15/64 of the whole search at 8 ranks.  Here the contiguous query range [start, n_frames - 1) is cut where the
    cumulative work sum(i - exclude) reaches r/world of the total — still contiguous blocks in rank order (the reference's row order is the
    concatenation, and the kernel keeps its plain global-index compare), equal work per rank to within one row."""
    a = min(start, n_frames - 1)
    b = max(n_frames - 1, a)
    if b <= a:
        return (a, a)

    def cum(i):
        """work of rows [a, i): row j costs max(j - exclude, 1) columns (an arithmetic series beyond row exclude + 1)"""
        k = min(max(exclude + 1, a), i)                            # rows [a, k) cost 1 each
        flat = k - a
        return flat + ((k - exclude) + (i - 1 - exclude)) * (i - k) // 2 if i > k else flat

    tot = cum(b)
    cuts = [a]
    for r in range(1, world):                                      # exact integer bisection; every rank runs the same arithmetic
        target = tot * r // world
        lo, hi = cuts[-1], b
        while lo < hi:
            mid = (lo + hi) // 2
            if cum(mid) < target:
                lo = mid + 1
            else:
                hi = mid
        cuts.append(lo)
    cuts.append(b)
    return cuts[rank], cuts[rank + 1]


def all_gather_descriptors(local, n_frames, group=None):
    """local [n_r, D] (this rank's contiguous frame range per shard_range) -> [n_frames, D] on every rank.
    One all_gather_into_tensor of ceil(C/world)-row padded blocks (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = (n_frames + world - 1) // world
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:n_frames]


def distributed_retrieval(local_desc, n_frames, k=K_DEFAULT, exclude=EXCLUDE_DEFAULT, start=START_DEFAULT, group=None, topk_fn=None):
    """Every rank: all-gather the descriptors, search its share of the query rows (search_range: contiguous, work-balanced).
    Returns (query frame ids [Q_r], idx [Q_r,k], d2 [Q_r,k]).  topk_fn defaults to the HIP kernel; the CPU (gloo) tests pass
    the oracle so that the sharding / exchange logic is exercised without a GPU."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(n_frames, world, rank)
    assert local_desc.shape[0] == hi - lo, "local descriptors must cover exactly this rank's frame range"
    full = all_gather_descriptors(local_desc, n_frames, group)
    q_lo, q_hi = search_range(n_frames, world, rank, start, exclude)     # equal causal work per rank (not the frame ranges)
    fn = topk_fn or retrieval_topk
    if q_hi <= q_lo:
        e = torch.empty((0, k), device=local_desc.device)
        return torch.arange(0), e.int(), e.float()
    idx, d2 = fn(full[q_lo:q_hi], q_lo, full, k, exclude)
    return torch.arange(q_lo, q_hi), idx, d2
