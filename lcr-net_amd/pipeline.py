"""Scan-parallel descriptor pipeline: raw scans -> 256-D descriptors, batch after batch, on two HIP streams.

The reference hides its CPU preprocessing behind DataLoader worker processes (utils/utils/torch.py:48-77, `num_workers=8`) while
the GPU runs the model.  The GPU-native equivalent: the preprocessing of batch k+1 (voxelise, subsamples, grids, radius
searches — many short, latency-bound launches and two tiny D2H length reads) runs on stream A while the encoder + NetVLAD of
batch k (throughput-bound MFMA/HBM kernels) run on stream B.  Ordering is by events; tensors produced on A and consumed on B
are `record_stream`ed so the caching allocator cannot recycle them early.
"""
import os
import threading
import weakref
import time

import torch

from .data import precompute_batch, precompute_batch_arena, voxelize_raw_scans


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        import ctypes
        _HIP = ctypes.CDLL("libamdhip64.so")
    return _HIP


def destroy_streams(streams):
    """hipStreamDestroy for streams made by `_native_streams` that never carried tensors (probe candidates that lost): torch's
    ExternalStream wrapper does not own its handle.  Streams that did carry work go through `release_streams` instead."""
    import ctypes
    for st in streams:
        h = getattr(st, "_lcr_handle", None)
        if h:
            st.synchronize()
            _hip().hipStreamDestroy(ctypes.c_void_p(h))
            st._lcr_handle = None


# Streams that carried a pipeline's work are never destroyed: torch's caching allocator remembers the streams a block was used on
# (`record_stream`) and records an event on each of them when the block is finally freed — possibly long after `close()`, e.g.
# while a traceback keeps a generator frame alive — and an event record on a destroyed stream poisons the context ("operation not
# permitted when stream is capturing" at the next unrelated call).  `close()` parks them here; the next pipeline on the device takes
# them back, so the number of native streams is bounded by the pipelines alive at the same time.
_PARKED = {}
_PARK_LOCK = threading.Lock()


def release_streams(streams):
    """Hand worked-on native streams back for reuse (idempotent per stream)."""
    for st in streams:
        if getattr(st, "_lcr_handle", None) and not getattr(st, "_lcr_parked", False):
            st.synchronize()
            st._lcr_parked = st._lcr_used = True
            with _PARK_LOCK:
                _PARKED.setdefault((st.device.index, st._lcr_priority), []).append(st)


def _native_streams(device, n, priority=0):
    """n HIP streams wrapped for torch: parked ones of the same device / priority first, then new ones created back to back (handle
    kept in `_lcr_handle` for `destroy_streams`)."""
    import ctypes
    hip = _hip()
    out = []
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    with _PARK_LOCK:
        parked = _PARKED.get((idx, int(priority)), [])
        while parked and len(out) < n:
            st = parked.pop()
            st._lcr_parked = False
            out.append(st)
    with torch.cuda.device(device):
        for _ in range(n - len(out)):
            h = ctypes.c_void_p()
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, int(priority))      # hipStreamNonBlocking
            if rc != 0:
                raise RuntimeError("hipStreamCreateWithPriority failed: %d" % rc)
            st = torch.cuda.ExternalStream(h.value, device=device)
            st._lcr_handle, st._lcr_priority, st._lcr_parked, st._lcr_used = h.value, int(priority), False, False
            out.append(st)
    return out


def _share_queue(a, b, spin_us=400):
    """True if streams a and b are served by the same hardware queue: a short kernel on b behind a long spin on a."""
    import ctypes
    from . import _lib
    L = _lib.lib()
    best = None
    for _ in range(3):                                                     # the shortest of three trials: a cold launch, a clock
        a.synchronize()                                                    # ramp or another process's kernel can only ADD time
        b.synchronize()
        t0, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(b)                                                       # b's clock starts before a's spin is queued
        _lib.check(L.lcr_stream_spin(spin_us, ctypes.c_void_p(a.cuda_stream)), "lcr_stream_spin")
        _lib.check(L.lcr_stream_spin(1, ctypes.c_void_p(b.cuda_stream)), "lcr_stream_spin")
        tb.record(b)
        a.synchronize()
        b.synchronize()
        us = t0.elapsed_time(tb) * 1e3
        best = us if best is None else min(best, us)
        if best < 0.5 * spin_us:
            return False
    return True


def distinct_queue_streams(device, n, priority=0, pool=10):
    """n streams on n DIFFERENT hardware queues (as far as the runtime has them; 4 by default).

    The runtime deals streams to its hardware queues by an internal schedule that depends on every stream created in the process
    before (torch's pool, RCCL, ...), and two BUSY streams on one queue serialise each other: with the pipeline's four streams on
    four queues 2.3 k scans/s, with both encoder streams on one queue 1.9 k, with each encoder stream behind a pre-processing chain
    1.8 k (rocprofv3 queue ids; shifting the creation order by one dummy stream was enough to fall from one case into the other).
    So a small pool is created and probed pairwise (`_share_queue`, ~1 ms per probe) and the first n mutually non-sharing streams
    are kept; if fewer exist, the pool order fills the rest."""
    if os.environ.get("LCR_NO_QUEUE_PROBE"):
        return _native_streams(device, n, priority)
    cand = _native_streams(device, max(pool, n), priority)
    import ctypes
    from . import _lib
    for c in cand:                                        # first launch on every candidate (code object load, queue creation)
        _lib.check(_lib.lib().lcr_stream_spin(1, ctypes.c_void_p(c.cuda_stream)), "lcr_stream_spin")
    torch.cuda.synchronize(device)
    chosen = []
    for c in cand:
        if len(chosen) == n:
            break
        if all(not _share_queue(x, c) for x in chosen):
            chosen.append(c)
    for c in cand:
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    losers = [c for c in cand if not any(c is x for x in chosen)]
    destroy_streams([c for c in losers if not getattr(c, "_lcr_used", False)])     # fresh candidates that lost the probe are not kept alive
    release_streams([c for c in losers if getattr(c, "_lcr_used", False)])           # parked ones taken for the probe go back
    return chosen


class HostIngest:
    """Host -> device leg of the pipeline (the reference: `_load_point_cloud(...)[:, :3]` in a DataLoader worker,
    dataset_overlap_online.py:245-253, then `to_cuda(data_dict)`, utils/engine/single_tester.py:59).

    A batch arrives as HOST tensors — points f32 [sum N, C] with C = 3 or 4 (a KITTI velodyne scan is [N,4]: x, y, z, intensity; it is
    uploaded unsliced and the voxel-key kernels step over the fourth column) and lengths i64 [B].  `upload` stages it in pinned memory
    (skipped when the caller's tensor is pinned already), copies it to a device slot on a dedicated COPY stream (SDMA: no CU time) and
    returns device views + the event the consumer's stream has to wait for.  `slots` batches can be in flight: a slot is taken by
    `upload` and handed back by `release` once the pre-processing call that read it has returned (raw mode: stage-0 points are produced
    inside the call, nothing references the raw rows afterwards)."""

    def __init__(self, device, slots):
        import queue
        self.device = device
        self.stream = torch.cuda.Stream(device)
        self.free = queue.Queue()
        for i in range(slots):
            self.free.put(i)
        self.pin = [None] * slots            # pinned staging (points as a flat f32 buffer, lengths)
        self.dev = [None] * slots
        self.pin_len = [None] * slots
        self.dev_len = [None] * slots
        self.src = [None] * slots            # the caller's pinned tensor while its copy may be in flight (kept alive until the slot is released)
        self.uploaded_bytes = 0

    def _room(self, slot, numel, B):
        if self.pin[slot] is None or self.pin[slot].numel() < numel:
            cap = int(numel * 1.25) + 1024
            self.pin[slot] = torch.empty(cap, dtype=torch.float32).pin_memory()
            self.dev[slot] = torch.empty(cap, dtype=torch.float32, device=self.device)
        if self.pin_len[slot] is None or self.pin_len[slot].numel() < B:
            self.pin_len[slot] = torch.empty(max(B, 64), dtype=torch.int64).pin_memory()
            self.dev_len[slot] = torch.empty(max(B, 64), dtype=torch.int64, device=self.device)

    def upload(self, points, lengths, stop=None):
        """-> (points_dev [N,C], lengths_dev [B], ready event, slot) or None if `stop` was set while waiting for a slot."""
        import queue
        if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] < 3 or not points.is_contiguous():
            raise RuntimeError("host scans must be a contiguous float32 [N, C >= 3] tensor")
        while True:
            try:
                slot = self.free.get(timeout=0.05)
                break
            except queue.Empty:
                if stop is not None and stop.is_set():
                    return None
        n, c, B = points.shape[0], points.shape[1], lengths.numel()
        self._room(slot, n * c, B)
        src = points.reshape(-1)
        if not points.is_pinned():
            # pageable -> pinned: a plain memcpy in this thread through numpy (torch's CPU copy_ fans a 15 MB copy out over the whole
            # intra-op pool — 256 threads on the GPU box — and took 12 ms per batch next to the pipeline's busy host threads)
            self.pin[slot].numpy()[:n * c] = src.numpy()
            src = self.pin[slot][:n * c]
        else:
            # pinned input: copied from where it lies.  CONTRACT: the caller does not modify or re-use that tensor until the descriptors
            # of its batch have been yielded (a loader that recycles one pinned buffer per batch must hand over pageable tensors or
            # keep `depth + workers + 1` buffers in rotation)
            self.src[slot] = points
        self.pin_len[slot][:B].copy_(lengths.to(torch.int64).reshape(-1))
        with torch.cuda.stream(self.stream):
            d = self.dev[slot][:n * c]
            d.copy_(src, non_blocking=True)
            dl = self.dev_len[slot][:B]
            dl.copy_(self.pin_len[slot][:B], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.uploaded_bytes += 4 * n * c
        return d.view(n, c), dl, ev, slot

    def release(self, slot):
        self.src[slot] = None
        self.free.put(slot)

    def reset(self):
        """All slots free again (only with no upload in flight and no consumer left: end of a run)."""
        import queue
        while True:
            try:
                self.free.get_nowait()
            except queue.Empty:
                break
        for i in range(len(self.pin)):
            self.src[i] = None
            self.free.put(i)


class DescriptorPipeline:
    def __init__(self, model, voxel_size=0.3, radius=1.275, num_stages=4, neighbor_limits=(64, 65, 74, 80), upsampling=False,
                 raw_voxel=None, overlap=True, producer_thread=True, depth=2, pre_workers=2):
        """raw_voxel: voxel size of the raw-scan ingest step (None = inputs are already voxelised like the reference's
        downsampled .npy scans; 0.3 = BASELINE configs[1]).  upsampling: also compute the 3 decoder-only upsampling lists."""
        self.model = model
        self.voxel_size, self.radius, self.num_stages = voxel_size, radius, num_stages
        self.limits = list(neighbor_limits)
        self.upsampling, self.raw_voxel, self.overlap = upsampling, raw_voxel, overlap
        dev = next(model.parameters()).device
        self.device = dev
        # The pre-processing chain is latency-bound (~130 dependent short launches per batch).  With ONE chain in flight it needs the
        # high-priority queue so that its launches are not parked behind the encoder's long kernels (+9..27 %).  Two chains in
        # flight (consecutive batches on two host threads / streams) hide that latency better — but only at NORMAL priority: two
        # queues of the same elevated priority run pathologically slowly on this stack (one batch 3.9 ms instead of 1.8, even on an
        # otherwise idle GPU), which is what made the second worker look like a loss at first.  2 workers at normal priority:
        # 2.17 -> 2.31 k scans/s; a third (2.27) or a third encoder stream (2.13) loses again.
        self.producer_thread, self.depth, self.pre_workers = producer_thread, depth, max(1, int(pre_workers))
        default_prio = "1" if (self.pre_workers == 1 or not producer_thread) else "0"
        prio = -1 if os.environ.get("LCR_PRE_PRIORITY", default_prio) != "0" else 0
        # all streams of the pipeline are created ONCE, here: W pre-processing + 2 encoder streams on distinct hardware queues
        if not overlap:
            self._streams = []
        elif prio == 0:
            self._streams = distinct_queue_streams(dev, self.pre_workers + 2, 0)
        else:                                             # elevated pre-processing streams live in their own queue class
            self._streams = _native_streams(dev, self.pre_workers, prio) + distinct_queue_streams(dev, 2, 0)
        self.pre_stream = self._streams[0] if overlap else None
        self.pre_streams = self._streams[:self.pre_workers] if (overlap and producer_thread) else []
        # a pipeline that is dropped without close() still hands its native streams back (the wrapper does not own the handle)
        self._finalizer = weakref.finalize(self, release_streams, list(self._streams))
        self._finalizer.atexit = False               # nothing to hand back to at interpreter exit (and the runtime may be gone)
        self.stats = {"pre_wait_s": 0.0, "pre_busy_s": 0.0, "enc_wait_s": 0.0, "batches": 0}   # where the two host threads wait
        self._ones_buf = None
        self.enc_streams = None      # set by enable_dual_encoder(): consecutive batches' encoders on alternating streams

    def close(self):
        """Give the pipeline's native streams back (idempotent; see `release_streams`).  The pipeline cannot run afterwards
        (`run` raises)."""
        release_streams(self._streams)
        self._streams, self.pre_streams, self.pre_stream, self.enc_streams = [], [], None, None
        self._closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def enable_dual_encoder(self, n=2):
        """Run the encoders of consecutive batches on two alternating streams so that the small-grid kernels of one (stage-3/4
        GEMMs, GroupNorm, NetVLAD) fill the gaps of the other.  Results are handed back in order on the caller's stream."""
        own = self._streams[self.pre_workers:self.pre_workers + n]
        self.enc_streams = own + [torch.cuda.Stream(self.device) for _ in range(n - len(own))]
        return self

    # ---- stages -----------------------------------------------------------------------------------------------------
    def preprocess_arena(self, points, lengths):
        """The device work of `preprocess` with (almost) no interpreter work: returns a PrecomputedArena (raw one-call mode) or
        the finished dictionary (two-call / pre-voxelised modes).  `finish` turns either into the encoder's input."""
        if self.raw_voxel is not None and not os.environ.get("LCR_PRE_TWO_CALLS"):
            # raw scans -> everything, one native call and ONE host round trip (the voxelisation's own read-back is gone)
            return precompute_batch_arena(points.contiguous(), lengths.to(points.device), self.num_stages, self.voxel_size, self.radius,
                                          self.limits, upsampling=self.upsampling, raw_voxel=self.raw_voxel)
        if self.raw_voxel is not None:
            points, lengths, _ = voxelize_raw_scans(points, lengths, self.raw_voxel)
        return precompute_batch(points.contiguous(), lengths, self.num_stages, self.voxel_size, self.radius, self.limits,
                                upsampling=self.upsampling)

    def _ones(self, n):
        if self._ones_buf is None or self._ones_buf.shape[0] < n:
            self._ones_buf = torch.ones(max(n, 1 << 18), 1, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()      # rare: read from other streams afterwards
        return self._ones_buf[:n]

    def finish(self, arena):
        dd = arena if isinstance(arena, dict) else arena.views()
        dd["features"] = self._ones(dd["points"][0].shape[0])
        dd["lengths_c_host"] = dd["lengths_host"][-1]
        return dd

    def preprocess(self, points, lengths):
        """points f32[N,3] (stacked raw or voxelised scans), lengths i64[B] -> data dict (on the current stream)."""
        return self.finish(self.preprocess_arena(points, lengths))

    def encode(self, dd):
        with torch.no_grad():
            return self.model(dd)["anc_global"]

    # ---- driver -----------------------------------------------------------------------------------------------------
    def run(self, batches, sync_to_caller=True):
        """batches: iterable of (points, lengths) device tensors.  Yields one [B,256] descriptor tensor per batch, in order.
        sync_to_caller=True: the yielded tensor is valid on the CURRENT stream (the consumer may use it without further
        synchronisation).  sync_to_caller=False (threaded two-encoder mode): yields (descriptors, done_event, stream) and never
        enqueues a wait on the caller's stream — a consumer that keeps working on the producing stream (e.g. an all-gather issued
        under `torch.cuda.stream(stream)`) then leaves the caller's hardware queue free of barrier packets: that queue is shared with
        one of the pipeline's four streams (5 streams, 4 hardware queues), and a wait-for-the-encoder parked in it stalls whatever
        else runs there."""
        if getattr(self, "_closed", False):
            raise RuntimeError("DescriptorPipeline is closed")      # its streams were given back: threaded mode would start no producer and wait forever
        if not self.overlap:
            for pts, lens in batches:
                if not pts.is_cuda:                      # host batch, no pipelining: plain upload on the current stream
                    pts, lens = pts.to(self.device), lens.to(self.device)
                yield self.encode(self.preprocess(pts, lens))
            return
        if self.producer_thread:
            yield from self._run_threaded(batches, sync_to_caller)
            return
        main = torch.cuda.current_stream(self.device)
        pre = self.pre_stream
        pending = None            # (data dict, ready event) of the batch whose encoder has not been launched yet
        first = True
        for pts, lens in batches:
            if not pts.is_cuda:
                pts, lens = pts.to(self.device), lens.to(self.device)
            if pending is not None:
                dd, ready = pending
                main.wait_event(ready)
                desc = self.encode(dd)                  # enqueued on `main`, runs while the next batch is pre-processed
            if first:
                pre.wait_stream(main)                   # inputs were produced on `main`; later batches must be resident already
                first = False
            with torch.cuda.stream(pre):
                ndd = self.preprocess(pts, lens)        # host blocks only on `pre` (length read-backs)
                ready = torch.cuda.Event()
                ready.record(pre)
            for v in ndd.values():
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(main)
            if pending is not None:
                yield desc
            pending = (ndd, ready)
        if pending is not None:
            dd, ready = pending
            main.wait_event(ready)
            yield self.encode(dd)

    def _run_threaded(self, batches, sync_to_caller=True):
        """Same overlap, but the pre-processing (whose two length read-backs block the host) runs in its own host thread(s),
        `depth` batches ahead, so the encoder stream never waits for the host to come back from a synchronisation.  ctypes
        releases the GIL during every kernel launch / synchronisation, so the threads do interleave.  With `pre_workers` > 1,
        consecutive batches are pre-processed concurrently on separate streams: the pre-processing of ONE batch is a chain of
        ~400 dependent short launches and two host round trips, i.e. latency- not throughput-bound."""
        import queue
        import threading
        main = torch.cuda.current_stream(self.device)
        dev = self.device
        W = max(1, int(self.pre_workers))
        streams = self.pre_streams[:W]
        for st in streams:
            st.wait_stream(main)
        out = queue.Queue()
        slots = threading.Semaphore(self.depth + W - 1)      # batches pre-processed but not yet consumed
        stop = threading.Event()                             # set when the consumer leaves (exhaustion, break, exception)
        batches = iter(batches)
        first = next(batches, None)
        feeder, ingest = None, None
        if first is not None and not first[0].is_cuda:
            # HOST batches: a feeder thread uploads them on the copy stream, up to `depth + W` batches ahead of the pre-processing
            # workers, which then only wait for an event on their own stream
            if self.raw_voxel is None and first[0].shape[1] != 3:
                raise RuntimeError("host scans with more than 3 columns need raw_voxel (the raw-scan ingest produces the [n,3] stage-0 points)")
            ingest = self._ingest = getattr(self, "_ingest", None) or HostIngest(dev, self.depth + W + 1)
            fed = queue.Queue()

            def feed():
                try:
                    torch.cuda.set_device(dev)
                    item = first
                    while item is not None and not stop.is_set():
                        up = ingest.upload(item[0], item[1], stop)
                        if up is None:
                            break
                        fed.put(up)
                        item = next(batches, None)
                    fed.put(None)
                except BaseException as e:                   # surfaces in a worker, from there in the consumer
                    fed.put(e)

            def fed_items():
                while True:
                    up = fed.get()
                    if up is None:
                        fed.put(None)                        # every worker sees the end marker
                        return
                    if isinstance(up, BaseException):
                        fed.put(up)
                        raise up
                    yield up

            feeder = threading.Thread(target=feed, daemon=True)
            feeder.start()
            it = enumerate(fed_items())
        else:
            import itertools
            it = enumerate(itertools.chain([first] if first is not None else [], batches))
        it_lock = threading.Lock()

        def producer(st):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(st):
                    while True:
                        t0 = time.perf_counter()
                        slots.acquire()
                        t1 = time.perf_counter()
                        if stop.is_set():
                            break
                        with it_lock:
                            nxt = next(it, None)
                        if nxt is None:
                            slots.release()
                            break
                        if ingest is not None:
                            k, (pts, lens, up_ev, up_slot) = nxt
                            st.wait_event(up_ev)              # the H2D copy of this batch (copy stream)
                            try:
                                if self.raw_voxel is None:
                                    # pre-voxelised host batches: stage 0 of the data dictionary IS the input (points[0], lengths[0],
                                    # segment_lengths[0] are the tensors handed in), so it must not live in the ring slot — the feeder
                                    # overwrites the slot with batch k + slots as soon as it is released, long before the encoder of
                                    # batch k has run.  Raw mode rebuilds stage 0 inside the arena and needs no copy.
                                    pts, lens = pts.clone(), lens.clone()
                                dd = self.preprocess_arena(pts, lens)     # returns after synchronising `st`: the slot is dead
                            finally:
                                ingest.release(up_slot)
                        else:
                            k, (pts, lens) = nxt
                            dd = self.preprocess_arena(pts, lens)
                        self.stats["pre_wait_s"] += t1 - t0           # blocked: the encoder side is behind
                        self.stats["pre_busy_s"] += time.perf_counter() - t1
                        self.stats["batches"] += 1
                        ev = torch.cuda.Event()
                        ev.record(st)
                        out.put((k, dd, ev))
                out.put(None)
            except BaseException as e:   # surface errors in the consumer
                out.put(e)

        threads = [threading.Thread(target=producer, args=(st,), daemon=True) for st in streams]
        for th in threads:
            th.start()
        ready = {}           # out-of-order arrivals (several workers)
        try:
            yield from self._consume(out, slots, ready, W, main, sync_to_caller)
        finally:
            # consumer gone (normally or not): wake every producer blocked in slots.acquire(), let them see the flag, and join.
            # Arenas still queued are dropped after their producing streams have drained.
            stop.set()
            for _ in range(W + self.depth + 1):
                slots.release()
            for th in threads:
                th.join()
            if feeder is not None:
                feeder.join()
                ingest.stream.synchronize()
                ingest.reset()                               # batches uploaded but never consumed (early exit) give their slots back
            for st in streams:
                st.synchronize()

    def _consume(self, out, slots, ready, W, main, sync_to_caller):
        k, finished, pending = 0, 0, None
        while True:
            while k not in ready and finished < W:
                t0 = time.perf_counter()
                item = out.get()
                self.stats["enc_wait_s"] += time.perf_counter() - t0   # blocked: the pre-processing side is behind
                if item is None:
                    finished += 1
                elif isinstance(item, BaseException):
                    raise item
                else:
                    ready[item[0]] = item[1:]
            if k not in ready:
                break
            arena, ev = ready.pop(k)
            slots.release()
            es = main if self.enc_streams is None else self.enc_streams[k % len(self.enc_streams)]
            if isinstance(arena, dict):
                for v in arena.values():
                    for t in (v if isinstance(v, (list, tuple)) else [v]):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(es)
            else:
                arena.out.record_stream(es)                 # every view shares this one allocation
                if not arena.raw:
                    # pre-voxelised mode: stage 0 (points[0] / lengths[0]) is NOT a view of `out` but the tensor handed in — for host
                    # batches a clone made on the producer's stream.  The encoder reads it on `es`, asynchronously: without this the
                    # block returns to the producer stream's pool when the consumer drops `dd` and a later batch's clone can
                    # overwrite it under the running encoder (advisor r5).
                    for t in (arena.points, arena.lengths):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(es)
            dd = self.finish(arena)                         # tensor views are built here, off the pre-processing thread
            es.wait_event(ev)
            if self.enc_streams is None:
                yield self.encode(dd)
            else:
                if k < len(self.enc_streams):
                    es.wait_stream(main)                    # weights / inputs prepared on the caller's stream
                with torch.cuda.stream(es):
                    desc = self.encode(dd)
                    done = torch.cuda.Event()
                    done.record(es)
                if not sync_to_caller:
                    yield (desc, done, es)
                else:
                    desc.record_stream(main)
                    if pending is not None:
                        main.wait_event(pending[1])
                        yield pending[0]
                    pending = (desc, done)
            k += 1
        if pending is not None:
            main.wait_event(pending[1])
            yield pending[0]


def gpu_node_cpus(device_index):
    """The host cores of the NUMA node GPU `device_index` hangs off (sysfs, via the device's PCI address), restricted to the cores this
    process may use; None when the topology cannot be read (containers without sysfs, single-node hosts)."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % addr).read())
        if node < 0:
            return None
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        allowed = set(os.sched_getaffinity(0))
        cpus = [c for c in cpus if c in allowed]
        return cpus or None
    except Exception:                                  # noqa: BLE001
        return None


def compact_core_set(device_index, n):
    """n neighbouring cores next to the GPU for a pipeline's host threads, or None (leave the threads where the scheduler puts them).
    Why: the pair path is host-bound (a thread issues ~1 000 launches per pair and hands the interpreter lock to its peers at every
    blocking call); with the threads free to roam a 256-core box, 16 pairs per call ran at 690 +- 40 pairs/s, pinned to 8 cores of the
    GPU's node at 746 +- 10 (one pair per call 228 -> 257; profiles/r06_pair_core_binding.log).  The scans/s pipeline is GPU-bound and
    does not care (2 878-2 915 scans/s bound or not)."""
    if os.environ.get("LCR_PIPE_BIND", "1") == "0":
        return None
    cpus = gpu_node_cpus(device_index)
    if cpus is None or len(cpus) <= n:
        return None                                    # nothing to compact (already bound to a small share, e.g. one rank of eight)
    return cpus[:n]


class PairPipeline:
    """Registration pairs (BASELINE config 5, reference loop: experiments/inference/infer_registration.py) through the full pair
    model, `workers` pairs in flight: one host thread + one HIP stream per worker, results handed back in input order.

    One pair is ~1000 short, dependent launches and four host round trips (NMS sizes, match counts, ...), i.e. bound by the host
    and by launch latency, not by the GPU; a second pair in flight fills the gaps (80 -> 100-120 pairs/s on the demo pair).  At ONE pair
    per call more than two workers lose to interpreter-lock contention (59 pairs/s with three, round 2); with pairs batched per call
    (pairs_per_call >= 8: most of a call is native launch sequences that release the lock) a third worker fills the latency-bound tail:
    584 / 628 / 629 pairs/s with 2 / 3 / 4 workers at 16 pairs per call (round 5); round 6, with a third less kernel time per call: 643 / 675-690 /
    695-721 with 2 / 3 / 4.  With the worker threads pinned to 8 cores next to the GPU (`compact_core_set`, the default) more calls in flight
    pay at every batching: one pair per call 260 / 344 / 400 / 425 pairs/s with 2 / 3 / 4 / 8 workers (10: collapse — more threads than cores),
    16 per call 733 / 747 / 777 with 3 / 4 / 5."""

    def __init__(self, model, voxel_size=0.3, radius=1.275, num_stages=4, neighbor_limits=(74, 68, 70, 67), workers=None, pairs_per_call=1,
                 upsampling="nearest"):
        """pairs_per_call > 1: consecutive pairs are stacked and go through `LCRNet.forward_pairs` together (one collate, one encoder /
        transformer / vote-encoder / decoder pass for all of them; only the matching + registration tail runs pair by pair).
        upsampling: "nearest" (default) builds the three decoder-only upsampling lists as ONE column per row — KPDecoder reads column 0 only
        (nearest_upsample, backbone4.py:355-367), a limit-1 search takes the arg-min path of the kernel and writes 1/70th of the rows; True
        builds the reference collate's full rows (identical model outputs: tests/test_pairs_batched_gpu.py)."""
        self.upsampling = upsampling
        self.pairs_per_call = max(1, min(32, int(pairs_per_call)))
        if workers is None:                                  # the measured best with pinned worker threads (profiles/r06_pair_core_binding.log)
            workers = 4 if self.pairs_per_call == 1 else 5
        self.model, self.workers = model, max(1, int(workers))
        self.voxel_size, self.radius, self.num_stages, self.limits = voxel_size, radius, num_stages, list(neighbor_limits)
        self.device = next(model.parameters()).device
        # worker streams: created and probed ONCE per pipeline (two busy streams on one hardware queue serialise each other)
        self._streams = distinct_queue_streams(self.device, self.workers) if self.workers > 1 else []
        # the worker threads pin THEMSELVES to a few neighbouring cores next to the GPU (the caller's thread is left alone); LCR_PIPE_BIND=0: off
        self._cores = compact_core_set(self.device.index if self.device.index is not None else torch.cuda.current_device(), 8)
        self._finalizer = weakref.finalize(self, release_streams, list(self._streams))
        self._finalizer.atexit = False               # nothing to hand back to at interpreter exit (and the runtime may be gone)

    def close(self):
        """Give the worker streams back (idempotent; see `release_streams`)."""
        release_streams(self._streams)
        self._streams = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def one(self, points, lengths):
        """points f32[N,3] = the two clouds of a pair stacked (already voxelised), lengths i64[2] -> the model's output dict."""
        dd = precompute_batch(points.contiguous(), lengths, self.num_stages, self.voxel_size, self.radius, self.limits, upsampling=self.upsampling)
        del dd["segment_lengths"]            # pair semantics of the reference: GroupNorm statistics over BOTH clouds
        dd["features"] = torch.ones(points.shape[0], 1, device=points.device)
        dd["lengths_c_host"] = dd["lengths_host"][-1]
        with torch.no_grad():
            return self.model(dd)

    def many(self, pairs):
        """pairs: list of (points, lengths[2]) -> list of output dicts (one stacked call; `one` for a single pair)."""
        if len(pairs) == 1:
            return [self.one(*pairs[0])]
        points = torch.cat([p for p, _ in pairs]).contiguous()
        lengths = torch.cat([l for _, l in pairs])
        dd = precompute_batch(points, lengths, self.num_stages, self.voxel_size, self.radius, self.limits, upsampling=self.upsampling)
        del dd["segment_lengths"]            # forward_pairs takes GroupNorm statistics per PAIR (both clouds), not per cloud
        dd["features"] = torch.ones(points.shape[0], 1, device=points.device)
        dd["lengths_c_host"] = dd["lengths_host"][-1]
        with torch.no_grad():
            return self.model.forward_pairs(dd)

    def _grouped(self, pairs):
        group = []
        for item in pairs:
            group.append(item)
            if len(group) == self.pairs_per_call:
                yield group
                group = []
        if group:
            yield group

    def run(self, pairs):
        """pairs: iterable of (points, lengths) device tensors.  Yields one output dict per pair, in order, valid on the caller's
        current stream."""
        import queue
        import threading
        groups = self._grouped(pairs)        # the work items: groups of `pairs_per_call` pairs; a result is a list of dicts
        if self.workers == 1:
            for g in groups:
                yield from self.many(g)
            return

        dev = self.device
        main = torch.cuda.current_stream(dev)
        it = enumerate(groups)
        it_lock = threading.Lock()
        out = queue.Queue()
        slots = threading.Semaphore(2 * self.workers)       # finished groups not yet consumed

        if len(self._streams) < self.workers:
            raise RuntimeError("PairPipeline is closed")
        worker_streams = self._streams[:self.workers]
        stream_it = iter(worker_streams)
        stop = threading.Event()

        def worker():
            try:
                torch.cuda.set_device(dev)
                if self._cores:
                    try:
                        os.sched_setaffinity(0, self._cores)        # pid 0 = the calling THREAD on Linux
                    except OSError:
                        pass
                with it_lock:
                    st = next(stream_it)
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    while True:
                        slots.acquire()
                        if stop.is_set():
                            break
                        with it_lock:
                            nxt = next(it, None)
                        if nxt is None:
                            slots.release()
                            break
                        k, group = nxt
                        res = self.many(group)
                        ev = torch.cuda.Event()
                        ev.record(st)
                        out.put((k, res, ev))
                out.put(None)
            except BaseException as e:
                out.put(e)

        threads = [threading.Thread(target=worker, daemon=True) for _ in range(self.workers)]
        for th in threads:
            th.start()
        try:
            ready, finished, k = {}, 0, 0
            while True:
                while k not in ready and finished < self.workers:
                    item = out.get()
                    if item is None:
                        finished += 1
                    elif isinstance(item, BaseException):
                        raise item
                    else:
                        ready[item[0]] = item[1:]
                if k not in ready:
                    break
                res, ev = ready.pop(k)
                slots.release()
                main.wait_event(ev)
                for d in res:
                    for v in d.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(main)
                yield from res
                k += 1
        finally:
            stop.set()                                   # consumer gone: wake the workers, let them see the flag, join
            for _ in range(3 * self.workers + 1):
                slots.release()
            for th in threads:
                th.join()
            for st in worker_streams:
                st.synchronize()
