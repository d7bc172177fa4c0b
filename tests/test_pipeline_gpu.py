"""GPU: the two-stream DescriptorPipeline yields the same descriptors as the single-stream path, batch after batch."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_scan

pytestmark = pytest.mark.gpu


def test_overlapped_pipeline_matches_sequential():
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.pipeline import DescriptorPipeline
    from lcrnet_amd.weights import seeded_state_dict
    golden = np.load(os.path.join(GOLDEN, "model_golden.npz"))
    seed = json.load(open(os.path.join(GOLDEN, "model_manifest.json")))["seed"]
    m = create_model().eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed))
    m = m.cuda()
    names = [["003854", "000958"], ["004481"], ["000026", "000560", "003528"], ["003854"]]
    batches = []
    for grp in names:
        scans = [load_scan(n) for n in grp]
        batches.append((torch.from_numpy(np.concatenate(scans)).cuda(),
                        torch.tensor([len(s) for s in scans], dtype=torch.int64, device="cuda")))
    limits = [74, 68, 70, 67]
    seq = [d.clone() for d in DescriptorPipeline(m, neighbor_limits=limits, overlap=False).run(batches)]
    for rep in range(4):                                 # repeat: stream races would show up as run-to-run differences
        pipe = DescriptorPipeline(m, neighbor_limits=limits, overlap=True, producer_thread=rep % 2 == 0)
        if rep == 2:
            pipe.enable_dual_encoder()                   # encoders of consecutive batches on alternating streams
        ovl = [d.clone() for d in pipe.run(batches * (3 if rep == 2 else 1))][:len(batches)]
        torch.cuda.synchronize()
        assert len(ovl) == len(seq)
        for a, b in zip(seq, ovl):
            assert a.shape == b.shape and (a - b).abs().max().item() < 1e-6
    for grp, d in zip(names, seq):
        for i, n in enumerate(grp):
            want = torch.from_numpy(golden[f"{n}/anc_global"])[0]
            assert (d[i].cpu() - want).abs().max().item() < 1e-4


def test_native_precompute_matches_python_path():
    """lcr_precompute_batch (one native call, fork-join side streams) returns exactly the lists of the op-by-op Python path."""
    from lcrnet_amd.data import precompute_batch
    scans = [load_scan(n) for n in ["003854", "000958", "004481"]]
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device="cuda")
    limits = [74, 68, 70, 67]
    for ups in (True, False):
        for rep in range(3):                             # repeat: a missing stream dependency would show up as differences
            a = precompute_batch(pts, lens, 4, 0.3, 1.275, limits, upsampling=ups, native=False)
            b = precompute_batch(pts, lens, 4, 0.3, 1.275, limits, upsampling=ups, native=True)
            torch.cuda.synchronize()
            assert a["lengths_host"] == b["lengths_host"]
            for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
                assert len(a[key]) == len(b[key])
                for x, y in zip(a[key], b[key]):
                    assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y), key
            for x, y in zip(a["order"], b["order"]):      # cell-sorted order: same cells, order inside a cell is free
                assert x.shape == y.shape and torch.equal(torch.sort(x.long())[0], torch.sort(y.long())[0])


@pytest.mark.parametrize("sizes", [[2000], [37, 5000, 1], [300, 0, 4000]])
def test_native_precompute_ragged_batches(sizes):
    """Single cloud, tiny clouds and an EMPTY cloud inside the batch: native call == op-by-op path."""
    from lcrnet_amd.data import precompute_batch
    rng = np.random.default_rng(5)
    scan = load_scan("003854")
    clouds = [scan[rng.choice(len(scan), n, replace=False)] if n else np.zeros((0, 3), np.float32) for n in sizes]
    pts = torch.from_numpy(np.concatenate(clouds).astype(np.float32)).cuda()
    lens = torch.tensor(sizes, dtype=torch.int64, device="cuda")
    limits = [20, 20, 20, 20]
    a = precompute_batch(pts, lens, 4, 0.3, 1.275, limits, native=False)
    b = precompute_batch(pts, lens, 4, 0.3, 1.275, limits, native=True)
    torch.cuda.synchronize()
    assert a["lengths_host"] == b["lengths_host"]
    for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
        for x, y in zip(a[key], b[key]):
            assert x.shape == y.shape and torch.equal(x, y), key


@pytest.mark.parametrize("stages", [5, 6])
def test_native_precompute_more_searches_than_one_multi_launch(stages):
    """num_stages 5 / 6 with upsampling = 13 / 16 searches, more than one lcr_radius_query_multi launch takes
    (LCR_RADIUS_QUERY_MULTI_MAX = 12): the native call issues the list in slices and returns the op-by-op lists."""
    from lcrnet_amd.data import precompute_batch
    scans = [load_scan(n) for n in ["003854", "000958"]]
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device="cuda")
    limits = [30] * stages
    a = precompute_batch(pts, lens, stages, 0.3, 1.275, limits, upsampling=True, native=False)
    b = precompute_batch(pts, lens, stages, 0.3, 1.275, limits, upsampling=True, native=True)
    torch.cuda.synchronize()
    assert a["lengths_host"] == b["lengths_host"] and len(b["upsampling"]) == stages - 1
    for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
        assert len(a[key]) == len(b[key])
        for x, y in zip(a[key], b[key]):
            assert x.shape == y.shape and torch.equal(x, y), key


def test_raw_scan_mode_of_the_native_precompute():
    """Raw scans -> stage 0 -> everything in ONE native call == voxelize_raw_scans followed by precompute_batch; a capacity guess
    that is too small is detected on the device and the call repeats with the safe bound."""
    import lcrnet_amd.synthetic as synthetic
    from lcrnet_amd.data import precompute_batch, precompute_batch_native, voxelize_raw_scans
    scans = [synthetic.synthetic_scan(i)[::3] for i in range(3)]
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device="cuda")
    limits = [40, 40, 40, 40]
    p0, l0, _ = voxelize_raw_scans(pts, lens, 0.3)
    a = precompute_batch(p0.contiguous(), l0, 4, 0.3, 1.275, limits, native=False)
    for cap in (None, 100):                               # default guess; a guess far too small (forces the retry)
        b = precompute_batch_native(pts, lens, 4, 0.3, 1.275, limits, raw_voxel=0.3, capacity=cap)
        torch.cuda.synchronize()
        assert a["lengths_host"] == b["lengths_host"]
        for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
            assert len(a[key]) == len(b[key])
            for x, y in zip(a[key], b[key]):
                assert x.shape == y.shape and torch.equal(x, y), key


def test_pair_pipeline_two_in_flight_equals_sequential():
    """Registration pairs with two pairs in flight (two host threads / streams): same outputs, same order, as one at a time."""
    import os
    from conftest import load_scan
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.pipeline import PairPipeline
    from lcrnet_amd.weights import seeded_state_dict
    dev = torch.device("cuda", 0)
    cfg = make_cfg()
    cfg["neighbor_limits"] = [74, 68, 70, 67]
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
    m = m.to(dev)
    a, b = load_scan("003854"), load_scan("000958")
    stacks = []
    for k in range(5):                       # five different pairs: the demo pair, swapped, and cropped variants
        x, y = (a, b) if k % 2 == 0 else (b, a)
        x, y = x[: len(x) - 700 * k], y[: len(y) - 500 * k]
        stacks.append((torch.from_numpy(np.concatenate([x, y])).to(dev), torch.tensor([len(x), len(y)], dtype=torch.int64, device=dev)))
    seq = list(PairPipeline(m, workers=1).run(stacks))
    par = list(PairPipeline(m, workers=2).run(stacks))
    torch.cuda.synchronize()
    assert len(seq) == len(par) == 5
    for s, p in zip(seq, par):
        assert s["length"].tolist() == p["length"].tolist()
        assert torch.equal(s["pos_node_corr_indices"], p["pos_node_corr_indices"])
        assert torch.allclose(s["estimated_transform"], p["estimated_transform"], atol=1e-5)


def test_pipeline_fuzz_ragged_batches_are_bit_identical_to_sequential():
    """Raw-scan pipeline (two pre-processing workers, two encoder streams) over batches of 1..10 clouds of random sizes — thinned
    scans, clouds of a few points — three times over: every descriptor tensor equals, bit for bit, the one of the same batch
    pre-processed and encoded alone on one stream (arena recycling, stream hand-offs and the per-scan GroupNorm segments)."""
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.pipeline import DescriptorPipeline
    from lcrnet_amd.weights import seeded_state_dict
    m = create_model().eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 99))
    m = m.cuda()
    base = [load_scan(n) for n in ["003854", "000958", "004481", "000026", "000560", "003528"]]
    rng = np.random.default_rng(5)
    batches = []
    for k in range(14):
        clouds = []
        for i in range(int(rng.integers(1, 11))):
            s = base[int(rng.integers(0, len(base)))]
            c = s[rng.random(len(s)) < rng.uniform(0.05, 1.0)]
            if rng.random() < 0.15:
                c = c[:int(rng.integers(1, 50))]
            clouds.append(np.ascontiguousarray(c, dtype=np.float32))
        batches.append((torch.from_numpy(np.concatenate(clouds)).cuda(), torch.tensor([len(c) for c in clouds], dtype=torch.int64, device="cuda")))
    limits = [74, 68, 70, 67]
    with DescriptorPipeline(m, neighbor_limits=limits, upsampling=True, raw_voxel=0.3) as pipe:
        seq = []
        for p, l in batches:
            seq.append(pipe.encode(pipe.preprocess(p, l)).cpu())
            torch.cuda.synchronize()
        assert all(torch.isfinite(d).all() and d.shape == (int(l.numel()), 256) for d, (_, l) in zip(seq, batches))
        pipe.enable_dual_encoder(2)
        for rep in range(3):
            got = [d.cpu() for d in pipe.run(batches)]
            torch.cuda.synchronize()
            assert len(got) == len(seq)
            for a, b in zip(seq, got):
                assert a.shape == b.shape and torch.equal(a, b)


def _ingest_model():
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.weights import seeded_state_dict
    m = create_model().eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
    return m.cuda()


def test_strided_rows_subsample_equals_sliced_rows():
    """a-1 on KITTI velodyne rows f32[N,4] (x, y, z, intensity) consumed unsliced (lcr_grid_subsample_rows) == the same op on the
    host-sliced [:, :3] copy (dataset_overlap_online.py:245) == the oracle, bit for bit, order included; 3 ragged clouds incl. an empty one."""
    from oracle import ops as oracle_ops
    from lcrnet_amd.modules.ops import grid_subsample
    rng = np.random.default_rng(3)
    clouds = [load_scan("003854"), np.zeros((0, 3), np.float32), load_scan("000958")[:7001]]
    xyz = np.concatenate(clouds).astype(np.float32)
    for cols in (4, 7):
        rows = np.concatenate([xyz, rng.standard_normal((len(xyz), cols - 3)).astype(np.float32) * 1e6], axis=1)   # junk in the unused columns
        lens = np.array([len(c) for c in clouds], dtype=np.int64)
        for voxel in (0.3, 0.6):
            p4, l4 = grid_subsample(torch.from_numpy(rows).cuda(), torch.from_numpy(lens).cuda(), voxel)
            p3, l3 = grid_subsample(torch.from_numpy(xyz).cuda(), torch.from_numpy(lens).cuda(), voxel)
            wp, wl = oracle_ops.grid_subsample(xyz, lens, voxel)
            assert p4.shape[1] == 3 and l4.tolist() == l3.tolist() == wl.tolist()
            assert torch.equal(p4, p3) and np.array_equal(p4.cpu().numpy().view(np.uint32), wp.view(np.uint32))
    with pytest.raises(RuntimeError):
        grid_subsample(torch.zeros(10, 2).cuda(), torch.tensor([10]).cuda(), 0.3)


def test_host_xyzi_batches_give_the_resident_descriptors_bit_for_bit():
    """Ingest leg: host batches f32[N,4] (pinned AND pageable) uploaded on the copy stream inside the threaded two-encoder pipeline
    -> the descriptors of the resident f32[N,3] batches, bit for bit; every slot of the upload ring is reused several times."""
    from lcrnet_amd.pipeline import DescriptorPipeline
    m = _ingest_model()
    rng = np.random.default_rng(11)
    names = [["003854", "000958"], ["004481"], ["000026", "000560", "003528"], ["003854"], ["000958", "004481"]]
    dev_batches, host_batches = [], []
    for k, grp in enumerate(names * 3):                   # 15 batches through a ring of depth + W + 1 = 5 slots
        scans = [load_scan(n) + np.float32(0.01 * (k // len(names))) for n in grp]       # repeats of a group are different clouds
        xyz = np.concatenate(scans).astype(np.float32)
        lens = torch.tensor([len(s) for s in scans], dtype=torch.int64)
        rows = torch.from_numpy(np.concatenate([xyz, rng.random((len(xyz), 1), dtype=np.float32)], axis=1))
        dev_batches.append((torch.from_numpy(xyz).cuda(), lens.cuda()))
        host_batches.append((rows.pin_memory() if k % 2 == 0 else rows, lens))
    limits = [74, 68, 70, 67]
    with DescriptorPipeline(m, neighbor_limits=limits, raw_voxel=0.3, overlap=True) as pipe:
        pipe.enable_dual_encoder()
        want = [d.clone() for d in pipe.run(dev_batches)]
        for rep in range(2):
            got = [d.clone() for d in pipe.run(host_batches)]
            torch.cuda.synchronize()
            assert len(got) == len(want)
            for a, b in zip(want, got):
                assert a.shape == b.shape and torch.equal(a, b)
        assert pipe._ingest.uploaded_bytes == 2 * sum(r.numel() * 4 for r, _ in host_batches)
        # a consumer that leaves early must not leave the feeder / workers blocked on ring slots
        it = pipe.run(host_batches)
        first = next(it).clone()
        it.close()
        assert torch.equal(first, want[0])
        again = [d.clone() for d in pipe.run(host_batches[:3])]
        assert all(torch.equal(a, b) for a, b in zip(again, want[:3]))
    # un-pipelined path takes host batches too
    seq = [d.clone() for d in DescriptorPipeline(m, neighbor_limits=limits, raw_voxel=0.3, overlap=False).run(host_batches[:2])]
    for a, b in zip(seq, want[:2]):
        assert (a - b).abs().max().item() < 1e-6
    # four columns without the raw-scan step: stage-0 points must be [n,3]
    with pytest.raises(RuntimeError):
        list(DescriptorPipeline(m, neighbor_limits=limits, overlap=True).run(host_batches[:1]))


def test_prevoxelised_host_batches_do_not_alias_the_upload_ring():
    """Host batches of PRE-VOXELISED [n,3] points (raw_voxel=None): stage 0 of the data dictionary is the input itself, so the pipeline
    must copy it out of the upload ring before the slot is handed back — with more batches than ring slots the feeder otherwise
    overwrites points[0] / lengths[0] of batch k with batch k + slots before the encoder of batch k has run (advisor, round 4).
    Batches of different scans and sizes, twice round the ring, against the resident path bit for bit."""
    from lcrnet_amd.data import voxelize_raw_scans
    from lcrnet_amd.pipeline import DescriptorPipeline
    m = _ingest_model()
    names = [["003854", "000958"], ["004481"], ["000026", "000560", "003528"], ["003854"], ["000958", "004481"], ["000560"], ["003528", "000026"]]
    dev_batches, host_batches = [], []
    for k, grp in enumerate(names * 3):                   # 21 batches through a ring of depth + W + 1 slots
        scans = [load_scan(n) + np.float32(0.01 * (k // len(names))) for n in grp]
        xyz = torch.from_numpy(np.concatenate(scans).astype(np.float32)).cuda()
        lens = torch.tensor([len(s) for s in scans], dtype=torch.int64).cuda()
        p, l, lh = voxelize_raw_scans(xyz, lens, 0.3)
        p = p[:sum(lh)].contiguous()
        dev_batches.append((p, l))
        hp = p.cpu()
        host_batches.append((hp.pin_memory() if k % 2 == 0 else hp, l.cpu()))
    limits = [74, 68, 70, 67]
    with DescriptorPipeline(m, neighbor_limits=limits, raw_voxel=None, overlap=True) as pipe:
        pipe.enable_dual_encoder()
        want = [d.clone() for d in pipe.run(dev_batches)]
        for rep in range(2):
            got = [d.clone() for d in pipe.run(host_batches)]
            torch.cuda.synchronize()
            assert len(got) == len(want)
            for k, (a, b) in enumerate(zip(want, got)):
                assert a.shape == b.shape and torch.equal(a, b), f"batch {k} (pass {rep})"


def test_weights_reloaded_through_cpu_do_not_meet_stale_derived_tensors():
    """model.cpu() -> load_state_dict(other seed) -> model.cuda(): every buffer is a NEW tensor with version 0, and the caching
    allocator readily puts it at the address of the one it replaces — the derived-tensor caches (host copy of the kernel points,
    transposed KPConv weights, the native encoder's table) must not take that for "unchanged" (found by tools/fuzz_float_parity_gpu.py:
    one case of eight ran with the previous seed's kernel points).  Same model object, five seeds in a row, each against a fresh model."""
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.modules.kpconv.kpconv import KPConv
    from lcrnet_amd.pipeline import DescriptorPipeline
    from lcrnet_amd.weights import seeded_state_dict
    limits = [74, 68, 70, 67]
    scan = load_scan("004481")[::2].copy()
    batch = [(torch.from_numpy(scan).cuda(), torch.tensor([len(scan)], dtype=torch.int64, device="cuda"))]

    def describe(model):
        with DescriptorPipeline(model, neighbor_limits=limits, overlap=False) as pipe:
            return [d.clone() for d in pipe.run(batch)][0]

    m = create_model().eval().cuda()
    for seed in (11, 12, 13, 14, 15):
        m = m.cpu()
        m.load_state_dict(seeded_state_dict(m.state_dict(), seed))
        m = m.cuda()
        for mod in m.modules():
            if isinstance(mod, KPConv):
                assert np.array_equal(mod.kernel_points_host(), mod.kernel_points.cpu().numpy())
        fresh = create_model().eval()
        fresh.load_state_dict(seeded_state_dict(fresh.state_dict(), seed))
        assert torch.equal(describe(m), describe(fresh.cuda()))
