"""CPU: the N>1 path of the retrieval (frame sharding + all-gather of descriptors + per-rank search) on gloo, world_size 2,
with the oracle as the per-rank search function: the union of the ranks' rows must equal the single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_topk(queries, q0, database, k, exclude):
    from oracle import torch_ref
    C = database.shape[0]
    # the oracle searches frames [start, stop): emulate "queries = frames q0..q0+Q-1"
    qs, idx, d2 = torch_ref.retrieval_topk(database, k=k, exclude=exclude, start=q0, stop=q0 + queries.shape[0])
    return idx.int(), d2.float()


def _worker(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lcrnet_amd.retrieval import distributed_retrieval, shard_range
    g = torch.Generator().manual_seed(0)
    desc = torch.nn.functional.normalize(torch.randn(n_frames, 256, generator=g), dim=1)
    lo, hi = shard_range(n_frames, world, rank)
    qs, idx, d2 = distributed_retrieval(desc[lo:hi].clone(), n_frames, k=10, exclude=100, start=101, topk_fn=_oracle_topk)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), qs=qs.numpy(), idx=idx.numpy(), d2=d2.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [701, 350])
def test_sharded_retrieval_equals_single_process(tmp_path, n_frames):
    from oracle import torch_ref
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    qs = np.concatenate([p["qs"] for p in parts])
    idx = np.concatenate([p["idx"] for p in parts])
    g = torch.Generator().manual_seed(0)
    desc = torch.nn.functional.normalize(torch.randn(n_frames, 256, generator=g), dim=1)
    wq, widx, wd2 = torch_ref.retrieval_topk(desc, k=10, exclude=100, start=101)
    assert np.array_equal(qs, wq.numpy())                       # every query frame owned by exactly one rank, in order
    assert np.array_equal(idx, widx.numpy())


def test_shard_range_covers_everything():
    from lcrnet_amd.retrieval import shard_range
    for n in (1, 7, 8, 4541, 23201):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_search_range_is_contiguous_and_work_balanced():
    """Query rows are cut by cumulative causal work (row i costs i - exclude columns): contiguous blocks in rank order that tile
    [start, C - 1) exactly, each within 2 % (or one row) of 1/world of the work; the frame-range split it replaces left rank 0 of 8 idle."""
    from lcrnet_amd.retrieval import search_range
    for n in (102, 103, 150, 701, 4541, 23201):
        for w in (1, 2, 3, 4, 8):
            r = [search_range(n, w, k, 101, 100) for k in range(w)]
            assert r[0][0] == 101 and r[-1][1] == n - 1, (n, w, r)
            assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(a <= b for a, b in r)
            work = [sum(max(i - 100, 1) for i in range(a, b)) for a, b in r]
            tot = sum(work)
            if n >= 4541:
                assert max(work) <= tot / w * 1.02 + n, (n, w, work)
    old = [(max(k * 2901, 101), min((k + 1) * 2901, 23200)) for k in range(8)]          # frame ranges of 23 201 frames on 8 ranks
    old_work = [sum(i - 100 for i in range(a, b)) for a, b in old]
    assert max(old_work) / (sum(old_work) / 8) > 1.8                                       # what the balanced cut removes
