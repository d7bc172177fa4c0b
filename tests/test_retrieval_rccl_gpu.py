"""GPU: the exchange step of the sharded retrieval on RCCL itself.

A 1-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device: profiles/r02_rccl_two_ranks_one_device_refused.log),
so the N > 1 LOGIC is covered by the gloo world-2 CPU test (tests/test_retrieval_dist_cpu.py) and the 2-rank single-device dry run
of bench.py; here the same code path runs through a world-size-1 `nccl` (= RCCL) group: process-group creation on the device,
`all_gather_into_tensor` of the padded descriptor block as an RCCL kernel on the HIP stream, then the HIP top-k."""
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_all_gather_and_retrieval_over_rccl_world1():
    import torch.distributed as dist
    from lcrnet_amd.retrieval import distributed_retrieval, retrieval_topk
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        C = 1201                                               # KITTI 10's frame count
        g = torch.Generator().manual_seed(4)
        desc = torch.nn.functional.normalize(torch.randn(C, 256, generator=g), dim=1).cuda()
        qs, idx, d2 = distributed_retrieval(desc, C, k=50)
        torch.cuda.synchronize()
        widx, wd2 = retrieval_topk(desc[101:C - 1], 101, desc, 50, 100)
        assert qs[0] == 101 and qs[-1] == C - 2
        assert torch.equal(idx, widx) and torch.equal(d2, wd2)
        # the raw collective, uneven rows (padding path of all_gather_descriptors)
        from lcrnet_amd.retrieval import all_gather_descriptors
        assert torch.equal(all_gather_descriptors(desc[:777], 777), desc[:777])
    finally:
        dist.destroy_process_group()
