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
