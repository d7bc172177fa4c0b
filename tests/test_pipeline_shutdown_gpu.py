"""GPU: the pipelines' worker threads and native streams are owned, shut down and released.

* a consumer that stops iterating early (break / exception) must not leave producer threads blocked in the slot semaphore holding
  arenas and streams: `run()`'s finally block wakes and joins them;
* `close()` gives the native streams back to a per-device park (idempotent; they are NOT destroyed: the caching allocator may still
  record an event on a stream a freed block was used on), after which the pipeline refuses to run; the next pipeline reuses them;
* repeated `run()` calls reuse the streams created in `__init__` (no per-call stream creation)."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.weights import seeded_state_dict
    m = create_model().eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
    return m.cuda()


def _batch():
    import lcrnet_amd.synthetic as synthetic
    scans = [synthetic.synthetic_scan(20 + i, n_azimuth=400) for i in range(2)]
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device="cuda")
    return pts, lens


def test_descriptor_pipeline_early_exit_and_close():
    from lcrnet_amd.pipeline import DescriptorPipeline
    m = _model()
    pts, lens = _batch()
    before = threading.active_count()
    pipe = DescriptorPipeline(m, 0.3, 1.275, 4, [40, 40, 40, 40], raw_voxel=0.3, pre_workers=2)
    pipe.enable_dual_encoder(2)
    streams = list(pipe._streams)
    full = [d.cpu() for d in pipe.run([(pts, lens)] * 4)]
    assert len(full) == 4 and all(torch.equal(full[0], d) for d in full)
    gen = pipe.run(((pts, lens) for _ in range(50)))
    first = next(gen).cpu()
    gen.close()                                            # consumer leaves after one batch: GeneratorExit inside run()
    assert torch.equal(first, full[0])
    assert threading.active_count() == before              # producers were woken, saw the stop flag and were joined
    with pytest.raises(ZeroDivisionError):
        for i, _ in enumerate(pipe.run(((pts, lens) for _ in range(50)))):
            if i == 1:
                1 / 0                                      # exception in the consumer
    assert threading.active_count() == before
    again = [d.cpu() for d in pipe.run([(pts, lens)] * 2)]   # still usable, same streams
    assert torch.equal(again[1], full[0]) and pipe._streams == streams
    pipe.close()
    pipe.close()                                           # idempotent
    assert pipe._streams == [] and all(s._lcr_parked for s in streams)
    pipe2 = DescriptorPipeline(m, 0.3, 1.275, 4, [40, 40, 40, 40], raw_voxel=0.3, pre_workers=2)
    assert any(any(s is t for t in streams) for s in pipe2._streams)      # parked streams are taken back, not leaked
    out2 = [d.cpu() for d in pipe2.run([(pts, lens)] * 2)]
    assert torch.equal(out2[0], full[0])
    pipe2.close()


def test_pair_pipeline_early_exit_and_close():
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.pipeline import PairPipeline
    from lcrnet_amd.weights import seeded_state_dict
    from conftest import LIMITS, load_scan
    cfg = make_cfg()
    cfg["neighbor_limits"] = LIMITS
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
    m = m.cuda()
    a, b = torch.from_numpy(load_scan("003528")).cuda(), torch.from_numpy(load_scan("004481")).cuda()
    pair = (torch.cat([a, b]), torch.tensor([len(a), len(b)], dtype=torch.int64, device="cuda"))
    before = threading.active_count()
    with PairPipeline(m, neighbor_limits=LIMITS, workers=2, pairs_per_call=2) as pp:
        gen = pp.run(pair for _ in range(40))
        one = next(gen)
        assert one["estimated_transform"].shape == (4, 4)
        gen.close()
        assert threading.active_count() == before
        outs = list(pp.run([pair] * 3))                    # a last group of one pair
        assert len(outs) == 3
    assert pp._streams == []
    with pytest.raises(RuntimeError):
        list(pp.run([pair] * 2))
