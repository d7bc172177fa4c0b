"""The six stack-mode collates of experiments/lcrnet/data.py:77-406, mirrored by name in lcrnet_amd.data (f-4), against the dicts the
IMPORTED reference returned for the same seeded samples (tests/golden/make_golden_collates.py -> collates_golden.npz).
CPU part: key sets, stacking order, popped / unwrapped keys, values — exact.  GPU part: the precomputed lists of one case equal the
reference's own (reference C++ ops) bit for bit (SHA-256 of the int64 / f32 tensors)."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "collates_golden.npz"), allow_pickle=False)


def _check(name, out, gold):
    assert sorted(out.keys()) == [str(k) for k in gold[f"{name}/__keys"]], name
    for k, v in out.items():
        if isinstance(v, list):
            assert int(gold[f"{name}/{k}#len"]) == len(v), (name, k)
            items = [(f"{name}/{k}#{i}", x) for i, x in enumerate(v)]
        else:
            assert f"{name}/{k}#len" not in gold.files, (name, k, "the reference keeps a list here")
            items = [(f"{name}/{k}", v)]
        for key, x in items:
            want = gold[key]
            got = x.numpy() if torch.is_tensor(x) else np.asarray(x)
            assert got.shape == want.shape and got.dtype == want.dtype, (key, got.shape, want.shape, got.dtype, want.dtype)
            assert np.array_equal(got, want), key


def test_collates_equal_the_reference_without_precompute(gold):
    import lcrnet_amd.data as mine
    from make_golden_collates import ARGS, samples
    for name, (fn, batch) in samples().items():
        out = getattr(mine, fn)(batch, *ARGS, precompute_data=False)
        _check(name, out, gold)


@pytest.mark.gpu
def test_online_collate_precompute_equals_the_reference_lists(gold):
    import lcrnet_amd.data as mine
    from make_golden_collates import ARGS, samples
    fn, batch = samples()["online_b2"]
    out = getattr(mine, fn)(batch, *ARGS, precompute_data=True)
    assert out["features"].is_cuda and out["batch_size"] == 2
    for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
        for i, t in enumerate(out[key]):
            a = np.ascontiguousarray(t.cpu().numpy())
            assert list(a.shape) == gold[f"online_b2_pre/{key}{i}_shape"].tolist(), (key, i, a.shape)
            assert hashlib.sha256(a.tobytes()).hexdigest() == str(gold[f"online_b2_pre/{key}{i}_sha"]), (key, i)
