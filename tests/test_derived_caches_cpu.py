"""CPU: the derived-tensor caches (host kernel points, transposed / split weights, the native encoder's table key) follow the weights:
in-place updates, replaced tensors and Module._apply all invalidate them; deep copies and pickles start cold."""
import copy
import io

import numpy as np
import torch

from lcrnet_amd import functional as F
from lcrnet_amd import native_encoder
from lcrnet_amd.model_family import create_model
from lcrnet_amd.modules.kpconv.kpconv import KPConv
from lcrnet_amd.modules.kpconv.modules import LastUnaryBlock, UnaryBlock


def test_weight_stamp_is_an_identity_not_an_address():
    t = torch.zeros(8)
    s = F.WeightStamp(t)
    assert s.same(t)
    u = torch.zeros(8)
    assert not s.same(u)                                   # another object, whatever its address
    t.add_(1)
    assert not s.same(t)                                   # same object, newer version
    assert not copy.deepcopy(F.WeightStamp(u)).same(u)     # copies match nothing


def test_kpconv_caches_follow_the_weights():
    k = KPConv(4, 8, 15, 1.0, 0.6)
    kp0, wt0 = k.kernel_points_host(), k.weights_t()
    assert k.kernel_points_host() is kp0 and k.weights_t() is wt0                     # hits
    with torch.no_grad():
        k.kernel_points.mul_(2.0)
        k.weights.add_(1.0)
    assert np.array_equal(k.kernel_points_host(), k.kernel_points.numpy()) and k.kernel_points_host() is not kp0
    assert torch.equal(k.weights_t(), k.weights.detach().reshape(60, 8).t())
    k._buffers["kernel_points"] = k.kernel_points.clone() * 3.0                       # a NEW tensor, version 0, like Module._apply makes
    assert np.array_equal(k.kernel_points_host(), k.kernel_points.numpy())
    k.kernel_points_host(), k.weights_t()
    k.double()                                                                         # _apply drops everything
    assert k._kp_cache is None and k._wt_cache is None
    assert k.weights_t().dtype == torch.float64


def test_split_weight_caches_follow_the_weights(monkeypatch):
    calls = []
    monkeypatch.setattr(F, "split_bf16x3", lambda w: calls.append(1) or w.detach().clone())
    monkeypatch.setattr(F, "publish_derived", lambda t: t)
    monkeypatch.setattr(F, "gemm_split_enabled", lambda: True)
    monkeypatch.setattr(F, "gemm_split_ok", lambda n, k: True)
    monkeypatch.setattr(F, "gemm_bsplit", lambda x, planes, bias=None, **kw: (x @ planes.t() + (bias if bias is not None else 0), None))
    u = UnaryBlock(32, 64, 8)
    a = u.weight_split()
    assert u.weight_split() is a and len(calls) == 1
    with torch.no_grad():
        u.mlp.weight.add_(1.0)
    assert u.weight_split() is not a and len(calls) == 2
    u.float()
    assert u._ws_cache is None
    last = LastUnaryBlock(32, 16)
    x = torch.randn(5, 32)
    y0 = last(x)
    assert torch.allclose(y0, x @ last.mlp.weight.t() + last.mlp.bias, atol=1e-6) and len(calls) == 3
    last(x)
    assert len(calls) == 3
    with torch.no_grad():
        last.mlp.weight.mul_(0.5)
    assert torch.allclose(last(x), x @ last.mlp.weight.t() + last.mlp.bias, atol=1e-6) and len(calls) == 4
    k = KPConv(4, 8, 15, 1.0, 0.6)
    s0 = k.weights_t_split()
    assert k.weights_t_split() is s0
    with torch.no_grad():
        k.weights.add_(1.0)
    assert k.weights_t_split() is not s0


def test_native_table_key_and_module_copies():
    m = create_model().eval()
    enc = m.encoder
    key = native_encoder._Key(enc)
    assert key.valid(enc)
    name, buf = next(iter(enc.named_buffers()))
    with torch.no_grad():
        buf.add_(0.0)                                      # an in-place write, even of zeros
    assert not key.valid(enc)
    key = native_encoder._Key(enc)
    owner = enc
    for part in name.split(".")[:-1]:
        owner = getattr(owner, part)
    owner._buffers[name.split(".")[-1]] = buf.clone()      # replaced by a new tensor
    assert not key.valid(enc)
    enc._native_table = (native_encoder._Key(enc), object())
    m2 = copy.deepcopy(m)
    assert m2.encoder._native_table is None                # raw-pointer tables never travel with a copy
    m.float()
    assert enc._native_table is None
    torch.save(m, io.BytesIO())


def test_native_table_key_sees_structural_changes():
    """The key keeps (container, name) slots instead of walking named_parameters() on every forward (round 6): a replaced sub-module, a
    replaced / added / removed parameter and the GEMM-form switch must all still fail the check; an untouched module passes it repeatedly."""
    import torch.nn as nn
    from lcrnet_amd.modules.kpconv.modules import ResidualBlock
    m = create_model().eval()
    enc = m.encoder
    key = native_encoder._Key(enc)
    assert key.valid(enc) and key.valid(enc)
    old = enc.encoder2_2
    enc.encoder2_2 = ResidualBlock(old.in_channels, old.out_channels, 15, 1.0, 0.6, 32)      # a NEW sub-module of the same shape
    assert not key.valid(enc)
    enc.encoder2_2 = old
    assert key.valid(enc)                                       # the very same objects again
    lin = enc.encoder3_2.unary1.mlp
    w = lin.weight
    lin.weight = nn.Parameter(w.detach().clone())               # a new Parameter object at a new address
    assert not key.valid(enc)
    lin.weight = w
    assert key.valid(enc)
    lin.register_buffer("extra", torch.zeros(1))                # a tensor that was not there when the key was made
    assert not key.valid(enc)
    del lin._buffers["extra"]
    assert key.valid(enc)
    was = F.gemm_split_enabled()
    F.set_gemm_split(not was)
    try:
        assert not key.valid(enc)                               # the table holds (or lacks) split planes: bound to the GEMM form
    finally:
        F.set_gemm_split(was)
    assert key.valid(enc)


def test_roformer_native_table_follows_the_weights():
    from lcrnet_amd import native_roformer
    from lcrnet_amd.modules.thdroformer import ThDRoFormer
    tf = ThDRoFormer(1024, 256, 128, 4, 2).eval()
    w = native_roformer._table(tf)
    assert w.num_blocks == 4 and [w.block_is_self[i] for i in range(4)] == [1, 0, 1, 0] and (w.d_in, w.d_model, w.d_out, w.heads) == (1024, 128, 256, 4)
    assert w.layers[1].q.w == tf.transformer.layers[1].attention.attention.proj_q.weight.data_ptr()
    assert w.layers[3].ln2_b == tf.transformer.layers[3].output.norm.bias.data_ptr() and abs(w.layers[0].ln1_eps - 1e-5) < 1e-12
    t1 = native_roformer.table_for(tf)
    assert native_roformer.table_for(tf) is t1                  # cached
    with torch.no_grad():
        tf.out_proj.bias.add_(1.0)
    assert native_roformer.table_for(tf) is not t1              # rebuilt after an in-place update
    assert native_roformer.eligible(tf, torch.zeros(1, 1024)) is False      # CPU features: the module tree (which then refuses CPU tensors itself)
    tf.double()
    assert "_native_table" not in tf.__dict__                   # _apply drops the raw-pointer table
    assert "_native_table" not in copy.deepcopy(tf).__dict__
    tk = ThDRoFormer(1024, 256, 128, 4, 2, k=[0.5, 0.5])
    assert not native_roformer.eligible(tk, torch.zeros(1, 1024))           # top-k attention runs through the module tree
