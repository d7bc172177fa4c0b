"""ctypes side of lcr_roformer_forward (csrc/roformer.hip): the weight table of a ThDRoFormer and the one-call forward.

Like native_encoder: the table holds raw device pointers into the module's own parameters, its validity is a WeightStamp of every
parameter (walked once, checked by dictionary lookups)."""
import ctypes

import torch

from . import _lib
from .native_encoder import _Key

MAX_BLOCKS = 16
_fp = ctypes.c_void_p


class LinearW(ctypes.Structure):
    _fields_ = [("w", _fp), ("b", _fp)]


class LayerW(ctypes.Structure):
    _fields_ = [("q", LinearW), ("k", LinearW), ("v", LinearW), ("lin", LinearW), ("ln1_w", _fp), ("ln1_b", _fp), ("ln1_eps", ctypes.c_float),
                ("expand", LinearW), ("squeeze", LinearW), ("ln2_w", _fp), ("ln2_b", _fp), ("ln2_eps", ctypes.c_float)]


class RoformerW(ctypes.Structure):
    _fields_ = [("d_in", ctypes.c_int), ("d_model", ctypes.c_int), ("d_out", ctypes.c_int), ("heads", ctypes.c_int), ("num_blocks", ctypes.c_int),
                ("block_is_self", ctypes.c_int * MAX_BLOCKS), ("emb1", LinearW), ("emb2", LinearW), ("in_proj", LinearW), ("out_proj", LinearW),
                ("layers", LayerW * MAX_BLOCKS)]


def _lin(m):
    return LinearW(m.weight.data_ptr(), m.bias.data_ptr() if m.bias is not None else None)


def _table(tf):
    w = RoformerW()
    t = tf.transformer
    w.d_in, w.d_model, w.d_out = tf.in_proj.in_features, tf.in_proj.out_features, tf.out_proj.out_features
    w.heads = t.layers[0].attention.attention.num_heads
    w.num_blocks = len(t.blocks)
    w.emb1, w.emb2, w.in_proj, w.out_proj = _lin(tf.embedding.encoder), _lin(tf.embedding.encoder2), _lin(tf.in_proj), _lin(tf.out_proj)
    for i, (kind, layer) in enumerate(zip(t.blocks, t.layers)):
        w.block_is_self[i] = int(kind == "self")
        a, o, L = layer.attention, layer.output, w.layers[i]
        L.q, L.k, L.v, L.lin = _lin(a.attention.proj_q), _lin(a.attention.proj_k), _lin(a.attention.proj_v), _lin(a.linear)
        L.ln1_w, L.ln1_b, L.ln1_eps = a.norm.weight.data_ptr(), a.norm.bias.data_ptr(), float(a.norm.eps)
        L.expand, L.squeeze = _lin(o.expand), _lin(o.squeeze)
        L.ln2_w, L.ln2_b, L.ln2_eps = o.norm.weight.data_ptr(), o.norm.bias.data_ptr(), float(o.norm.eps)
    return w


def eligible(tf, feats):
    t = tf.transformer
    return (feats.is_cuda and feats.dtype == torch.float32 and t.k is None and not t.parallel and len(t.blocks) <= MAX_BLOCKS
            and tf.embedding.encoder2.out_features * 2 == tf.in_proj.out_features)


def table_for(tf):
    cached = tf.__dict__.get("_native_table")
    if cached is None or not cached[0].valid(tf):
        from . import functional as F
        with F.derived_lock:
            cached = tf.__dict__.get("_native_table")
            if cached is None or not cached[0].valid(tf):
                cached = (_Key(tf), _table(tf))
                tf.__dict__["_native_table"] = cached
    return cached[1]


def forward(tf, points, feats, lens0, lens1):
    """points [n,3] / feats [n,d_in]: rows of all first clouds, then all second clouds; lens0 / lens1: host sequences (P entries each)
    -> (out [n,d_out], theta [n,d_model/2])."""
    w = table_for(tf)
    P = len(lens0)
    n = int(points.shape[0])
    dev = feats.device
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.lcr_roformer_ws_bytes(ctypes.byref(w), n, ctypes.byref(nbytes)), "lcr_roformer_ws_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    out = torch.empty((n, w.d_out), dtype=torch.float32, device=dev)
    theta = torch.empty((n, w.d_model // 2), dtype=torch.float32, device=dev)
    l0, l1 = (ctypes.c_int64 * P)(*[int(x) for x in lens0]), (ctypes.c_int64 * P)(*[int(x) for x in lens1])
    _lib.check(L.lcr_roformer_forward(ctypes.byref(w), _lib.ptr(points.contiguous()), _lib.ptr(feats.contiguous()), l0, l1, P, _lib.ptr(out),
                                      _lib.ptr(theta), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "lcr_roformer_forward")
    return out, theta
