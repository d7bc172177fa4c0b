"""Generate golden vectors for the floating-point modules by IMPORTING the reference Python model (build container only).

    python tests/golden/make_golden_model.py

What runs: the reference's own classes, unmodified, from /root/reference (never copied):
  experiments.lcrnet.model_family.LCRNet_GlobalDescrition.LCRNet_GlobalDescrition  (encoder + NetVLAD, eval forward)
  experiments.lcrnet.model_family.LCRNet.LCRNet                                    (sub-modules encoder / transformer /
                                                                                    GlobalDescritionHEAD called as
                                                                                    KeypointDetection does, LCRNet.py:124-151)
on CPU, with the index tensors produced by the reference C++ ops (oracle/_ref).  Third-party modules that the model files
import but that are not installed here are stubbed for the import only (easydict -> attribute dict; open3d -> just enough
to read the 482-byte kernel disposition PLY; IPython / ipdb / coloredlogs / pytorch_metric_learning -> empty), and
``np.int`` is restored (rpetransformer.py:48 needs it).  Weights: lcrnet_amd.weights.seeded_state_dict(seed=7351) loaded
into the reference model — the same pure function the tests apply to the build's model, so no checkpoint file is needed.

Outputs (tests/golden/model_golden.npz, model_manifest.json):
  * state-dict manifests (key -> shape, dtype) of both reference models;
  * per demo scan (single-scan stack, limits [74,68,70,67]): 256-D descriptor `anc_global`;
  * for 003854: 48 sampled rows of every encoder block output + its mean/abs-mean, full coarse features (844,1024);
  * for the pair 003854/000958 (pair-stacked, GroupNorm over both clouds like the reference): descriptors of both clouds
    from the PRE-transformer features (LCRNet.py:296-297), 3D-RoFormer outputs (sampled rows + checksums).
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
LIMITS = [74, 68, 70, 67]
SEED = 7351


def install_stubs():
    np.int = int

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    m = types.ModuleType("easydict")
    m.EasyDict = EasyDict
    sys.modules["easydict"] = m

    o3d = types.ModuleType("open3d")

    class _PCD:
        def __init__(self):
            self.points = None

    def read_point_cloud(path):
        raw = open(path, "rb").read()
        i = raw.index(b"end_header\n") + len(b"end_header\n")
        p = _PCD()
        p.points = np.frombuffer(raw[i:], dtype="<f8").reshape(-1, 3)
        return p

    o3d.io = types.SimpleNamespace(read_point_cloud=read_point_cloud, write_point_cloud=lambda *a, **k: None)
    o3d.geometry = types.SimpleNamespace(PointCloud=_PCD)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda x: x)
    o3d.visualization = types.SimpleNamespace()
    sys.modules["open3d"] = o3d
    for name in ["IPython", "ipdb", "coloredlogs", "pytorch_metric_learning", "pytorch_metric_learning.distances",
                 "pytorch_metric_learning.losses", "pytorch_metric_learning.miners", "pytorch_metric_learning.reducers"]:
        mod = types.ModuleType(name)
        mod.embed = lambda *a, **k: None
        mod.set_trace = lambda *a, **k: None
        mod.__getattr__ = lambda attr: (lambda *a, **k: None)
        sys.modules[name] = mod


def install_ref_ext():
    """`utils.ext` is looked up by name at import (modules/ops/grid_subsample.py:4); serve it from oracle/_ref."""
    from oracle import ops

    ext = types.ModuleType("utils.ext")

    def grid_subsampling(points, lengths, voxel):
        p, l = ops.grid_subsample(points.numpy(), lengths.numpy(), voxel, impl="ref")
        return torch.from_numpy(p), torch.from_numpy(l)

    def radius_neighbors(q, s, ql, sl, radius):
        return torch.from_numpy(ops.radius_search(q.numpy(), s.numpy(), ql.numpy(), sl.numpy(), radius, -1, impl="ref"))

    ext.grid_subsampling = grid_subsampling
    ext.radius_neighbors = radius_neighbors
    ext.radius_filter = None
    sys.modules["utils.ext"] = ext


def rows(n, k=48, seed=0):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False))


def main():
    install_stubs()
    sys.path.insert(0, REF)
    install_ref_ext()
    from lcrnet_amd.weights import seeded_state_dict
    from experiments.lcrnet.config_model import make_cfg
    from experiments.lcrnet.data import precompute_data_stack_mode
    from experiments.lcrnet.model_family.LCRNet_GlobalDescrition import LCRNet_GlobalDescrition
    from experiments.lcrnet.model_family.LCRNet import LCRNet

    cfg = make_cfg()
    cfg.neighbor_limits = LIMITS
    cfg.vis = False
    torch.manual_seed(0)
    np.random.seed(0)
    gd = LCRNet_GlobalDescrition(cfg).eval()
    full = LCRNet(cfg).eval()
    manifest = {
        "LCRNet_GlobalDescrition": {k: [list(v.shape), str(v.dtype)] for k, v in gd.state_dict().items()},
        "LCRNet": {k: [list(v.shape), str(v.dtype)] for k, v in full.state_dict().items()},
        "seed": SEED,
    }
    json.dump(manifest, open(os.path.join(OUT, "model_manifest.json"), "w"), indent=0)
    sd_full = seeded_state_dict(full.state_dict(), SEED)
    full.load_state_dict(sd_full, strict=True)
    gd.load_state_dict({k: v for k, v in sd_full.items() if k in gd.state_dict()}, strict=True)

    store = {}
    scans = {f[:-4]: np.load(os.path.join(OUT, "scans", f)) for f in sorted(os.listdir(os.path.join(OUT, "scans")))}

    # hook every encoder block
    trace = {}
    for name, mod in gd.encoder.named_children():
        mod.register_forward_hook(lambda m, i, o, name=name: trace.__setitem__(name, o.detach()))

    with torch.no_grad():
        for name, xyz in scans.items():
            pts = torch.from_numpy(xyz)
            dd = precompute_data_stack_mode(pts, torch.LongTensor([len(xyz)]), 4, 0.3, 1.275, LIMITS)
            dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
            dd["features"] = torch.ones(len(xyz), 1)
            dd["batch_size"] = 1
            out = gd(dd)
            store[f"{name}/anc_global"] = out["anc_global"].numpy()
            print(name, "descriptor norm", float(out["anc_global"].norm()), out["anc_global"][0, :4].numpy())
            if name == "003854":
                for bname, o in trace.items():
                    r = rows(o.shape[0])
                    store[f"{name}/{bname}_rows"] = r
                    store[f"{name}/{bname}_vals"] = o[r].numpy()
                    store[f"{name}/{bname}_stats"] = np.array([o.mean().item(), o.abs().mean().item(), o.shape[0], o.shape[1]])
                store[f"{name}/feats_c"] = trace["encoder4_3"].numpy()

        # pair, as registration_collate_fn_stack_mode stacks it (data.py:110-113): [ref(pos), src(anc)]
        a, b = scans["003854"], scans["000958"]
        pts = torch.from_numpy(np.concatenate([a, b]))
        dd = precompute_data_stack_mode(pts, torch.LongTensor([len(a), len(b)]), 4, 0.3, 1.275, LIMITS)
        dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
        feats_list = full.encoder(torch.ones(len(pts), 1), dd)
        feats_c = feats_list[-1]
        n0 = int(dd["lengths"][-1][0])
        pos_c, anc_c = feats_c[:n0], feats_c[n0:]
        pc = dd["points"][-1]
        e0, e1 = full.transformer(pc[:n0][None], pc[n0:][None], pos_c[None], anc_c[None])
        g0, g1 = full.GlobalDescritionHEAD(pos_c), full.GlobalDescritionHEAD(anc_c)
        store["pair/pos_global"], store["pair/anc_global"] = g0.numpy(), g1.numpy()
        store["pair/l2"] = np.sqrt(((g0 - g1) ** 2).sum().item())
        store["pair/n_c"] = np.array([n0, feats_c.shape[0] - n0])
        for tag, e in (("pos", e0[0]), ("anc", e1[0])):
            r = rows(e.shape[0], 64, seed=1)
            store[f"pair/{tag}_tf_rows"] = r
            store[f"pair/{tag}_tf_vals"] = e[r].numpy()
            store[f"pair/{tag}_tf_stats"] = np.array([e.mean().item(), e.abs().mean().item(), e.shape[0], e.shape[1]])
        r = rows(feats_c.shape[0], 64, seed=2)
        store["pair/feats_c_rows"], store["pair/feats_c_vals"] = r, feats_c[r].numpy()
        print("pair descriptor L2", store["pair/l2"], "coarse nodes", store["pair/n_c"])
    np.savez_compressed(os.path.join(OUT, "model_golden.npz"), **store)
    print("wrote model_golden.npz", os.path.getsize(os.path.join(OUT, "model_golden.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
