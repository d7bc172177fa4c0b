"""Golden for the loop-detection -> registration hand-off from the IMPORTED reference (build container only):
    python tests/golden/make_golden_top1.py

Runs, unmodified, experiments/inference/infer_loop_detection_find_top1.py:inference_one_epoch (descriptor files -> re-normalisation :75 ->
per-query search loop -> predicted_des_L2_dis.npz -> find_top1 -> `result/top1_with_thres_%.2f/%02d.txt`) on the 400-frame case of
make_golden_retrieval.py (duplicates, loops), with the descriptors stored UN-normalised (scaled by a per-frame factor) so that the
re-normalisation matters, at two thresholds; then the reference's own reader of that file (datasets/loop_closure/kitti/dataset.py:
make_dataset_kitti, mode 'infer').  faiss is replaced by the exhaustive stub of make_golden_retrieval.py (see its header).

Output: tests/golden/top1_golden.npz — the stored descriptors, the rows, per threshold the text of the file and the (pos, anc) pairs the
reference's reader makes of it, and a pose line the way infer_registration.py:77-78 formats it."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402
import make_golden_retrieval as mgr  # noqa: E402


def stored_descriptors():
    d, _ = mgr.small_case()
    rng = np.random.default_rng(21)
    sig = [0.01, 0.02, 0.03, 0.05]                      # squared distances ~ 256 sigma^2 = 0.026 .. 0.64 around the two thresholds
    for i in range(150, len(d)):
        if i % 3 == 0:
            v = d[i - 120] + sig[(i // 3) % 4] * rng.standard_normal(d.shape[1]).astype(np.float32)
            d[i] = v / np.linalg.norm(v)
    scale = np.random.default_rng(9).uniform(0.5, 3.0, (len(d), 1)).astype(np.float32)
    return (d * scale).astype(np.float32)


def main():
    mgm.install_stubs()
    sys.path.insert(0, mgm.REF)
    sys.modules["faiss"] = mgr._FaissStub("faiss")
    sys.modules.pop("IPython", None)                    # matplotlib inspects a loaded IPython; this script does not need the stub
    import matplotlib
    matplotlib.use("Agg")
    import experiments.inference.infer_loop_detection_find_top1 as ft
    from easydict import EasyDict
    from lcrnet_amd.io_formats import save_descriptor
    desc = stored_descriptors()
    store = {"stored_descriptors": desc}
    for thres in (0.11, 0.5):
        tmp = tempfile.mkdtemp(prefix="lcr_top1_")
        feat = os.path.join(tmp, "kitti")
        os.makedirs(feat)
        for i in range(len(desc)):
            save_descriptor(feat, 0, i, desc[i])
        cfg = EasyDict(ld_feature_dir=tmp + "/", dataset="kitti", data=EasyDict(dataset_root=os.path.join(tmp, "data")))
        ft.inference_one_epoch(cfg, [0], thres)
        rows = np.load(os.path.join(feat, "predicted_des_L2_dis.npz"))["arr_0"]
        name = "%s/result/top1_with_thres_%.2f/%02d.txt" % (cfg.data.dataset_root, thres, 0)
        text = open(name).read()
        # the reference's reader: `osp.join(txt_path, '%02d' % seq)` — it wants the file without the .txt suffix
        root = os.path.dirname(name)
        shutil.copy(name, os.path.join(root, "00"))
        from experiments.lcrnet.datasets.loop_closure.kitti.dataset import make_dataset_kitti
        meta = make_dataset_kitti(root, "infer", seq=[0])
        pairs = np.array([[m["frame0"], m["frame1"]] for m in meta], dtype=np.int64).reshape(-1, 2)
        tag = "thres_%.2f" % thres
        store.update({tag + "/text": np.array(text), tag + "/pairs_pos_anc": pairs})
        store["rows"] = rows
        print("%s: %d lines, %d pairs, first line %r" % (tag, text.count("\n"), len(pairs), text.splitlines()[0] if text else ""))
        shutil.rmtree(tmp)
    T = np.random.default_rng(4).standard_normal((4, 4)).astype(np.float32)
    M2 = T.reshape(-1)[:12]
    store["pose_T"] = T
    store["pose_line"] = np.array(f'{17} {250} {M2[0]:.6f} {M2[1]:.6f} {M2[2]:.6f} {M2[3]:.6f} {M2[4]:.6f} {M2[5]:.6f} {M2[6]:.6f} {M2[7]:.6f} '
                                  f'{M2[8]:.6f} {M2[9]:.6f} {M2[10]:.6f} {M2[11]:.6f} \n')     # the f-string of infer_registration.py:78
    np.savez_compressed(os.path.join(HERE, "top1_golden.npz"), **store)


if __name__ == "__main__":
    main()
