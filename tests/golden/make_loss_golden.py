"""Generates tests/golden/losses_b2.npz with the CPU oracle: the config-5 losses (src/ops.py:35-137, src/trainer.py:274-296) of
the three IEF stages on the golden path's own outputs (images of tests/golden/make_golden.py, seeded silhouettes / keypoints).
Produced by the oracle restatement in the build container -- ORACLE-GENERATED, i.e. "parity unpinned" w.r.t. the reference's own
outputs (the reference holds no fixture for src/ops.py and cannot run here); it pins the loss oracle against drift and the HIP loss
kernels on inputs a forward pass produces.  The quirks of the reference it encodes, each restated from the cited lines only:
  * silhouette points are (x = column, y = row) in tf.where order (row-major)            src/trainer.py:291, src/ops.py:123-125
  * L2 distance B -> A, L1 distance A -> B                                                src/ops.py:91-96
  * the denominator is silhouette_gt.shape[1] + silhouette_pred.shape[1] = 3 + 6890      src/ops.py:129-130
  * kp loss = sum(vis * |d|) / (2 * #visible), 0 when nothing is visible                  src/ops.py:35-47 (SUM_BY_NONZERO_WEIGHTS)
The nearest-neighbour part alone has an independent pin (scipy cKDTree, tests/test_oracle_kat.py).
Run:  python tests/golden/make_loss_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import hpe_amd  # noqa: E402,F401
from hpe_amd import synthetic  # noqa: E402
from oracle import hmr_oracle as O  # noqa: E402

B = 2
IMG_SEED = 2024
TARGET_SEED = 2025


def main():
    smpl = synthetic.make_smpl_model()
    enc = synthetic.make_encoder_params()
    reg = synthetic.make_regressor_params()
    mean = O.load_mean_param(synthetic.make_mean_params())
    img = synthetic.make_images(B, seed=IMG_SEED)
    seg, kp_gt = synthetic.make_lsp_targets(B, seed=TARGET_SEED)
    res = O.predict(img, enc, reg, O.SMPL(smpl), mean, all_stages=True)
    lo = O.val_losses(res["stage_verts"], res["stage_cams"], res["stage_kp2d"], seg, kp_gt)
    verts2d = [O.reproject_vertices(v, c, np.array([224.0, 224.0], np.float32)) for v, c in zip(res["stage_verts"], res["stage_cams"])]
    out = {
        "kpr_losses": np.asarray(lo["kpr_losses"], np.float64),
        "mr_losses": np.asarray(lo["mr_losses"], np.float64),
        "verts2d_strided": np.stack(verts2d)[:, :, ::53].astype(np.float32),
        "seg_count": np.array([(seg[b] > 0).sum() for b in range(B)], np.int64),
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "losses_b2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", out["kpr_losses"], out["mr_losses"])


if __name__ == "__main__":
    main()
