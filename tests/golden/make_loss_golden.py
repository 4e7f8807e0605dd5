"""Generates tests/golden/losses_b2.npz with the CPU oracle: the config-5 losses (src/ops.py:35-137, src/trainer.py:274-296) of
the three IEF stages on the golden path's own outputs (images of tests/golden/make_golden.py, seeded silhouettes / keypoints).
Produced by the oracle restatement in the build container ("parity unpinned" w.r.t. the reference's own outputs); pins the loss
oracle against drift and the HIP loss kernels on inputs a forward pass produces.  Run:  python tests/golden/make_loss_golden.py
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
