"""Generates tests/golden/path_b2.npz with the CPU oracle (oracle/hmr_oracle.py) in the build container.

The reference itself cannot run here (TensorFlow/Keras absent) and ships no fixtures, so these
vectors are produced by the oracle restatement ("parity unpinned" w.r.t. the reference's own outputs).  They pin
(a) the oracle against drift of numpy/torch versions between the build container and the GPU box and (b) the
HIP path at points inside the network (per-block checksums), on the seeded synthetic assets of
hpe_amd/synthetic.py.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import hpe_amd  # noqa: E402
from hpe_amd import synthetic  # noqa: E402
from oracle import hmr_oracle as O  # noqa: E402

B = 2
IMG_SEED = 2024


def main():
    smpl = synthetic.make_smpl_model()
    enc = synthetic.make_encoder_params()
    reg = synthetic.make_regressor_params()
    mean = O.load_mean_param(synthetic.make_mean_params())
    img = synthetic.make_images(B, seed=IMG_SEED)
    feat, taps = O.resnet50_features(img, enc, return_taps=True)
    osm = O.SMPL(smpl)
    res = O.predict(img, enc, reg, osm, mean, features=feat, all_stages=True)
    out = {
        "features_head": feat[:, :32],
        "features_sum": feat.astype(np.float64).sum(1),
        "stage_theta": np.stack(res["stage_theta"]),
        "stage_joints": np.stack(res["stage_joints"]),
        "stage_kp2d": np.stack(res["stage_kp2d"]),
        "stage_J24": np.stack(res["stage_J_transformed"]),
        "verts_strided": res["generated_verts"][:, ::53],
        "verts_sum": res["generated_verts"].astype(np.float64).sum(1),
    }
    for k, v in taps.items():
        out["tap_mean_" + k] = np.array([v.astype(np.float64).mean(), np.abs(v.astype(np.float64)).mean()])
    # SMPL-only vectors on plausible thetas
    th = synthetic.make_thetas(3, seed=99)
    v, j, Rs = osm(th[:, 75:], th[:, 3:75], get_skin=True)
    out.update(smpl_theta=th, smpl_joints=j, smpl_J24=osm.J_transformed, smpl_verts_strided=v[:, ::53], smpl_Rs=Rs)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "path_b2.npz")
    np.savez_compressed(path, **{k: np.asarray(v, np.float32 if np.asarray(v).dtype != np.float64 else np.float64) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
