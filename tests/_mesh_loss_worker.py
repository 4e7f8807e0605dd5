"""Cases of tests/test_gpu_parity.py::test_mesh_loss_grid_search_equals_full_search: the mesh reprojection loss of a fixed set of
seeded cases, evaluated by a loss-only context (never finalized: SMPL / networks are not needed for the loss operators) whose
pixel -> vertex search is selected by the ``mesh_a2b`` plan option (HpeConfig).  ``python _mesh_loss_worker.py [grid|mfma|valu]``
prints one JSON dict of per-image losses per case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpe_amd  # noqa: E402
from hpe_amd import synthetic  # noqa: E402


def cases():
    """(name, seg [B,224,224,1], sil_pred [B,6890,2]) -- spread meshes, concentrated ones, meshes off the silhouette and off the
    image, integer coordinates (exact ties) and duplicated vertices."""
    g = np.random.Generator(np.random.Philox(4242))
    B = 6
    seg, _ = synthetic.make_lsp_targets(B, seed=31)
    seg[1, :, :, 0] *= (np.arange(224)[None, :] % 3 == 0)
    seg[5] = 0.0
    seg[5, 3, 220] = 1.0
    v = np.zeros((B, 6890, 2), np.float32)
    for b in range(B):
        ys, xs = np.where(seg[b, :, :, 0] > 0)
        pick = g.integers(0, len(ys), 6890)
        v[b, :, 0] = xs[pick] + g.uniform(-1, 1, 6890)
        v[b, :, 1] = ys[pick] + g.uniform(-1, 1, 6890)
    out = [("spread over the silhouette", seg, v.copy())]
    w = v.copy()
    w[0] = 112.0 + g.normal(0, 6.0, (6890, 2))       # ~40 occupied cells: around the grid / full search switch
    w[1] = g.uniform(-150, 400, (6890, 2))           # two thirds outside the image
    w[2] = np.round(g.uniform(40, 190, (6890, 2)))   # integer coordinates: exact ties between different vertices
    w[2, 3000:4000] = w[2, :1000]                    # duplicated vertices
    w[3, :, 0] = g.uniform(0, 60, 6890)              # mesh in the left quarter, silhouette in the middle: many rings
    w[3, :, 1] = g.uniform(0, 224, 6890)
    w[4] = np.round(g.uniform(100, 124, (6890, 2)) * 2) / 2   # half-pixel lattice, 49 x 49 positions: ties of every order
    out.append(("edge cases", seg, w))
    return out


def run(mode):
    eng = hpe_amd.HpeEngine(device=0, max_batch=6, mesh_a2b=mode)
    res = {}
    for name, seg, v in cases():
        per_image = []
        for b in range(seg.shape[0]):
            val = hpe_amd.mesh_reprojection_loss(eng, torch.from_numpy(seg[b:b + 1]).cuda(), torch.from_numpy(v[b:b + 1]).cuda())
            per_image.append(float(val))
        both = float(hpe_amd.mesh_reprojection_loss(eng, torch.from_numpy(seg).cuda(), torch.from_numpy(v).cuda()))
        res[name] = {"per_image": per_image, "batch": both}
    eng.close()
    return res


if __name__ == "__main__":
    print("MESH_LOSS_JSON " + json.dumps(run(sys.argv[1] if len(sys.argv) > 1 else "grid")))
