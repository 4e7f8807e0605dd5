"""Time the mesh reprojection loss (src/ops.py:117-137) on the bench's own config-5 inputs, stage by stage, plus a mesh scaled to
cover the silhouette (what a trained regressor produces).  HPE_MESH_A2B selects the pixel->vertex search (grid / mfma / valu).

    python tools/mesh_loss_bench.py [B [case]]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hpe_amd  # noqa: E402
from hpe_amd import synthetic  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256

    class Cfg(object):
        img_size, num_stage, batch_size, data_format = 224, 3, B, "NHWC"
        checkpoint_dir = smpl_model_path = None
        encoder_dtype = "fp32"

    pred = hpe_amd.Predictor(Cfg(), smpl_model=synthetic.make_smpl_model(), mean_params=synthetic.make_mean_params(),
                             encoder_params=synthetic.make_encoder_params(), regressor_params=synthetic.make_regressor_params())
    eng = pred.engine
    images = torch.from_numpy(synthetic.make_images(B, seed=1000)).cuda()
    seg_np, _ = synthetic.make_lsp_targets(B, seed=2000)
    seg = torch.from_numpy(seg_np[..., 0].copy()).cuda()
    outs = eng.forward(images, all_stages=True, want=eng.DEFAULT_OUTPUTS + ("verts2d",))
    torch.cuda.synchronize()
    cases = [("stage %d" % (s + 1), outs[s]["verts2d"]) for s in range(3)]
    # mesh stretched to the silhouette's bounding box, image by image
    v = outs[0]["verts2d"].clone()
    ys, xs = torch.where(seg[0] > 0)
    for b in range(B):
        m = seg[b] > 0
        rows = torch.where(m.any(1))[0]
        cols = torch.where(m.any(0))[0]
        lo = v[b].min(0).values
        hi = v[b].max(0).values
        tgt_lo = torch.stack([cols.min(), rows.min()]).float()
        tgt_hi = torch.stack([cols.max(), rows.max()]).float()
        v[b] = (v[b] - lo) / (hi - lo) * (tgt_hi - tgt_lo) + tgt_lo
    cases.append(("mesh over the silhouette", v))
    # vertices spread evenly over the silhouette (a surface mesh of the person in the mask), 1 px of jitter
    g = torch.Generator().manual_seed(7)
    u = torch.empty_like(v)
    for b in range(B):
        ys, xs = torch.where(seg[b] > 0)
        pick = torch.randint(0, ys.numel(), (v.shape[1],), generator=g).to(ys.device)
        u[b, :, 0] = xs[pick].float()
        u[b, :, 1] = ys[pick].float()
    u += (torch.rand(u.shape, generator=g) - 0.5).to(u.device) * 2.0
    cases.append(("vertices spread evenly", u))
    if len(sys.argv) > 2:
        cases = [cases[int(sys.argv[2])]]
    for name, v2d in cases:
        for _ in range(2):
            val = hpe_amd.mesh_reprojection_loss(eng, seg, v2d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            val = hpe_amd.mesh_reprojection_loss(eng, seg, v2d)
        e1.record()
        torch.cuda.synchronize()
        ext = (v2d.amax((0, 1)) - v2d.amin((0, 1))).tolist()
        print("%-26s %8.3f ms/call   loss %.6e   mesh extent %.0f x %.0f px" % (name, e0.elapsed_time(e1) / 10, float(val), ext[0], ext[1]))


if __name__ == "__main__":
    main()
