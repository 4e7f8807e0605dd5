"""Phases of ONE workgroup of nn_a2b_grid_kernel (diagnostics build: HPE_EXTRA_FLAGS=-DHPE_A2B_STAMPS), 100 MHz wall-clock stamps:
    HPE_EXTRA_FLAGS=-DHPE_A2B_STAMPS python tools/a2b_phases.py [case]      (cases as in tools/mesh_loss_bench.py: 0-2 stages, 3 stretched, 4 spread)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import synthetic, build
build.build()
assert "-DHPE_A2B_STAMPS" in build.built_flags(), "needs the diagnostics build"
B = 256
class Cfg(object):
    img_size, num_stage, batch_size, data_format = 224, 3, B, "NHWC"
    checkpoint_dir = smpl_model_path = None
    encoder_dtype = "fp32"
pred = hpe_amd.Predictor(Cfg(), smpl_model=synthetic.make_smpl_model(), mean_params=synthetic.make_mean_params(),
                         encoder_params=synthetic.make_encoder_params(), regressor_params=synthetic.make_regressor_params(variant="bounded"))
eng = pred.engine
images = torch.from_numpy(synthetic.make_images(B, seed=1000)).cuda()
seg_np, _ = synthetic.make_lsp_targets(B, seed=2000)
seg = torch.from_numpy(seg_np[..., 0].copy()).cuda()
outs = eng.forward(images, all_stages=True, want=eng.DEFAULT_OUTPUTS + ("verts2d",))
torch.cuda.synchronize()
names = ["start -> bitmap staged", "histogram", "prefix scan", "scatter (sort)", "tiles (this wave)", "wait for the other waves"]
for st in range(3):
    cnt = torch.zeros(64, dtype=torch.int64, device="cuda")
    eng.set_loss_counter(cnt)
    for _ in range(3):
        eng.mesh_loss(seg, outs[st]["verts2d"])
    torch.cuda.synchronize()
    eng.set_loss_counter(None)
    s = cnt.cpu().numpy()[8:15]
    d = np.diff(s) / 100.0  # us
    print("bounded regressor, stage %d: " % (st + 1) + "; ".join("%s %.1f us" % (n, v) for n, v in zip(names, d)) + "; workgroup total %.1f us" % ((s[6] - s[0]) / 100.0))
