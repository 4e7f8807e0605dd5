"""Mesh-reprojection-loss kernels on REALISTIC meshes (SMPL at plausible thetas, projected with s ~ 0.8): grid/bitmap path vs
brute force (HPE_MESH_BRUTE=1).  The bench's random-init network collapses its meshes, which exercises only the fallback."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = hpe_amd.HpeEngine(device=0, max_batch=B); eng.load_smpl(synthetic.make_smpl_model()); eng.finalize()
th = synthetic.make_thetas(B, seed=1); th[:, 0] = 0.8; th[:, 1:3] *= 0.3
v2d = eng.smpl(torch.from_numpy(th).cuda(), want=("verts2d",))["verts2d"]
seg, _ = synthetic.make_lsp_targets(B, seed=2); seg = torch.from_numpy(seg[..., 0].copy()).cuda()
for _ in range(2): out = eng.mesh_loss(seg, v2d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = eng.mesh_loss(seg, v2d)
e1.record(); torch.cuda.synchronize()
print("B=%d brute=%s: %.3f ms per call, loss %.6f" % (B, os.environ.get("HPE_MESH_BRUTE", "0"), e0.elapsed_time(e1) / 10, float(out)))
