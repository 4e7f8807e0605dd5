"""Single-frame / small-batch forward latency (eager, synchronous) for one value of an environment knob per process:
    HPE_SPLITK_SLABS=8 python tools/latency_knob_sweep.py            (prints one line: B = 1, 2, 4, 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import synthetic
from oracle import hmr_oracle as O  # mean-theta helper only (measurement tool)
eng = hpe_amd.HpeEngine(device=0, max_batch=8)
eng.load_smpl(synthetic.make_smpl_model()); eng.load_encoder(synthetic.make_encoder_params()); eng.load_regressor(synthetic.make_regressor_params())
eng.load_mean_theta(O.load_mean_param(synthetic.make_mean_params())); eng.finalize()
out = []
for B in (1, 2, 4, 8):
    img = torch.from_numpy(synthetic.make_images(B, seed=B)).cuda()
    run, outs = eng.make_forward_plan(B, graph=False)
    for _ in range(10): run(img)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); n = 100
        for _ in range(n):
            run(img); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    out.append("B=%d %.3f ms" % (B, best * 1e3))
print(" ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("HPE_")) or "defaults", "|", "  ".join(out))
