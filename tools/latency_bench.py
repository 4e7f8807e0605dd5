"""Small-batch latency of the full forward: eager launches vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import synthetic
from oracle import hmr_oracle as O
eng = hpe_amd.HpeEngine(device=0, max_batch=64)
eng.load_smpl(synthetic.make_smpl_model()); eng.load_encoder(synthetic.make_encoder_params()); eng.load_regressor(synthetic.make_regressor_params())
eng.load_mean_theta(O.load_mean_param(synthetic.make_mean_params())); eng.finalize()
for B in (1, 8, 16, 32, 64):
    img = torch.from_numpy(synthetic.make_images(B, seed=B)).cuda()
    for graph in (False, True):
        run, outs = eng.make_forward_plan(B, graph=graph)
        for _ in range(5): run(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 50
        for _ in range(n):
            run(img); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("B=%2d graph=%d: %.3f ms per forward (sync each), %.1f img/s" % (B, graph, dt * 1e3, B / dt))
