"""Small-batch latency breakdown: encoder alone, regressor + SMPL tail alone (hpe_tail), whole forward; eager, sync after each call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import synthetic

smpl, enc, reg = synthetic.make_smpl_model(), synthetic.make_encoder_params(), synthetic.make_regressor_params()
mean = np.zeros((1, 85), np.float32); mean[0, 0] = 0.9; mean[0, 3] = np.pi
eng = hpe_amd.HpeEngine(device=0, max_batch=64)
eng.load_smpl(smpl); eng.load_encoder(enc); eng.load_regressor(reg); eng.load_mean_theta(mean); eng.finalize()


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in (1, 2, 4, 8, 16):
    img = torch.from_numpy(synthetic.make_images(B, seed=B)).cuda()
    run, _ = eng.make_forward_plan(B)
    feat = eng.encoder(img)
    th = eng.forward(img)[0]["theta"]
    t_all = timeit(lambda: run(img))
    t_enc = timeit(lambda: eng.encoder(img))
    t_tail = timeit(lambda: eng.tail(feat))
    t_smpl = timeit(lambda: eng.smpl(th, want=("verts", "joints", "kp2d", "J_transformed")))
    t_reg = timeit(lambda: eng.regress_stage(feat))
    print("B=%2d  forward %.3f ms   encoder %.3f   tail %.3f   (one SMPL call %.3f, one regress_stage incl. feature projection %.3f)" % (B, t_all, t_enc, t_tail, t_smpl, t_reg))
