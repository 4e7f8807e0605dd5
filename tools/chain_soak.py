"""Soak of the chained launches (counted vmcnt waits): N encoder passes per (dtype, batch), every result bitwise equal to the first.
    python tools/chain_soak.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
enc = synthetic.make_encoder_params()
for dtype in ("fp32", "bf16"):
    e = hpe_amd.HpeEngine(device=0, max_batch=256, encoder_dtype=dtype)
    e.load_encoder(enc); e.finalize()
    for B in (256, 100, 37):
        img = torch.from_numpy(synthetic.make_images(B, seed=900 + B)).cuda()
        first = e.encoder(img).clone()
        torch.cuda.synchronize()
        bad = 0
        t0 = time.time()
        n = N if B == 256 else N // 2
        for i in range(n):
            f = e.encoder(img)
            if not torch.equal(f, first):
                bad += 1
        torch.cuda.synchronize()
        print("%s B=%3d: %d passes, %d differ from the first (%.1f s)" % (dtype, B, n, bad, time.time() - t0), flush=True)
    e.close()
