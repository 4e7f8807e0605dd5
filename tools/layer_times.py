#!/usr/bin/env python3
"""Per-layer milliseconds of the 3x3 layers (or all conv layers) of one serial encoder pass at batch B for a set of plan-option
variants, measured in ONE process with the library's level-2 events (hpe_get_conv_timings); median of R passes.

    python tools/layer_times.py [B] [R] -- name=opt:val,opt:val ...      e.g.  base= f4=wino_f4:3 f4all=wino_f4:15
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hpe_amd  # noqa: E402
from hpe_amd import resnet_spec, synthetic  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if a != "--"]
    nums = [a for a in args if a.isdigit()]
    B = int(nums[0]) if nums else 256
    R = int(nums[1]) if len(nums) > 1 else 5
    variants = [a for a in args if "=" in a] or ["base="]
    only3 = "--all" not in args
    enc = synthetic.make_encoder_params()
    img = torch.from_numpy(synthetic.make_images(B, seed=5)).cuda()
    cols = {}
    for v in variants:
        name, spec = v.split("=", 1)
        opts = {}
        dtype = "fp32"
        for kv in filter(None, spec.split(",")):
            k, val = kv.split(":")
            if k == "dtype":
                dtype = val
            else:
                opts[k] = int(val)
        e = hpe_amd.HpeEngine(device=0, max_batch=B, encoder_dtype=dtype, **opts)
        e.load_encoder(enc)
        e.finalize()
        e.encoder(img)
        e.enable_timing(2)
        runs = []
        for _ in range(R):
            e.encoder(img)
            torch.cuda.synchronize()
            runs.append(e.conv_timings())
        cols[name] = np.median(np.array(runs), axis=0)
        e.close()
    specs = resnet_spec.CONV_SPECS
    print("%-18s" % ("layer (B=%d)" % B) + "".join("%12s" % n for n in cols))
    tot = {n: 0.0 for n in cols}
    tot3 = {n: 0.0 for n in cols}
    for i, s in enumerate(specs):
        for n in cols:
            tot[n] += cols[n][i]
            if s.kh == 3:
                tot3[n] += cols[n][i]
        if only3 and s.kh != 3:
            continue
        print("%-18s" % s.name + "".join("%12.4f" % cols[n][i] for n in cols))
    print("%-18s" % "3x3 layers" + "".join("%12.4f" % tot3[n] for n in cols))
    print("%-18s" % "all conv layers" + "".join("%12.4f" % tot[n] for n in cols))


if __name__ == "__main__":
    main()
