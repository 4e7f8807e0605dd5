"""Kernel timeline of single-frame forwards: run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/latency_trace.py [B]`,
then `python tools/latency_trace.py --parse DIR` prints the median forward: every kernel's duration and the idle gap in front of it."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(d):
    import csv
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: r[1])
    starts = [i for i, r in enumerate(rows) if "stem_fused" in r[0]]
    fw = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    fw = fw[len(fw) // 2:]
    spans = sorted((x[-1][2] - x[0][1], i) for i, x in enumerate(fw))
    x = fw[spans[len(spans) // 2][1]]
    print("median forward: %d kernels, first start -> last end %.1f us, sum of kernel times %.1f us" % (len(x), (x[-1][2] - x[0][1]) / 1e3, sum(e - s for _n, s, e in x) / 1e3))
    prev = x[0][1]
    for n, s, e in x:
        print("%8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, (s - prev) / 1e3, n[:110]))
        prev = max(prev, e)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        return parse(sys.argv[2])
    import numpy as np
    import torch
    import hpe_amd
    from hpe_amd import synthetic
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    eng = hpe_amd.HpeEngine(device=0, max_batch=max(B, 8))
    eng.load_smpl(synthetic.make_smpl_model()); eng.load_encoder(synthetic.make_encoder_params()); eng.load_regressor(synthetic.make_regressor_params())
    mean = np.zeros((1, 85), np.float32); mean[0, 0] = 0.9; mean[0, 3] = np.pi
    eng.load_mean_theta(mean); eng.finalize()
    img = torch.from_numpy(synthetic.make_images(B, seed=1)).cuda()
    run, _ = eng.make_forward_plan(B)
    for _ in range(40):
        run(img); torch.cuda.synchronize()


if __name__ == "__main__":
    main()
