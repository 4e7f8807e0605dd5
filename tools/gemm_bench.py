"""Micro-benchmark of the implicit-GEMM kernel in dense mode (steady-state MFMA efficiency, no conv addressing).
usage: python tools/gemm_bench.py  [M N K tile]..."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hpe_amd
from hpe_amd import _lib, synthetic

eng = hpe_amd.HpeEngine(device=0, max_batch=8)
eng.load_regressor(synthetic.make_regressor_params())
eng.load_mean_theta(np.zeros(85, np.float32))
eng.finalize()
cases = [(32768, 256, 2304, 0), (65536, 256, 2304, 0), (50176, 256, 2304, 0), (65536, 64, 576, 1), (65536, 256, 64, 0), (65536, 256, 1024, 0),
         (32768, 256, 2304, 3), (16384, 256, 2304, 2)]
if len(sys.argv) > 4:
    v = list(map(int, sys.argv[1:]))
    cases = [tuple(v[i:i + 4]) for i in range(0, len(v), 4)]
for M, N, K, tile in cases:
    x = torch.randn(M, K, device="cuda")
    wr = ((N + 127) // 128) * 128
    w = torch.randn(wr, K, device="cuda")
    y = torch.empty(M, N, device="cuda")
    def run():
        _lib.check(eng.lib.hpe_debug_gemm(eng._h, x.data_ptr(), w.data_ptr(), M, N, K, wr, tile, None, 0, y.data_ptr(), None))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    clk = ""
    if os.environ.get("HPE_CLK"):
        bm0, bn0 = [(128, 128), (128, 64), (64, 64), (64, 128), (128, 128), (128, 64), (256, 128)][tile]
        nw = ((M + bm0 - 1) // bm0) * ((N + bn0 - 1) // bn0)
        dbg = torch.zeros(2 * nw, dtype=torch.int64, device="cuda")
        _lib.check(eng.lib.hpe_debug_set_dbg(eng._h, dbg.data_ptr()))
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        _lib.check(eng.lib.hpe_debug_set_dbg(eng._h, None))
        d = dbg.cpu().numpy().reshape(-1, 2).astype(np.float64)
        ghz = d[:, 0] / d[:, 1] * 0.1
        clk = " clk(GHz) med=%.3f min=%.3f max=%.3f loop_us=%.1f" % (np.median(ghz), ghz.min(), ghz.max(), np.median(d[:, 1]) / 100.0)
    ref = x[:256] @ w[:N].T
    err = float((y[:256] - ref).abs().max() / ref.abs().max())
    bm, bn = [(128, 128), (128, 64), (64, 64), (64, 128), (128, 128), (128, 64), (256, 128)][tile]
    wgs = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    print("M=%6d N=%4d K=%5d tile=%dx%d wgs=%5d  %8.3f ms  %6.1f TF  err=%.1e%s" % (M, N, K, bm, bn, wgs, ms, 2.0 * M * N * K / ms / 1e9, err, clk))
