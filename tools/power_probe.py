"""Run one GEMM configuration back to back for a few seconds and sample rocm-smi power/clock (diagnostics)."""
import os, subprocess, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import _lib, synthetic
eng = hpe_amd.HpeEngine(device=0, max_batch=8)
eng.load_regressor(synthetic.make_regressor_params()); eng.load_mean_theta(np.zeros(85, np.float32)); eng.finalize()
M, N, K, tile = [int(v) for v in sys.argv[1:5]]
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 6.0
x = torch.randn(M, K, device="cuda"); wr = ((N + 127) // 128) * 128; w = torch.randn(wr, K, device="cuda"); y = torch.empty(M, N, device="cuda")
def run():
    _lib.check(eng.lib.hpe_debug_gemm(eng._h, x.data_ptr(), w.data_ptr(), M, N, K, wr, tile, None, 0, y.data_ptr(), None))
for _ in range(5): run()
torch.cuda.synchronize()
samples = []
stop = False
def poll():
    while not stop:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = [l.split(":")[-1].strip() for l in o.splitlines() if "Package Power" in l]
        c = [l.split("(")[-1].strip(")") for l in o.splitlines() if "sclk" in l]
        samples.append((p[0] if p else "?", c[0] if c else "?"))
        time.sleep(0.7)
th = threading.Thread(target=poll); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50): run()
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1) / n
print("SCHED=%s STAGE=%s M=%d N=%d K=%d tile=%d: %.3f ms %.1f TF; power/clock samples (after 2 s): %s" % (
    os.environ.get("HPE_SCHED", "-"), os.environ.get("HPE_STAGE", "dma"), M, N, K, tile, ms, 2.0 * M * N * K / ms / 1e9, samples[3:]))
