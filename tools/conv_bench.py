"""Time single conv layers (loaded synthetic weights) through hpe_debug_conv at a given batch; HPE_TILE_WIDE selects the tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hpe_amd
from hpe_amd import synthetic, resnet_spec
B = int(sys.argv[1]); names = sys.argv[2:]
eng = hpe_amd.HpeEngine(device=0, max_batch=B)
eng.load_encoder(synthetic.make_encoder_params()); eng.finalize()
for name in names:
    idx = resnet_spec.CONV_INDEX[name]; s = resnet_spec.CONV_SPECS[idx]
    x = torch.randn(B, s.hin, s.hin, s.cin, device="cuda")
    for _ in range(3): eng.debug_conv(idx, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30; e0.record()
    for _ in range(n): eng.debug_conv(idx, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * s.kh * s.kw * s.cin * s.cout * s.hout * s.hout * B
    print("tile=%s %-16s B=%d M=%d: %.3f ms %.1f TF" % (os.environ.get("HPE_TILE_WIDE", "auto"), name, B, B * s.hout * s.hout, ms, fl / ms / 1e9))
