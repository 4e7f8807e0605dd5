#!/bin/bash
# timing ablations of w4_gemm_kernel (results wrong): per-layer ms at B=256 for HPE_W4_ABL = 0, 1 (no V DMA), 2 (no U DMA), 3 (no DMA), 4 (no MFMA), 8 (no frag reads), 12
for a in 0 1 2 3 4 8 12; do
  echo "== HPE_W4_ABL=$a"; HPE_W4_ABL=$a python tools/layer_times.py 256 3 -- f4=wino_f4:15 2>/dev/null | grep "res2b_branch2b\|res3b_branch2b\|res4b_branch2b\|res5b_branch2b"
done
