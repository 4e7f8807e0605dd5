#!/bin/bash
# timing ablations of w4_gemm_kernel (results wrong): per-layer ms at B=256 for HPE_W4_ABL = 0, 1 (no V DMA), 2 (no U DMA), 3 (no DMA), 4 (no MFMA), 8 (no frag reads), 12
# needs the diagnostics build: HPE_W4_ABL is read only under -DHPE_ABLATION (build.py rebuilds when the flags differ from the library on disk)
export HPE_EXTRA_FLAGS="-DHPE_ABLATION"
python - <<'PY' || exit 1
from hpe_amd import build
build.build()
assert "-DHPE_ABLATION" in build.built_flags(), "library was not built with -DHPE_ABLATION: the rows below would be baseline timings"
PY
for a in 0 1 2 3 4 8 12; do
  echo "== HPE_W4_ABL=$a"; HPE_W4_ABL=$a python tools/layer_times.py 256 3 -- f4=wino_f4:15 2>/dev/null | grep "res2b_branch2b\|res3b_branch2b\|res4b_branch2b\|res5b_branch2b"
done
