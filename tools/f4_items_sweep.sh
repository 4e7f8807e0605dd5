#!/bin/bash
# which launches should take F(4x4), and which of them the 32-cout variant?  serial-pass conv sums (tools/layer_times.py) per batch for
# HPE_WINO4_MIN_ITEMS (smallest F(4x4) launch, in 32-tile x 32-cout workgroups) x HPE_WINO4_N32 (64-cout workgroup count below which the
# 32-cout kernel runs; 0 = never)
for B in ${@:-32 64 128 256}; do
  for cfg in "1000000 0" "128 0" "128 320" "128 600" "256 320" "512 320"; do
    set -- $cfg
    HPE_WINO4_MIN_ITEMS=$1 HPE_WINO4_N32=$2 python tools/layer_times.py $B 3 -- x= 2>/dev/null | grep "3x3 layers\|all conv" | tr '\n' ' ' | sed "s/^/B=$B min_items=$1 n32_below=$2  /"; echo
  done
done
