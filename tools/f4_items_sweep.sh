#!/bin/bash
# which batch sizes should take F(4x4)?  serial-pass conv sums at B = 48 .. 128 for F4 off / on with different work-item thresholds
for B in 48 64 96 128; do
  for thr in 100000 64 128 192 256; do
    HPE_WINO4_MIN_ITEMS=$thr python tools/layer_times.py $B 3 -- x= 2>/dev/null | grep "3x3 layers\|all conv" | tr '\n' ' ' | sed "s/^/B=$B min_items=$thr  /"; echo
  done
done
