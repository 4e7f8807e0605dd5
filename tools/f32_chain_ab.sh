#!/bin/bash
# A/B of the fp32 step with / without the chained 1x1 launch on the identity blocks of stage 2 (plan option chain_fuse bit 8 / HPE_CHAIN=8)
python tools/layer_times.py 256 5 --all -- off= on=chain_fuse:8 2>/dev/null | grep -E "res2[abc]_branch2[abc]|all conv"
for m in 0 8 0 8; do
  HPE_CHAIN=$m python bench.py --steps 30 --warmup 5 --cpu-sample 0 --sustain 0 --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HPE_CHAIN=$m', d['value'], d['ms_per_step'], d['roofline']['serial']['sum_of_53_launch_ms'])"
done
