#!/bin/bash
# A/B of the fp32 step with Winograd F(4x4,3x3) on a set of map sizes (HPE_WINO_F4 bit mask: 1 = 7x7, 2 = 14x14, 4 = 28x28, 8 = 56x56).
# Usage: tools/f4_ab.sh OUTDIR [masks...]
OUT=${1:-gpurun_out/f4}; shift
MASKS=${@:-0 3 0 3}
mkdir -p $OUT
i=0
for m in $MASKS; do
  i=$((i+1))
  HPE_WINO_F4=$m python bench.py --steps 20 --warmup 5 --cpu-sample 16 --sustain 0 --no-legs > $OUT/bench_f4_${m}_$i.json 2> $OUT/bench_f4_${m}_$i.err || { tail -5 $OUT/bench_f4_${m}_$i.err; exit 1; }
  python - <<PY
import json
d = json.load(open("$OUT/bench_f4_${m}_$i.json"))
print("HPE_WINO_F4=$m  %.1f img/s  %.4f ms/step  span %.4f ms  serial sum %.4f ms  parity worst %.3g kp2d_rms %.3g" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["serial"]["sum_of_53_launch_ms"], d["parity"]["worst_gated"], d["parity"]["kp2d_rel_rms"]))
PY
done
