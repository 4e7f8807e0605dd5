#!/bin/bash
# A/B of the bf16 encoder with the 3x3 layers on the implicit GEMM vs the halo-resident kernel (plan option halo3 / HPE_HALO3): per-layer
# milliseconds of a serial pass and the pipelined step rate.  Usage: tools/halo3_ab.sh OUTDIR [masks...]
OUT=${1:-gpurun_out/halo3}; shift
MASKS=${@:-0 15}
mkdir -p $OUT
python tools/layer_times.py 256 5 --all -- off=dtype:bf16,halo3:0 on=dtype:bf16,halo3:15 > $OUT/layers_halo3.txt 2>&1 || exit 1
grep -E "branch2b|3x3|all conv|layer" $OUT/layers_halo3.txt
for m in $MASKS; do
  HPE_HALO3=$m python bench.py --encoder-dtype bf16 --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 --no-legs \
      > $OUT/bench_bf16_halo3_$m.json 2> $OUT/bench_bf16_halo3_$m.err || exit 1
  python - <<PY
import json
d = json.load(open("$OUT/bench_bf16_halo3_$m.json"))
print("HPE_HALO3=$m  %.1f img/s  %.4f ms/step  span %.4f ms  serial sum %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["serial"]["sum_of_53_launch_ms"]))
PY
done
