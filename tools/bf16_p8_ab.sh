#!/bin/bash
# A/B of the bf16 encoder with and without the 256 x 256 phase-interleaved kernel (HPE_BF16_P8 bit mask), per-layer times from
# bench.py's level-2 event pass (serial, unchunked B = 256) and the pipelined step rate.  Usage: tools/bf16_p8_ab.sh OUTDIR [masks...]
OUT=${1:-gpurun_out/p8}; shift
MASKS=${@:-0 1 3 7}
mkdir -p $OUT
for m in $MASKS; do
  HPE_BF16_P8=$m HPE_BENCH_LAYERS=1 python bench.py --encoder-dtype bf16 --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 --no-legs \
      > $OUT/bench_bf16_p8_$m.json 2> $OUT/layers_bf16_p8_$m.txt || exit 1
  python - <<PY
import json
d = json.load(open("$OUT/bench_bf16_p8_$m.json"))
print("HPE_BF16_P8=$m  %.1f img/s  %.4f ms/step  span %.4f ms  serial sum %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["serial"]["sum_of_53_launch_ms"]))
PY
done
