#!/bin/bash
# A/B of the bf16 encoder with and without the chained 1x1 launches (plan option chain_fuse / HPE_CHAIN): per-layer milliseconds of a
# serial pass (level-2 events, one process) and the pipelined step rate.  Usage: tools/chain_ab.sh OUTDIR [masks...]
OUT=${1:-gpurun_out/chain}; shift
MASKS=${@:-0 7}
mkdir -p $OUT
python tools/layer_times.py 256 5 --all -- off=dtype:bf16,chain_fuse:0 on=dtype:bf16,chain_fuse:7 s4=dtype:bf16,chain_fuse:23 > $OUT/layers_chain.txt 2>&1 || exit 1
for m in $MASKS; do
  HPE_CHAIN=$m python bench.py --encoder-dtype bf16 --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 --no-legs \
      > $OUT/bench_bf16_chain_$m.json 2> $OUT/bench_bf16_chain_$m.err || exit 1
  python - <<PY
import json
d = json.load(open("$OUT/bench_bf16_chain_$m.json"))
print("HPE_CHAIN=$m  %.1f img/s  %.4f ms/step  span %.4f ms  serial sum %.4f ms  parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["serial"]["sum_of_53_launch_ms"], d.get("parity", {}).get("pass")))
PY
done
