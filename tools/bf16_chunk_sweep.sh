#!/bin/bash
# bf16 step rate vs batch-chunk size / chunk streams (does a smaller working set per chunk keep the chained layers' tensors in the 256 MB
# Infinity Cache?).  Usage: tools/bf16_chunk_sweep.sh OUTDIR
OUT=${1:-gpurun_out/chunks}; mkdir -p $OUT
run() {
  env "$@" python bench.py --encoder-dtype bf16 --steps 30 --warmup 5 --cpu-sample 0 --sustain 0 --no-legs 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %9.1f img/s  %.4f ms/step  span %.4f' % ('$*', d['value'], d['ms_per_step'], d['roofline']['launch_ms']))"
}
run HPE_STREAMS=2
run HPE_STREAMS=1
run HPE_STREAMS=2 HPE_CHUNK=64
run HPE_STREAMS=1 HPE_CHUNK=64
run HPE_STREAMS=2 HPE_CHUNK=32
run HPE_STREAMS=1 HPE_CHUNK=32
run HPE_STREAMS=3 HPE_CHUNK=64
run HPE_STREAMS=2 HPE_CHUNK=86
run HPE_STREAMS=2
