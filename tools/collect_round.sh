#!/bin/bash
# bench lines of the round for profiles/ (run on the GPU box from the repo root): bash tools/collect_round.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
python bench.py --steps 20 --warmup 3 2>/dev/null > $OUT/bench_fp32.json
python bench.py --steps 30 --warmup 3 --batch 64 --cpu-sample 0 2>/dev/null > $OUT/bench_fp32_b64.json
python bench.py --steps 30 --warmup 3 --encoder-dtype bf16 --cpu-sample 16 2>/dev/null > $OUT/bench_bf16.json
python bench.py --steps 10 --warmup 2 --config5 --cpu-sample 0 2>/dev/null > $OUT/bench_config5.json
HPE_STREAMS=1 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null > $OUT/bench_fp32_serial.json
HPE_WINO_MINC=0 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null > $OUT/bench_fp32_direct_only.json
python tools/latency_bench.py 2>/dev/null | grep "B=" > $OUT/latency_small_batch.txt
for f in $OUT/bench_*.json; do echo $f; cut -c1-170 $f; done
