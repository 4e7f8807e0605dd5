#!/bin/bash
# bench lines of the round for profiles/ (run on the GPU box from the repo root): bash tools/collect_round.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
HPE_POWER_TRACE=$OUT/power_trace_fp32.csv python bench.py --steps 20 --warmup 5 2>/dev/null > $OUT/bench_fp32.json
python bench.py --steps 30 --warmup 5 --batch 64 --cpu-sample 0 --sustain 0 2>/dev/null > $OUT/bench_fp32_b64.json
HPE_POWER_TRACE=$OUT/power_trace_bf16.csv python bench.py --steps 30 --warmup 5 --encoder-dtype bf16 --cpu-sample 16 2>/dev/null > $OUT/bench_bf16.json
python bench.py --steps 10 --warmup 3 --config5 --cpu-sample 0 2>/dev/null > $OUT/bench_config5.json
HPE_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 2>/dev/null > $OUT/bench_fp32_rccl_world1.json
HPE_STREAMS=1 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 2>/dev/null > $OUT/bench_fp32_streams1.json
python bench.py --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 --no-pipeline 2>/dev/null > $OUT/bench_fp32_no_pipeline.json
python bench.py --steps 30 --warmup 5 --cpu-sample 0 --sustain 0 --no-pipeline --encoder-dtype bf16 2>/dev/null > $OUT/bench_bf16_no_pipeline.json
HPE_FORCE_DIST=1 python bench.py --steps 30 --warmup 5 --cpu-sample 0 --sustain 0 --encoder-dtype bf16 2>/dev/null > $OUT/bench_bf16_rccl_world1.json
HPE_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --sustain 0 --config5 2>/dev/null > $OUT/bench_config5_rccl_world1.json
HPE_STEM_FUSED=0 HPE_DUAL=0 HPE_WIDE128_MIN_TILES=0 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 2>/dev/null > $OUT/bench_fp32_r1_structure.json
HPE_STEM_FUSED=0 HPE_DUAL=0 HPE_BF16_RULES=0 HPE_BF16_128_MIN_TILES=512 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 --encoder-dtype bf16 2>/dev/null > $OUT/bench_bf16_r1_structure.json
HPE_BENCH_LAYERS=1 HPE_CONCURRENT_TILES=1 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --sustain 0 2>$OUT/layers_fp32.txt > /dev/null
HPE_BENCH_LAYERS=1 HPE_CONCURRENT_TILES=1 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --sustain 0 --encoder-dtype bf16 2>$OUT/layers_bf16.txt > /dev/null
python tools/latency_bench.py 2>/dev/null | grep "B=" > $OUT/latency_small_batch.txt
for m in grid mfma; do echo "== HPE_MESH_A2B=$m"; HPE_MESH_A2B=$m python tools/mesh_loss_bench.py 2>/dev/null; done > $OUT/mesh_loss_search.txt
HPE_MESH_A2B=mfma python bench.py --steps 10 --warmup 3 --config5 --cpu-sample 0 --sustain 0 2>/dev/null > $OUT/bench_config5_full_search.json
for f in $OUT/bench_*.json; do echo $f; cut -c1-170 $f; done
