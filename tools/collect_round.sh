#!/bin/bash
# bench lines of the round for profiles/ (run on the GPU box from the repo root): bash tools/collect_round.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1
mkdir -p $OUT
Q="--cpu-sample 0 --sustain 0 --no-legs"
# the driver's command: headline + every single-GPU config + graph / from_host legs in one line
HPE_POWER_TRACE=$OUT/power_trace_fp32.csv python bench.py --steps 20 --warmup 5 2>$OUT/bench_fp32.err > $OUT/bench_fp32.json
python bench.py --steps 30 --warmup 5 --batch 64 $Q 2>/dev/null > $OUT/bench_fp32_b64.json
HPE_POWER_TRACE=$OUT/power_trace_bf16.csv python bench.py --steps 30 --warmup 5 --encoder-dtype bf16 --cpu-sample 16 --no-legs 2>/dev/null > $OUT/bench_bf16.json
python bench.py --steps 10 --warmup 3 --config5 --cpu-sample 0 2>/dev/null > $OUT/bench_config5.json
python bench.py --steps 10 --warmup 3 --config5 --regressor survey --cpu-sample 0 --sustain 0 2>/dev/null > $OUT/bench_config5_survey_regressor.json
# round 4: the encoders without the chained 1x1 launches (one launch per layer), same box
HPE_CHAIN=0 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null > $OUT/bench_fp32_chain_off.json
HPE_CHAIN=0 python bench.py --steps 30 --warmup 5 --encoder-dtype bf16 $Q 2>/dev/null > $OUT/bench_bf16_chain_off.json
# round 4: the bf16 encoder with its 3x3 layers on the implicit GEMM (halo3 = 0), same box
HPE_HALO3=0 python bench.py --steps 30 --warmup 5 --encoder-dtype bf16 $Q 2>/dev/null > $OUT/bench_bf16_halo3_off.json
HPE_HALO3=0 HPE_CHAIN=0 HPE_BF16_W8_MIN_TILES=0 python bench.py --steps 30 --warmup 5 --encoder-dtype bf16 $Q 2>/dev/null > $OUT/bench_bf16_r3_plan.json
HPE_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null > $OUT/bench_fp32_rccl_world1.json
HPE_FORCE_DIST=1 python bench.py --steps 30 --warmup 5 $Q --encoder-dtype bf16 2>/dev/null > $OUT/bench_bf16_rccl_world1.json
HPE_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 $Q --config5 2>/dev/null > $OUT/bench_config5_rccl_world1.json
HPE_STREAMS=1 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null > $OUT/bench_fp32_streams1.json
python bench.py --steps 20 --warmup 5 $Q --no-pipeline 2>/dev/null > $OUT/bench_fp32_no_pipeline.json
python bench.py --steps 20 --warmup 5 $Q --graph 2>/dev/null > $OUT/bench_fp32_graph.json
python bench.py --steps 20 --warmup 5 $Q --from-host 2>/dev/null > $OUT/bench_fp32_from_host.json
# the round-2 plan (F(2x2) Winograd only) and the round-1 structure on the same box
HPE_WINO_F4=0 HPE_CHAIN=0 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null > $OUT/bench_fp32_r2_plan.json
HPE_WINO_F4=0 HPE_CHAIN=0 HPE_STEM_FUSED=0 HPE_DUAL=0 HPE_WIDE128_MIN_TILES=0 python bench.py --steps 20 --warmup 5 $Q --no-pipeline 2>/dev/null > $OUT/bench_fp32_r1_structure.json
HPE_BENCH_LAYERS=1 HPE_CONCURRENT_TILES=1 python bench.py --steps 5 --warmup 2 $Q 2>$OUT/layers_fp32.txt > /dev/null
HPE_BENCH_LAYERS=1 HPE_CONCURRENT_TILES=1 python bench.py --steps 5 --warmup 2 $Q --encoder-dtype bf16 2>$OUT/layers_bf16.txt > /dev/null
HPE_CHAIN=0 HPE_BENCH_LAYERS=1 HPE_CONCURRENT_TILES=1 python bench.py --steps 5 --warmup 2 $Q --encoder-dtype bf16 2>$OUT/layers_bf16_chain_off.txt > /dev/null
HPE_HALO3=0 HPE_BENCH_LAYERS=1 HPE_CONCURRENT_TILES=1 python bench.py --steps 5 --warmup 2 $Q --encoder-dtype bf16 2>$OUT/layers_bf16_halo3_off.txt > /dev/null
python tools/latency_bench.py 2>/dev/null | grep "B=" > $OUT/latency_small_batch.txt
python tools/latency_breakdown.py 2>/dev/null | grep "B=" > $OUT/latency_breakdown.txt
for m in grid mfma; do echo "== HPE_MESH_A2B=$m"; HPE_MESH_A2B=$m python tools/mesh_loss_bench.py 2>/dev/null; done > $OUT/mesh_loss_search.txt
for f in $OUT/bench_*.json; do echo $f; cut -c1-170 $f; done
