#!/bin/bash
# first half of tools/round_all.sh (one gpurun call): GPU test log + the rocprofv3 passes.  Usage: tools/round_part_a.sh <dir under gpurun_out>
D=$1
mkdir -p gpurun_out/$D
python -m pytest tests -m gpu -x -q > gpurun_out/$D/gpu_tests_full.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$D/gpu_tests_full.log
tail -3 gpurun_out/$D/gpu_tests_full.log
bash tools/profile_step.sh $D/prof_fp32 && bash tools/profile_step.sh $D/prof_bf16 --encoder-dtype bf16 && bash tools/profile_step.sh $D/prof_c5 --config5 && HPE_CHAIN=0 bash tools/profile_step.sh $D/prof_bf16nochain --encoder-dtype bf16 && echo "profiles done"
