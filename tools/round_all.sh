#!/bin/bash
# everything a round commits under profiles/: GPU test log, rocprofv3 passes (fp32 / bf16 / config 5), bench lines.  Usage: tools/round_all.sh <dir under gpurun_out>
D=$1
mkdir -p gpurun_out/$D
python -m pytest tests -m gpu -x -q > gpurun_out/$D/gpu_tests_full.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$D/gpu_tests_full.log
tail -3 gpurun_out/$D/gpu_tests_full.log
bash tools/profile_step.sh $D/prof_fp32 && bash tools/profile_step.sh $D/prof_bf16 --encoder-dtype bf16 && bash tools/profile_step.sh $D/prof_c5 --config5 && HPE_CHAIN=0 bash tools/profile_step.sh $D/prof_bf16nochain --encoder-dtype bf16 && echo "profiles done"
bash tools/collect_round.sh $D > gpurun_out/$D/collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/$D/collect.log
