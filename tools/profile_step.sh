#!/bin/bash
# rocprofv3 passes over one bench step (run on the GPU box from the repo root): kernel trace + separate PMC passes.
# usage: bash tools/profile_step.sh <outdir under gpurun_out> [extra bench.py arguments, e.g. --encoder-dtype bf16 / --config5]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
EXTRA="$@"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HPE_STREAMS=1
export HPE_CONCURRENT_TILES=1  # the tiles the default (chunk-stream) run uses, on serial launches
B="$GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --no-roofline --sustain 0 $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $B --steps 1 --warmup 1 > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 $B --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- python3 $B --steps 1 --warmup 0 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma -o m -- python3 $B --steps 1 --warmup 0 > $OUT/mfma.log 2>&1
unset HPE_STREAMS HPE_CONCURRENT_TILES
rocprofv3 --kernel-trace --stats -d $OUT/kt3 -o kt3 -- python3 $B --steps 5 --warmup 2 > $OUT/kt3.log 2>&1
find $OUT -name "*.db" | sort
