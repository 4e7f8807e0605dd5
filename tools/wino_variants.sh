#!/bin/bash
# Winograd 3x3 variants of the fp32 encoder: per-kernel time (kernel trace) and MfmaUtil (PMC pass) of ONE bench step, HPE_STREAMS=1.
# usage (GPU box, repo root): bash tools/wino_variants.sh <outdir under gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HPE_STREAMS=1
B="$GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --no-roofline --sustain 0 --no-pipeline"
run() {  # name, env assignments...
    name=$1; shift
    rm -rf /tmp/wv_kt /tmp/wv_pm
    env "$@" true
    ( export "$@"; rocprofv3 --kernel-trace -d /tmp/wv_kt -o kt -- python3 $B --steps 1 --warmup 1 > /tmp/wv.log 2>&1;
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/wv_pm -o m -- python3 $B --steps 1 --warmup 0 > /tmp/wv.log 2>&1 )
    echo "== $name ($*)" >> $OUT/wino_variants.txt
    python3 $GRAFT_REPO_ROOT/tools/wino_variant_row.py /tmp/wv_kt/kt_results.db /tmp/wv_pm/m_results.db >> $OUT/wino_variants.txt
}
: > $OUT/wino_variants.txt
run default HPE_WINO_FUSED=1
run fused_on_14x14 HPE_WINO_FUSED_MINHW=14
run blocked_V_everywhere HPE_WINO_FUSED=0
run stream_K HPE_WINO_STREAMK=1
run direct_only HPE_WINO_MINC=0
cat $OUT/wino_variants.txt
