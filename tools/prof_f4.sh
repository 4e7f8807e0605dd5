#!/bin/bash
# rocprofv3 kernel trace of one fp32 bench run with HPE_WINO_F4=$1 (serial steps, 1 chunk stream so that kernel times add up)
M=${1:-7}; OUT=${2:-gpurun_out/prof_f4_$M}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HPE_WINO_F4=$M HPE_STREAMS=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/raw -o f4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 --sustain 0 --no-legs --no-pipeline > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%-90s calls %6s  total %9.3f ms  avg %8.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
