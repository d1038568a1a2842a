#!/bin/bash
# usage: prof_stats.sh <tag>  -- rocprofv3 kernel-trace stats of a short one-batch-at-a-time bench run -> gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipelines 1 --decode-groups 1 > /tmp/prof_$1.log 2>&1
f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/$1_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg %10.1f us  tot %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
