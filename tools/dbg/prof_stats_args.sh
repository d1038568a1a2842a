#!/bin/bash
# usage: prof_stats_args.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; shift
rm -rf /tmp/prof_$T
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -- python3 $R/bench.py --no-cpu-baseline "$@" > /tmp/prof_$T.log 2>&1
f=$(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${T}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-72s calls %6s avg %10.1f us  tot %8.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
