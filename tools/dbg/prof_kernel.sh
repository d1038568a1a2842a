#!/bin/bash
# usage: prof_kernel.sh <kernel-substring> [bench args]  -- average duration of the matching kernels (rocprofv3 stats, one batch at a time)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
rm -rf /tmp/p1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipelines 1 --decode-groups 1 --no-single-extra "$@" > /tmp/p1.log 2>&1
f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$K" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]: print("%s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
