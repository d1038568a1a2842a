#!/bin/bash
# usage: prof_stats_default.sh <tag>  -- rocprofv3 kernel-trace stats of the DEFAULT bench command (three batches in flight)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python3 $R/bench.py --no-cpu-baseline > /tmp/prof_$1.log 2>&1 || { tail -5 /tmp/prof_$1.log; exit 1; }
f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/$1_kernel_stats.csv
grep '"metric"' /tmp/prof_$1.log | tail -1 > $R/gpurun_out/$1_bench_line.json
python3 - "$f" $R/gpurun_out/$1_bench_line.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if 'gemm256' in r['Name']]
calls = sum(int(r['Calls']) for r in g); tot = sum(float(r['TotalDurationNs']) for r in g)
print(f"gemm256 launches {calls}, weighted average {tot/calls/1e3:.1f} us")
d = json.load(open(sys.argv[2]))
print("bench.py roofline.avg_launch_ms", d['roofline']['avg_launch_ms'], "value", d['value'])
PY
