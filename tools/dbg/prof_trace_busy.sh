#!/bin/bash
# usage: prof_trace_busy.sh <tag> [bench args]  -- kernel trace of bench.py; prints how much of the timed region the GPU had
# 0, 1, 2, ... kernels resident (union of [start, end) intervals), per-stream busy time, into gpurun_out/<tag>_busy.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf /tmp/ktrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-single-extra "$@" > /tmp/ktrace.log 2>&1 || { tail -5 /tmp/ktrace.log; exit 1; }
f=$(find /tmp/ktrace -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $R/gpurun_out/${TAG}_busy.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ev.append((s, 1, r)); ev.append((e, -1, r))
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
# a 2 s window around the median encoder-GEMM launch: inside the 30-step timed region (4.5 s), away from build and extras
import statistics
mid = int(statistics.median(int(r['Start_Timestamp']) for r in rows if 'gemm256' in r['Kernel_Name']))
lo, hi = mid - 10**9, mid + 10**9
ev.sort(key=lambda x: (x[0], x[1]))
depth = 0; last = None; hist = collections.Counter()
for t, d, r in ev:
    if last is not None and t > lo and last < hi:
        a, b = max(last, lo), min(t, hi)
        if b > a: hist[depth] += b - a
    depth += d; last = t
tot = sum(hist.values())
print(f"window {tot/1e6:.1f} ms of trace; kernels resident at once -> share of time")
for k in sorted(hist): print(f"  {k}: {100*hist[k]/tot:5.1f} %")
# time by kernel family inside the window (sum of durations, overlapping counted separately)
fam = collections.Counter()
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e <= lo or s >= hi: continue
    n = r['Kernel_Name']
    key = 'gemm256' if 'gemm256' in n else 'enc_attn' if 'enc_attn' in n else 'dec_attn' if 'dec_attn' in n else 'skinny' if 'skinny' in n else 'logit_step' if 'logit_step' in n else 'layernorm' if 'layernorm' in n else 'other'
    fam[key] += min(e, hi) - max(s, lo)
for k, v in fam.most_common(): print(f"  {k:12s} {v/1e6:8.1f} ms summed ({100*v/tot:5.1f} % of the window)")
PY
cat $R/gpurun_out/${TAG}_busy.txt; tail -c 300 /tmp/ktrace.log
