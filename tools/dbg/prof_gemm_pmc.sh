#!/bin/bash
# usage: prof_gemm_pmc.sh <tag> [gbench args]  -- cache-side PMC passes of one gbench shape (counters only, one group per run)
# (a fifth group of TA_* counters never finished on this pool and was dropped)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --list-avail > $R/gpurun_out/${TAG}_avail.txt 2>&1
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rm -rf /tmp/gp_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/gp_$i -- $R/tools/bin/gbench "$@" > /tmp/gp_$i.log 2>&1 || { echo "group $i ($grp) failed"; tail -3 /tmp/gp_$i.log; continue; }
  f=$(find /tmp/gp_$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$grp" >> $R/gpurun_out/${TAG}_gemm_pmc.txt <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'gemm' not in k: continue
    acc[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, ' '.join(f'{c}={sum(v)/len(v):.4g}(n={len(v)})' for c, v in d.items()))
PY
done
cat $R/gpurun_out/${TAG}_gemm_pmc.txt
