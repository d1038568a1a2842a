#!/bin/bash
# usage: [PMC_GROUPS="A B C|D E F"] prof_gemm_pmc.sh <tag> [gbench args]
# Cache-side PMC passes of one gbench shape (counters only, one '|'-separated group per rocprofv3 run); appends the per-launch
# averages of the GEMM kernels to gpurun_out/<tag>_gemm_pmc.txt.  (A group of TA_* counters never finished on this pool.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
DEFAULT="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum|TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
IFS='|' read -ra GRPS <<< "${PMC_GROUPS:-$DEFAULT}"
i=0
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  rm -rf /tmp/gp_$i
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/gp_$i -- $R/tools/bin/gbench "$@" > /tmp/gp_$i.log 2>&1 || { echo "group $i ($grp) failed" | tee -a $R/gpurun_out/${TAG}_gemm_pmc.txt; tail -3 /tmp/gp_$i.log; continue; }
  f=$(find /tmp/gp_$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $R/gpurun_out/${TAG}_gemm_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'gemm' not in k: continue
    acc[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, ' '.join(f'{c}={sum(v)/len(v):.4g}(n={len(v)})' for c, v in d.items()))
PY
done
cat $R/gpurun_out/${TAG}_gemm_pmc.txt
