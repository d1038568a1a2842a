#!/bin/bash
# usage: prof_pmc.sh <tag>  -- two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counters only, with --kernel-trace) of a short
# one-batch-at-a-time bench run, summarised into gpurun_out/<tag>_pmc_hbm_traffic.json by tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipelines 1 --decode-groups 1 --no-single-extra > /tmp/pmc_$C.log 2>&1 || { tail -5 /tmp/pmc_$C.log; exit 1; }
  echo "pass $C done"
done
python3 $R/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/$1_pmc_hbm_traffic.json
