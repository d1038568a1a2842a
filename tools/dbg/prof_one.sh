#!/bin/bash
# usage: prof_one.sh <kernel-substring> "<bench.py flags>" ...  -- average duration of one kernel under each set of bench flags
# (the A/B switches are bench.py flags that map to nh_set_option: --no-graphs, --no-ln-fusion; "" = the product configuration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
for cfg in "$@"; do
  rm -rf /tmp/p1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipelines 1 --decode-groups 1 --no-single-extra $cfg > /tmp/p1.log 2>&1
  f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1)
  echo "[$cfg] => $(grep "$K" $f | awk -F, '{printf "%s calls %s avg %.1f us\n", substr($1,1,40), $2, $4/1000}')"
done
