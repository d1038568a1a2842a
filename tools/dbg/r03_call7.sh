#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step tools/bin/gstamps > gpurun_out/r03_gstamps2.txt 2>&1
for g in gbench_gm2 gbench gbench_gm8 gbench_gm16; do echo "== $g"; step tools/bin/$g; done > gpurun_out/r03_gbench_gm.txt 2>&1
step bash tools/dbg/prof_stats_args.sh r03_b3 --workload longform20 --rank-share 0/8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_prof_b3.log 2>&1
cat gpurun_out/r03_gstamps2.txt | grep "per tile"
