#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step bash tools/dbg/prof_stats_args.sh r03_g2 --pipelines 1 --decode-groups 2 --steps 2 --warmup 2 --no-single-extra > gpurun_out/r03_prof_g2.log 2>&1; cat gpurun_out/r03_prof_g2.log | tail -18
