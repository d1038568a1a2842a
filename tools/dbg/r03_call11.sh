#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step python -m pytest tests/test_gpu_parity.py -x -q -k "layers or full_depth" > gpurun_out/r03_pytest11.log 2>&1; tail -2 gpurun_out/r03_pytest11.log
step bash tools/dbg/prof_stats_args.sh r03_attn --pipelines 1 --steps 2 --warmup 1 --no-single-extra > gpurun_out/r03_prof_attn.log 2>&1; grep "enc_attn" gpurun_out/r03_prof_attn.log
