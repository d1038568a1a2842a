#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r03_pytest10.log 2>&1; tail -3 gpurun_out/r03_pytest10.log
step bash tools/dbg/prof_stats_args.sh r03_attn --pipelines 1 --steps 2 --warmup 1 --no-single-extra > gpurun_out/r03_prof_attn.log 2>&1; head -8 gpurun_out/r03_prof_attn.log
step python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-single-extra > gpurun_out/r03_b10.json 2> gpurun_out/r03_b10.err
python -c "import json,sys; j=json.loads(open('gpurun_out/r03_b10.json').read().strip().splitlines()[-1]); print('bench', round(j['value']), round(j['ms_per_step'],1))"
