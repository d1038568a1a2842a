#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step python -m pytest tests/test_gpu_threads.py tests/test_gpu_kernels.py -q > gpurun_out/r03_pytest6.log 2>&1; tail -2 gpurun_out/r03_pytest6.log
for m in "" "--private-weights" "" "--private-weights"; do
  step python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-single-extra $m > gpurun_out/r03_b6.json 2> gpurun_out/r03_b6.err
  python -c "import json,sys; j=json.loads(open('gpurun_out/r03_b6.json').read().strip().splitlines()[-1]); print('[$m]', round(j['value']), round(j['ms_per_step'],1), j['config']['weight_sets_per_gpu'])"
done
