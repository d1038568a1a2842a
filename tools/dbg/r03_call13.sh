#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
for cfg in "--batch 32 --pipelines 3" "--batch 64 --pipelines 2" "--batch 64 --pipelines 3" "--batch 64 --pipelines 1" "--batch 32 --pipelines 4"; do
  step python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-single-extra $cfg > gpurun_out/r03_b13.json 2> gpurun_out/r03_b13.err
  python -c "import json; j=json.loads(open('gpurun_out/r03_b13.json').read().strip().splitlines()[-1]); print('[$cfg]', round(j['value']), round(j['ms_per_step'],1), j['phases_ms'])"
done
