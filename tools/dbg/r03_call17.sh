#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
for cfg in "--decode-groups 2 --pipelines 3" "--decode-groups 2 --pipelines 4" "--decode-groups 2 --pipelines 3 --batch 48" "--decode-groups 2 --pipelines 3"; do
  step python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-single-extra $cfg > gpurun_out/r03_b17.json 2> gpurun_out/r03_b17.err
  python -c "import json; j=json.loads(open('gpurun_out/r03_b17.json').read().strip().splitlines()[-1]); print('[$cfg]', round(j['value']), round(j['ms_per_step'],1))" || tail -3 gpurun_out/r03_b17.err
done
