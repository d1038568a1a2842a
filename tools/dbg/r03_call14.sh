#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step python -m pytest tests/test_gpu_parity.py -x -q -k "decoded_together or batch_64" > gpurun_out/r03_pytest14.log 2>&1; tail -3 gpurun_out/r03_pytest14.log
for cfg in "--decode-groups 1 --pipelines 3" "--decode-groups 2 --pipelines 3" "--decode-groups 3 --pipelines 2" "--decode-groups 3 --pipelines 3" "--decode-groups 2 --pipelines 2"; do
  step python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-single-extra $cfg > gpurun_out/r03_b14.json 2> gpurun_out/r03_b14.err
  python -c "import json; j=json.loads(open('gpurun_out/r03_b14.json').read().strip().splitlines()[-1]); print('[$cfg]', round(j['value']), round(j['ms_per_step'],1))" || tail -3 gpurun_out/r03_b14.err
done
