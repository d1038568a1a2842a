#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
for t in _mid _r02 .; do
  (cd $t && step python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-single-extra) > gpurun_out/r03_bisect_$(basename $(cd $t; pwd)).json 2> gpurun_out/r03_bisect_$(basename $(cd $t; pwd)).err
done
for f in gpurun_out/r03_bisect_*.json; do python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(j['value']), round(j['ms_per_step'],1))"; done
