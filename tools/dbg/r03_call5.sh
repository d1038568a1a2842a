#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
for m in "" first single noprio sdfirst pad1 pad2 pad3; do
  NORMA_HIP_STREAMS=$m step python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-single-extra > gpurun_out/r03_streams_$m.json 2> gpurun_out/r03_streams_$m.err
  python -c "import json,sys; j=json.loads(open('gpurun_out/r03_streams_$m.json').read().strip().splitlines()[-1]); print('streams=[$m]', round(j['value']), round(j['ms_per_step'],1))"
done
