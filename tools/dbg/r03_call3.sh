#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
(cd _r02 && step python bench.py --steps 12 --warmup 3 --no-cpu-baseline) > gpurun_out/r03_ab_r02tree.json 2> gpurun_out/r03_ab_r02tree.err
step python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r03_ab_r03tree.json 2> gpurun_out/r03_ab_r03tree.err
(cd _r02 && step python bench.py --steps 12 --warmup 3 --no-cpu-baseline) > gpurun_out/r03_ab_r02tree_b.json 2> gpurun_out/r03_ab_r02tree_b.err
step tools/bin/gstamps > gpurun_out/r03_gstamps.txt 2>&1
step python bench.py --workload varlen --steps 3 --no-cpu-baseline > gpurun_out/r03_varlen.json 2> gpurun_out/r03_varlen.err
step python bench.py --workload longform20 --rank-share 0/8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_share_longform20.json 2> gpurun_out/r03_share_longform20.err
step python bench.py --workload longform20 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_full_longform20.json 2> gpurun_out/r03_full_longform20.err
step python bench.py --workload lv3b64 --rank-share 0/8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_share_lv3b64.json 2> gpurun_out/r03_share_lv3b64.err
for f in r03_ab_r02tree r03_ab_r03tree r03_ab_r02tree_b; do python -c "import json,sys; j=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', round(j['value']), round(j['ms_per_step'],1), round(j['extra'].get('xrt_one_batch_at_a_time_per_gpu',0)))"; done
