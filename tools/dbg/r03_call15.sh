#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r03_pytest_full.log 2>&1; tail -3 gpurun_out/r03_pytest_full.log
step python bench.py --steps 20 --warmup 5 > gpurun_out/r03_default_bench_line_full.json 2> gpurun_out/r03_default_bench.err
python -c "import json; j=json.loads(open('gpurun_out/r03_default_bench_line_full.json').read().strip().splitlines()[-1]); print('bench', round(j['value']), round(j['ms_per_step'],1), j['roofline']['achieved'], j['roofline']['time_weighted_frac'], j['cpu_baseline']['value'])"
step bash tools/dbg/prof_stats_default.sh r03_default
step bash tools/dbg/prof_stats_args.sh r03_one_batch --pipelines 1 --decode-groups 1 --steps 3 --warmup 1 --no-single-extra > gpurun_out/r03_prof_one_batch.log 2>&1
step bash tools/dbg/prof_pmc.sh r03
step bash tools/dbg/prof_pmc_sq.sh r03
