#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_audio.py tests/test_gpu_sampling.py tests/test_gpu_threads.py -x -q > gpurun_out/r03_pytest12.log 2>&1; tail -3 gpurun_out/r03_pytest12.log
step python bench.py --workload longform20 --rank-share 0/8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_share_longform20_b.json 2> gpurun_out/r03_share_longform20_b.err
python -c "import json; j=json.loads(open('gpurun_out/r03_share_longform20_b.json').read().strip().splitlines()[-1]); print('longform20 0/8', round(j['ms_per_step'],1), j['phases_ms'])"
step python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r03_b12.json 2> gpurun_out/r03_b12.err
python -c "import json; j=json.loads(open('gpurun_out/r03_b12.json').read().strip().splitlines()[-1]); print('bench', round(j['value']), round(j['ms_per_step'],1), j['phases_ms'], j['extra'].get('xrt_one_batch_at_a_time_per_gpu'))"
