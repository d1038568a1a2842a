#!/bin/bash
# one gpurun call: full GPU suite, shared- vs private-weights bench, GEMM tail experiment, cumask teardown order 0
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03_pytest2.log 2>&1
tail -3 gpurun_out/r03_pytest2.log
step python bench.py --steps 12 --warmup 3 > gpurun_out/r03_bench_shared.json 2> gpurun_out/r03_bench_shared.err
step python bench.py --steps 12 --warmup 3 --private-weights --no-cpu-baseline > gpurun_out/r03_bench_private.json 2> gpurun_out/r03_bench_private.err
step python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_shared2.json 2> gpurun_out/r03_bench_shared2.err
for M in 48000 47872 52224 65536; do echo "== M=$M"; step tools/bin/gbench $M; done > gpurun_out/r03_gbench_tail.txt 2>&1
step timeout -k 5 120 tools/bin/cumask destroy 0 > gpurun_out/r03_cumask_destroy0.txt 2>&1
tail -4 gpurun_out/r03_cumask_destroy0.txt
