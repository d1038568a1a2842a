#!/bin/bash
cd $GRAFT_REPO_ROOT
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return 0; }
step timeout -k 10 300 tools/bin/gemm_check 6 > gpurun_out/r03_gemm_check8.txt 2>&1; tail -2 gpurun_out/r03_gemm_check8.txt
step python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -x -q > gpurun_out/r03_pytest8.log 2>&1; tail -2 gpurun_out/r03_pytest8.log
step tools/bin/gbench > gpurun_out/r03_gbench8.txt 2>&1; cat gpurun_out/r03_gbench8.txt
step tools/bin/gstamps > gpurun_out/r03_gstamps8.txt 2>&1; grep "per tile" gpurun_out/r03_gstamps8.txt
