#!/bin/bash
# usage: prof_pmc_sq.sh <tag>  -- one rocprofv3 PMC pass of SQ counters (counters only, with --kernel-trace) -> gpurun_out/<tag>_pmc_sq.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --pipelines 1 --decode-groups 1 --no-single-extra > /tmp/pmc_sq.log 2>&1 || { tail -5 /tmp/pmc_sq.log; exit 1; }
python3 $R/tools/pmc_sq_summary.py /tmp/pmc_sq $R/gpurun_out/$1_pmc_sq.json
