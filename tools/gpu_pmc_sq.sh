#!/bin/bash
# MFMA-pipe utilisation of the GEMM kernels inside the bench step (one PMC pass, kernel trace only) -> gpurun_out/pmc_sq.txt
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_sq -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-uninstrumented-leg --init N02 > $R/gpurun_out/pmc_sq.log 2>&1
python $R/tools/pmc_sq.py /tmp/pmc_sq > $R/gpurun_out/pmc_sq.txt
cat $R/gpurun_out/pmc_sq.txt | cut -c1-160
