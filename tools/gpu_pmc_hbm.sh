#!/bin/bash
# HBM traffic per kernel launch of the bench command: two PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass) +
# a copy-kernel calibration of known size.  Output: gpurun_out/hbm_traffic.json (copy to profiles/).
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=${GRAFT_REPO_ROOT:-$PWD}
export ICG_PMC_COMMIT=${ICG_PMC_COMMIT:-$(cat $R/tools/.head_commit 2>/dev/null || echo unknown)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/calib.py <<'PY'
import torch
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
PY
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-uninstrumented-leg --init N02 > $R/gpurun_out/pmc_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/cal_$C -o run -- python /tmp/calib.py > $R/gpurun_out/cal_$C.log 2>&1
done
python $R/tools/pmc_hbm.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/cal_FETCH_SIZE /tmp/cal_WRITE_SIZE 1073741824 > $R/gpurun_out/hbm_traffic.json
head -c 1500 $R/gpurun_out/hbm_traffic.json; tail -c 600 $R/gpurun_out/hbm_traffic.json
