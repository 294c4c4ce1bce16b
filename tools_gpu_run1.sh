#!/bin/bash
# first GPU pass: kernel parity, end-to-end parity, smoke, short bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > gpurun_out/gpu.txt 2>&1
nproc >> gpurun_out/gpu.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/kern.log 2>&1
echo "kern exit $?" >> gpurun_out/kern.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity exit $?" >> gpurun_out/parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --workload cfg1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg1.log 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg3.log 2>&1
tail -5 gpurun_out/kern.log gpurun_out/parity.log gpurun_out/smoke.log gpurun_out/bench_cfg1.log gpurun_out/bench_cfg3.log
