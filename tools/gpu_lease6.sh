#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/sg2_route_check.txt
timeout 600 python -m pytest tests/test_sg2_fused_gpu.py tests/test_stylegan2.py tests/test_decision_replay_gpu.py -m gpu -q > gpurun_out/l6_tests.log 2>&1
echo "tests exit $?"; tail -n 8 gpurun_out/l6_tests.log | cut -c1-250
cat gpurun_out/sg2_route_check.txt
ICG_PGEMM_MFMA32=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -k "plane or wino or conv2d" > gpurun_out/l6_mfma32_tests.log 2>&1
echo "mfma32 tests exit $?"; tail -n 6 gpurun_out/l6_mfma32_tests.log | cut -c1-250
( echo "## ICG_PGEMM_MFMA32=0 (v_mfma_f32_16x16x4_f32, production)"; ICG_PGEMM_MFMA32=0 timeout 300 python tools/pgemm_bench.py nn_only;
  echo "## ICG_PGEMM_MFMA32=1 (v_mfma_f32_32x32x2_f32 on the 128-column tile)"; ICG_PGEMM_MFMA32=1 timeout 300 python tools/pgemm_bench.py nn_only;
  echo "## ICG_PGEMM_MFMA32=0 again (box drift check)"; ICG_PGEMM_MFMA32=0 timeout 300 python tools/pgemm_bench.py nn_only ) > gpurun_out/pgemm_mfma32.txt 2>&1
cat gpurun_out/pgemm_mfma32.txt | cut -c1-150
