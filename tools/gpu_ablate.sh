#!/bin/bash
mkdir -p gpurun_out
for v in LOAD_FIXED ADDR_ONLY SKIP_GLOAD; do
  echo "== $v" >> gpurun_out/ablate.log
  ICG_LIB=$PWD/tools/libdbg_$v.so timeout 300 python tools/conv_bench.py "D.b3.conv2" "D.b0.conv2" 2>&1 | grep -v amdgpu >> gpurun_out/ablate.log
done
echo "== baseline" >> gpurun_out/ablate.log
timeout 300 python tools/conv_bench.py "D.b3.conv2" "D.b0.conv2" 2>&1 | grep -v amdgpu >> gpurun_out/ablate.log
cat gpurun_out/ablate.log
