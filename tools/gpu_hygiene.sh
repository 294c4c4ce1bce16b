#!/bin/bash
# Round-6 hygiene lease: the whole -m gpu suite under ICG_POISON=nan (every torch.empty-family CUDA allocation pre-filled with NaN / 0x7f) and
# under ICG_DOUBLE_RUN=1 (every C-ABI call twice with differently poisoned outputs, bit-equal), then process-to-process determinism of the
# cfg3 step (10 processes) and of the four StyleGAN2 phases (8 processes).  Outputs: gpurun_out/hygiene_*.log
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
ICG_POISON=nan timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/hygiene_poison_nan.log 2>&1
echo "ICG_POISON=nan: $(tail -n 1 gpurun_out/hygiene_poison_nan.log | cut -c1-120)"
ICG_DOUBLE_RUN=1 timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/hygiene_double_run.log 2>&1
echo "ICG_DOUBLE_RUN=1: $(grep 'passed\|failed' gpurun_out/hygiene_double_run.log | tail -n 1 | cut -c1-120)"
grep "ICG_DOUBLE_RUN:" gpurun_out/hygiene_double_run.log | cut -c1-200
bash tools/gpu_determinism.sh 10 > gpurun_out/hygiene_determinism_cfg3.log 2>&1; tail -n 1 gpurun_out/hygiene_determinism_cfg3.log
bash tools/gpu_determinism.sh 8 sg2 > gpurun_out/hygiene_determinism_sg2.log 2>&1; tail -n 1 gpurun_out/hygiene_determinism_sg2.log
