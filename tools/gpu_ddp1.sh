#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_stylegan_ops.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_sg.log 2>&1; tail -n 3 gpurun_out/tests_sg.log
ICG_FORCE_DDP=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_ddp1.log 2>&1
tail -n 3 gpurun_out/bench_ddp1.log | cut -c1-400
ICG_FORCE_DDP=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --sync-bn > gpurun_out/bench_syncbn1.log 2>&1
tail -n 1 gpurun_out/bench_syncbn1.log | cut -c1-300
