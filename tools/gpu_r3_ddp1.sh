#!/bin/bash
# bench.py's N > 1 code path on one GPU: RCCL process group of size 1 (ICG_FORCE_DDP=1) and the torchrun launch line of the driver with one rank
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
ICG_FORCE_DDP=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_ddp1_force.log 2>&1; echo "force-ddp rc=$?"
tail -1 gpurun_out/r3_ddp1_force.log | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_ddp1_torchrun.log 2>&1; echo "torchrun rc=$?"
tail -1 gpurun_out/r3_ddp1_torchrun.log | cut -c1-200
python - <<'PY'
import json
for f in ("gpurun_out/r3_ddp1_force.log", "gpurun_out/r3_ddp1_torchrun.log"):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f, d["ms_per_step"], d["n_gpus"], d["config"].get("comm"))
PY
