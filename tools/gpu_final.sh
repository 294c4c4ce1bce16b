#!/bin/bash
# The round's closing lease: full GPU suite, the evidence set (tools/gpu_evidence.sh secondary), and the extras
# (launch count, single-rank DDP line, cfg4 kernel stats + phase times, HBM-bound kernel table, parity report).  Outputs: gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
bash tools/gpu_full_tests.sh | tail -n 12
cd $R; bash tools/gpu_evidence.sh secondary
cd $R; bash tools/gpu_launch_count.sh | head -3 | cut -c1-160
cd $R
ICG_FORCE_DDP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_ddp1.log 2>&1
tail -n 1 gpurun_out/bench_ddp1.log | cut -c1-200
timeout 300 python tools/sg2_phase_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sg2_phase_times.txt; tail -n 4 gpurun_out/sg2_phase_times.txt
timeout 400 python tools/hbm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/hbm_bench.txt; grep -c "TB/s\|GB/s" gpurun_out/hbm_bench.txt
timeout 900 python tools/parity_report.py --stats > gpurun_out/parity_report.txt 2>&1; grep -c PARITY gpurun_out/parity_report.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o bench -- python $R/bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > $R/gpurun_out/prof/rocprof_cfg4.log 2>&1
cp /tmp/prof4/bench_kernel_stats.csv $R/gpurun_out/prof/bench_cfg4_fp16_kernel_stats.csv
cd $R; bash tools/gpu_step_trace.sh > /dev/null 2>&1; head -n 1 gpurun_out/step_trace.txt; grep "idle gaps" gpurun_out/step_trace.txt | cut -c1-160
cd $R; timeout 300 python tools/pgemm_bench.py > gpurun_out/pgemm_bench.txt 2>&1; grep "^sum" gpurun_out/pgemm_bench.txt
cd $R; timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_bench.txt; cat gpurun_out/attn_bench.txt | cut -c1-200
