#!/bin/bash
# One parameterised GPU-lease script (replaces the per-call gpu_r2_* / gpu_r3_* scripts):  bash tools/gpu_call.sh TAG STEP [STEP ...]
# STEP = "tests:<seconds>:<pytest args>" | "py:<seconds>:<script and args>" | "bench:<seconds>:<bench.py args>" | "sh:<seconds>:<command>"
# Each step runs under its own `timeout`, logs to gpurun_out/<TAG>_<n>.log and prints its tail.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG="$1"; shift
n=0
for step in "$@"; do
  n=$((n + 1))
  kind="${step%%:*}"; rest="${step#*:}"; secs="${rest%%:*}"; cmd="${rest#*:}"
  log="gpurun_out/${TAG}_${n}.log"
  case "$kind" in
    tests) timeout "$secs" python -m pytest -m gpu -q -p no:cacheprovider $cmd > "$log" 2>&1 ;;
    py)    timeout "$secs" python $cmd > "$log" 2>&1 ;;
    bench) timeout "$secs" python bench.py $cmd > "$log" 2>&1 ;;
    sh)    timeout "$secs" bash -c "$cmd" > "$log" 2>&1 ;;
    *) echo "unknown step kind $kind"; continue ;;
  esac
  echo "== step $n [$kind] rc=$? : $cmd"
  grep -vE "amdgpu.ids|^$" "$log" | tail -${GPU_CALL_TAIL:-25} | cut -c1-400
done
