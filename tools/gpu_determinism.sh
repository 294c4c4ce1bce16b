#!/bin/bash
# N processes of tools/determinism_cfg3.py (or, with `sg2` as second argument, tools/determinism_sg2.py) with per-parameter gradient digests;
# prints the lines that are not the same in all of them.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/det
export PYTHONDONTWRITEBYTECODE=1
N=${1:-8}; shift
rm -f gpurun_out/det/run_*.txt
for i in $(seq 1 $N); do
  if [ "$1" = "sg2" ]; then
    timeout 300 python tools/determinism_sg2.py 2>&1 | grep "^init\|^step\|^grad" > gpurun_out/det/run_$i.txt
  else
    timeout 300 python tools/determinism_cfg3.py 3 --grads $* 2>&1 | grep "^init\|^step\|^grad" > gpurun_out/det/run_$i.txt
  fi
done
python - <<PY
import glob, collections
runs = [open(f).read().splitlines() for f in sorted(glob.glob("gpurun_out/det/run_*.txt"))]
n = min(len(r) for r in runs)
bad = 0
for k in range(n):
    vals = collections.Counter(r[k] for r in runs)
    if len(vals) > 1:
        bad += 1
        if bad <= 40:
            print("DIFFERS:", " | ".join("%dx %s" % (c, v[:110]) for v, c in vals.most_common()))
print("%d processes, %d lines each, %d lines differ" % (len(runs), n, bad))
PY
