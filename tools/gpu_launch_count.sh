#!/bin/bash
# steady-state launches and kernel time per cfg3 step: the difference of two kernel traces (3 and 6 steps in the process), so that the
# one-off launches (parameter upload, Adam state, orthogonal init, first-call spectral-norm path) cancel -> gpurun_out/launch_count.txt
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
for n in 2 5; do
  rm -rf /tmp/lc_$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lc_$n -o run -- python $R/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-uninstrumented-leg --init N02 > $R/gpurun_out/lc_$n.log 2>&1
done
python - <<'PY' > $R/gpurun_out/launch_count.txt
import csv
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = load("/tmp/lc_2/run_kernel_stats.csv"), load("/tmp/lc_5/run_kernel_stats.csv")
steps = 3
rows = []
for n, (c, t) in b.items():
    c0, t0 = a.get(n, (0, 0.0))
    if c != c0:
        rows.append((n, (c - c0) / steps, (t - t0) / steps / 1e6))
print("steady state per cfg3 step (difference of a 6-step and a 3-step process, --init N02): %.1f launches, %.2f ms of kernels" % (
    sum(r[1] for r in rows), sum(r[2] for r in rows)))
at = [r for r in rows if "at::" in r[0] or "rocclr" in r[0]]
print("of which ATen / rocclr: %.1f launches, %.2f ms" % (sum(r[1] for r in at), sum(r[2] for r in at)))
for n, c, t in sorted(rows, key=lambda r: -r[1]):
    print("%7.1f  %8.3f ms  %s" % (c, t, n[:150]))
PY
head -40 $R/gpurun_out/launch_count.txt | cut -c1-200
