#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
for w in G D; do
  rm -rf /tmp/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o attn -- python $R/tools/attn_block_prof.py $w 2>&1 | grep -v amdgpu.ids | tail -n 2
  cp /tmp/prof_$w/attn_kernel_stats.csv $R/gpurun_out/attn_${w}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof_$w/attn_kernel_stats.csv")))
for r in rows[:24]:
    print("%8.3f ms/iter %5.1f calls/iter %8.1f us  %s" % (float(r["TotalDurationNs"])/6e6, int(r["Calls"])/6, float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
done
