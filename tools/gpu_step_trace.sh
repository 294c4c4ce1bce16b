#!/bin/bash
# Per-LAUNCH view of one steady-state cfg3 step (round 6): kernel trace of a 4-step process, the launches of the LAST step listed in
# execution order with duration and grid size, and the 60 longest ones -> gpurun_out/step_trace.txt.  (The stats csv of tools/gpu_prof.sh
# averages a kernel over layers of very different sizes: `icg_gemm_kernel<1,1,3,2>` is 0.5 .. 5 ms.)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-uninstrumented-leg --no-kernel-timer --init N02 $* > $R/gpurun_out/step_trace_run.log 2>&1
python - <<'PY' > $R/gpurun_out/step_trace.txt
import csv, glob, re
f = glob.glob("/tmp/st/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the step boundary: adam_kernel launches close a phase; a step = D phase + G phase -> find the last two groups of adam launches
idx = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
groups = []
for i in idx:
    if groups and i - groups[-1][-1] <= 3:
        groups[-1].append(i)
    else:
        groups.append([i])
end = groups[-1][-1]
start = groups[-3][-1] + 1 if len(groups) >= 3 else 0
step = rows[start:end + 1]
t0 = int(step[0]["Start_Timestamp"])
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print("last step of the process: %d launches, %.2f ms of kernels, %.2f ms wall" % (len(step), tot / 1e6, (int(step[-1]["End_Timestamp"]) - t0) / 1e6))
def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:90]
print("---- the 60 longest launches")
for r in sorted(step, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:60]:
    print("%9.1f us  at %8.2f ms  grid %-18s wg %-5s %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e6,
          "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), short(r["Kernel_Name"])))
print("---- execution order (launches >= 100 us)")
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= 100:
        print("%9.1f us  at %8.2f ms  grid %-12s %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")), short(r["Kernel_Name"])))
PY
head -70 $R/gpurun_out/step_trace.txt | cut -c1-170
python - <<'PY' >> $R/gpurun_out/step_trace.txt
import csv, glob, re
f = glob.glob("/tmp/st/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
groups = []
for i in idx:
    if groups and i - groups[-1][-1] <= 3:
        groups[-1].append(i)
    else:
        groups.append([i])
end = groups[-1][-1]
start = groups[-3][-1] + 1 if len(groups) >= 3 else 0
step = rows[start:end + 1]
t0 = int(step[0]["Start_Timestamp"])
gaps = []
for a, b in zip(step, step[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a, b))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("(anonymous namespace)::", ""))[:70]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in step]
print("---- launches by duration: " + "; ".join("%s us: %d launches, %.2f ms" % (lab, sum(1 for d in durs if lo <= d < hi), sum(d for d in durs if lo <= d < hi) / 1e3)
                                             for lab, lo, hi in (("< 10", 0, 10), ("10 - 50", 10, 50), ("50 - 200", 50, 200), ("200 - 1000", 200, 1000), (">= 1000", 1000, 1e12))))
fam = {}
for r, d in zip(step, durs):
    k = short(r["Kernel_Name"])
    fam.setdefault(k, [0, 0.0])
    fam[k][0] += 1
    fam[k][1] += d
print("---- per kernel in this step (>= 0.5 ms in total)")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    if t >= 500:
        print("%8.2f ms  %4d launches  %s" % (t / 1e3, c, k))
print("---- idle gaps between consecutive launches: total %.2f ms; > 20 us: %d gaps = %.2f ms; the 25 largest" % (
    sum(max(g[0], 0) for g in gaps) / 1e6, sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
for g, a, b in sorted(gaps, key=lambda t: -t[0])[:25]:
    print("%8.1f us at %8.2f ms  after %-50s before %s" % (g / 1e3, (int(a["End_Timestamp"]) - t0) / 1e6, short(a["Kernel_Name"]), short(b["Kernel_Name"])))
PY
tail -n 28 $R/gpurun_out/step_trace.txt | cut -c1-200
