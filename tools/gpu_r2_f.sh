#!/bin/bash
# round 2, call F: suite after the ReLU-backward epilogue fusion, cfg3 bench with and without it
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/tests_gpu.log | tail -35
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg3.log 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fuse-relu-backward > gpurun_out/bench_cfg3_nofuse.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "cfg3_nofuse"):
    f = "gpurun_out/bench_%s.log" % w
    ok = False
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); ok = True
            r = d.get("roofline") or {}
            print("BENCH", w, d["ms_per_step"], d["value"], json.dumps({a: r.get(a) for a in ("kernel", "achieved", "frac", "traffic")}), json.dumps(r.get("step")))
    if not ok:
        print("BENCH", w, "NO JSON LINE"); print(open(f).read()[-2500:])
PY
