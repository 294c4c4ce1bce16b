#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs of the same command, csv output) into
HBM bytes per launch per kernel:  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (counters are in KiB; the factor 2
on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md §HBM, checked here on a copy kernel of known size).

    python tools/pmc_hbm.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> [calib_fetch_dir calib_write_dir calib_bytes]
"""
import csv
import glob
import sys as _sys
import os as _os
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return tot, cnt, files


def main():
    fdir, wdir = sys.argv[1:3]
    ft, fc, ff = load(fdir, "FETCH_SIZE")
    wt, wc, wf = load(wdir, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 2 "
                     "--warmup 1 --no-cpu-baseline --init N02`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                     "MI355X_MICROARCH.md HBM section; tools/pmc_hbm.py",
           "steps_profiled": int(os.environ.get("ICG_PMC_STEPS", "3")),       # --steps 2 --warmup 1
           "commit": os.environ.get("ICG_PMC_COMMIT", "unknown"),
           "csrc_sha256": __import__("bench").csrc_sha256(),      # bench.py prints traffic_stale when the kernel sources differ
           "kernels": {}}
    for k in sorted(ft, key=lambda k: -ft[k]):
        if fc[k] == 0:
            continue
        rd = 2.0 * ft[k] * 1024 / fc[k]
        wr = wt.get(k, 0.0) * 1024 / max(wc.get(k, 0), 1)
        out["kernels"][k] = {"launches": fc[k], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                             "hbm_bytes_per_launch": round(rd + wr)}
    if len(sys.argv) >= 6:
        cf, cc, _ = load(sys.argv[3], "FETCH_SIZE")
        cw, cwc, _ = load(sys.argv[4], "WRITE_SIZE")
        expect = float(sys.argv[5])
        k = max(cf, key=lambda k: cf[k])
        out["calibration"] = {"kernel": k, "expected_read_bytes": expect, "expected_write_bytes": expect,
                              "raw_FETCH_SIZE_KiB_per_launch": cf[k] / cc[k],
                              "raw_WRITE_SIZE_KiB_per_launch": cw.get(k, 0.0) / max(cwc.get(k, 0), 1),
                              "read_ratio_after_x2": 2.0 * cf[k] * 1024 / cc[k] / expect,
                              "write_ratio": cw.get(k, 0.0) * 1024 / max(cwc.get(k, 0), 1) / expect}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
