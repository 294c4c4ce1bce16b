#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES pass (csv) per kernel:
MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (4 SIMDs per CU).

    python tools/pmc_sq.py <pass dir>
"""
import csv
import glob
import os
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
                cnt[k] += 1
rows = []
for k, c in tot.items():
    busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
    if busy > 0 and any(t in k for t in ("icg_gemm", "icg_pgemm", "icg_pconv", "icg_fwino_kernel", "icg_attn", "thin_", "narrow_")):
        rows.append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy), busy, cnt[k], k))
for util, busy, n, k in sorted(rows, key=lambda r: -r[1]):
    print(f"{util:6.3f} MFMA-pipe utilisation  {n:5d} launches  {k}")
