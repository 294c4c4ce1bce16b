#!/usr/bin/env python
"""How far apart are the UNMODIFIED reference's own gradients with and without fp16 blocks, at cfg4's real network?
Reads the two committed goldens (tests/golden/stylegan2_cfg4_r256.npz: fp32; ..._fp16.npz: num_fp16_res=4, conv_clamp=256,
phase_gain 1024) and prints, per loss phase, the difference of the gradient fingerprints (64 samples per tensor) in units of the
tensor rms.  This is the noise floor any second fp16 implementation with other rounding points sits on: the tolerances of
tests/test_stylegan2.py for `cfg4_r256_fp16` are set against it.  CPU only, no reference needed."""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
a, b = np.load(os.path.join(G, "stylegan2_cfg4_r256.npz")), np.load(os.path.join(G, "stylegan2_cfg4_r256_fp16.npz"))
for ph in ("Gmain", "Greg", "Dmain", "Dreg"):
    names = json.loads(str(a[f"grad/{ph}/names"]))
    assert names == json.loads(str(b[f"grad/{ph}/names"]))
    sa, sb = a[f"grad/{ph}/samp"], b[f"grad/{ph}/samp"] / 1024.0
    rows = []
    for i, n in enumerate(names):
        rms = float(np.sqrt((sa[i] ** 2).mean()))
        if rms < 1e-12:
            continue
        d = np.abs(sb[i] - sa[i])
        rows.append((float(d.max()) / rms, float(np.sqrt((d ** 2).mean())) / rms, n))
    rows.sort(reverse=True)
    print("%s: reference with fp16 blocks vs reference in fp32 -- max / rms of the difference over the 64 samples, in units of the tensor rms" % ph)
    for r in rows[:4]:
        print("    %.3f  %.3f  %s" % r)
    print("    median over %d tensors: max %.3f, rms %.3f" % (len(rows), np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
