#!/usr/bin/env python
"""Attribution of the gradient outliers of the real-width parity cases to flipped discrete decisions (tools only).

  python tools/mask_attribution.py masks CASE...   (CPU, minutes)  tests/decision_replay.py::make_masks -> tools/_masks/CASE.npz
                                                   (--golden: -> tests/golden/decisions_CASE.npz, the committed fixture of
                                                   tests/test_decision_replay_gpu.py)
  python tools/mask_attribution.py run CASE...     (GPU)  the HIP train step twice: as is, and with the fp64 forward's ReLU signs and
                                                   max-pool winners imposed (tests/decision_replay.py); per gradient tensor the rms /
                                                   max distance from the fp64 reference's samples (tests/golden/*_f64.npz) in both
                                                   runs, beside the fp32 reference's own."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from tests.decision_replay import GOLDEN_DIR, Nudger, hip_step, make_masks


def run(case):
    import tests.helpers as H
    g32 = H.load_golden(case)
    g64 = np.load(os.path.join(H.GOLDEN_DIR, "biggan_%s_f64.npz" % case), allow_pickle=False)
    plain, _, _ = hip_step(case)
    fixed, census, pool_census = hip_step(case, Nudger)
    total = sum(int(np.prod(s)) for _, s, _, _ in census)
    flips = sum(c[2] for c in census)
    print("== %s: %d ReLU prologues, %d elements, %d sign flips against the fp64 forward (%.2e of the elements), %d left after the nudge" % (
        case, len(census), total, flips, flips / total, sum(c[3] for c in census)))
    for name, shape, fl, left in sorted(census, key=lambda c: -c[2])[:4]:
        print("     %-28s %-22s flips %6d (%.1e)" % (name, "x".join(map(str, shape)), fl, fl / np.prod(shape)))
    pw = sum(int(np.prod(s)) // 4 for s, _, _ in pool_census)
    print("   %d max-pool inputs, %d windows, %d with another winner than the fp64 forward (%.2e), %d left after the nudge" % (
        len(pool_census), pw, sum(c[1] for c in pool_census), sum(c[1] for c in pool_census) / max(pw, 1), sum(c[2] for c in pool_census)))
    rows, noise = [], 0
    top = max(np.sqrt((g64[pf + "samp"] ** 2).sum(axis=1) / np.maximum((g64[pf + "samp"] != 0).sum(axis=1), 1)).max()
              for pf in ("step1/G_grad/", "step1/D_grad/"))
    for tag in "GD":
        prefix = "step1/%s_grad/" % tag
        names = json.loads(str(g32[prefix + "names"]))
        for i, n in enumerate(names):
            if (tag, n) not in plain:
                continue
            r32, r64 = g32[prefix + "samp"][i], g64[prefix + "samp"][i]
            k = int(np.count_nonzero(r64)) or 1
            rms = np.sqrt((r64 ** 2).sum() / k)
            if rms < 1e-6 * top:            # the whole gradient is rounding noise (mathematically zero: a bias feeding a BatchNorm)
                noise += 1
                continue
            st = lambda e: (np.sqrt((e ** 2).sum() / k) / rms, np.abs(e).max() / rms)
            rows.append((tag + " " + n, k) + st(plain[(tag, n)] - r64) + st(fixed[(tag, n)] - r64) + st(r32 - r64))
    rows.sort(key=lambda r: -r[3])
    print("   distance from the fp64 reference in units of the tensor rms: rms / max of the samples")
    print("   %-42s %6s  %-19s  %-19s  %-19s" % ("tensor (12 largest HIP max of %d; %d noise tensors left out)" % (len(rows), noise), "n", "HIP",
                                                 "HIP, fp64 decisions", "fp32 reference"))
    for r in rows[:12]:
        print("   %-42s %6d  %.2e / %.2e  %.2e / %.2e  %.2e / %.2e" % r)
    a = np.maximum(np.array([r[2:] for r in rows]), 1e-30)
    print("   all %d tensors, median rms: HIP %.2e -> %.2e with the fp64 decisions (fp32 reference %.2e);  largest max: HIP %.2e -> %.2e (fp32 reference %.2e)" % (
        len(rows), np.median(a[:, 0]), np.median(a[:, 2]), np.median(a[:, 4]), a[:, 1].max(), a[:, 3].max(), a[:, 5].max()))
    print("   ratio max / rms of the per-tensor error (a Gaussian over 4096 samples gives ~4): median HIP %.1f -> %.1f, fp32 reference %.1f;  largest HIP %.1f -> %.1f, fp32 reference %.1f" % (
        np.median(a[:, 1] / a[:, 0]), np.median(a[:, 3] / a[:, 2]), np.median(a[:, 5] / a[:, 4]),
        (a[:, 1] / a[:, 0]).max(), (a[:, 3] / a[:, 2]).max(), (a[:, 5] / a[:, 4]).max()))


if __name__ == "__main__":
    args = sys.argv[1:]
    golden = "--golden" in args
    if golden:
        args.remove("--golden")
    mode, cases = args[0], args[1:]
    for c in cases:
        if mode == "masks":
            make_masks(c, os.path.join(GOLDEN_DIR, "decisions_%s.npz" % c) if golden else None)
        else:
            run(c)
