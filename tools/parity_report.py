#!/usr/bin/env python
"""Parity report (GPU): runs the golden train-step comparison of tests/test_parity_gpu.py for every (case, Winograd variant)
at the STRICT tolerances and prints the worst error/tolerance ratio per tensor group instead of stopping at the first
failure.  Usage: python tools/parity_report.py [case ...]      ->  one line per (case, variant), `ok` when every ratio <= 1."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pytest

import tests.helpers as H
from tests import test_parity_gpu as T


BIG = 10 ** 9
ROUTES = {   # --routes: which layers take a Winograd form (production thresholds otherwise), to attribute the gradient error
    "prod": {},
    "plain_only": {"RS_WINOGRAD_MIN_CHANNELS": {True: (BIG, BIG, BIG), False: (BIG, BIG, BIG)}},
    "rs_only": {"WINOGRAD_MIN_CHANNELS": BIG},
    "no_wgrad": {"WINOGRAD_WGRAD": False, "RS_WINOGRAD_MIN_CHANNELS": {True: (96, 192, BIG), False: (192, 192, BIG)}},
    "min192": {"WINOGRAD_MIN_CHANNELS": 192, "RS_WINOGRAD_MIN_CHANNELS": {True: (192, 192, 192), False: (192, 192, 192)}},
    "min384": {"WINOGRAD_MIN_CHANNELS": 384, "RS_WINOGRAD_MIN_CHANNELS": {True: (384, 384, 384), False: (384, 384, 384)}},
    "min768": {"WINOGRAD_MIN_CHANNELS": 768, "RS_WINOGRAD_MIN_CHANNELS": {True: (768, 768, 768), False: (768, 768, 768)}},
}


def error_stats(case):
    """--stats: per gradient tensor of a real-width case, the distance of the HIP result and of the fp32 reference from the
    fp64 reference over the stored samples (4096 per tensor): rms and max, in units of the tensor's rms.  Says whether the HIP
    path is as close to the exact gradient as the reference's own fp32 arithmetic is."""
    import json
    f64 = os.path.join(H.GOLDEN_DIR, "biggan_%s_f64.npz" % case)
    if not os.path.exists(f64):
        return
    g32, g64 = H.load_golden(case), np.load(f64, allow_pickle=False)
    rows = []
    for what, name, rms, tol, samp in H.STATS:
        prefix = {"G grad ": "step1/G_grad/", "D grad ": "step1/D_grad/"}.get(what)
        if prefix is None:
            continue
        names = json.loads(str(g32[prefix + "names"]))
        i = names.index(name)
        r32, r64 = g32[prefix + "samp"][i], g64[prefix + "samp"][i]
        k = int(np.count_nonzero(r64)) or 1
        s = max(rms, 1e-30)
        e_hip64, e_ref, e_hip32 = samp - r64, r32 - r64, samp - r32
        rows.append((what + name, k, rms, np.sqrt((e_hip64 ** 2).sum() / k) / s, np.abs(e_hip64).max() / s,
                     np.sqrt((e_ref ** 2).sum() / k) / s, np.abs(e_ref).max() / s, np.abs(e_hip32).max() / tol))
    rows.sort(key=lambda r: -r[4])
    print("  STATS %s: distance from the fp64 reference in units of the tensor rms -- HIP rms / max | fp32 reference rms / max | HIP vs fp32 golden max / tol" % case)
    for r in rows[:12]:
        print("    %-40s n=%5d rms %.2e   HIP %.2e / %.2e | ref32 %.2e / %.2e | x%.2f" % r)
    hip = np.array([r[3] for r in rows]); ref = np.array([r[5] for r in rows])
    print("    all %d tensors: median HIP rms error %.2e, median fp32-reference rms error %.2e, worst ratio HIP/ref of the rms errors %.1f (%s)" % (
        len(rows), np.median(hip), np.median(ref), float(np.max(hip / np.maximum(ref, 1e-12))), rows[int(np.argmax(hip / np.maximum(ref, 1e-12)))][0]))


def main():
    args = sys.argv[1:]
    routes = None
    stats = "--stats" in args
    if stats:
        args.remove("--stats")
    if "--routes" in args:
        args.remove("--routes")
        routes = list(ROUTES)
    cases = args or (H.REAL_CASES + H.BENCH_CASES + H.CASES)
    for case in cases:
        for wino in (routes or ([-1, 0] if case in H.REAL_CASES + H.BENCH_CASES else T.WINO_VARIANTS)):
            H.SOFT_REPORT = []
            H.STATS = [] if stats else None
            H.DENSE_REPORT = []
            mp = pytest.MonkeyPatch()
            t0 = time.time()
            err = ""
            try:
                if routes:
                    import ic_gan_amd.ops as _ops
                    for k, v in ROUTES[wino].items():
                        mp.setattr(_ops, k, v)
                    label, wino = wino, 0
                T._train_steps_case(case, wino, mp, strict=True)
            except AssertionError as exc:          # losses (np.testing) are hard asserts
                err = " HARD-FAIL " + str(exc).replace("\n", " ")[:200]
            finally:
                mp.undo()
            fails = sorted(H.SOFT_REPORT, key=lambda r: -r[1])
            worst = {}
            for msg, ratio in fails:
                grp = msg.split(" ")[0] + (" grad" if " grad " in msg else "")
                worst.setdefault(grp, (msg, ratio))
            line = "; ".join("%s x%.2f" % (m[:70], r) for m, r in worst.values())
            tag = ("route=%-10s" % label) if routes else ("wino=%2d" % wino)
            print("PARITY %-18s %s  %5.1fs  %s%s" % (case, tag, time.time() - t0, line or "ok (all ratios <= 1)", err), flush=True)
            if H.DENSE_REPORT:      # dense (4096-sample) gradient groups: the four criteria of helpers.check_group, worst tensor of each
                d = H.DENSE_REPORT
                a, b, c, e = (max(d, key=lambda r: r[j]) for j in (1, 2, 3, 4))
                print("  DENSE %d tensors: sub-grid max/tol x%.2f (%s); max/tol x%.2f of %.1f allowed (%s); rms/tol x%.3f of %.2f allowed (%s); "
                      "share above tol %.4f of %.4f allowed (%s)" % (len(d), a[1], a[0], b[2], H.DENSE_MAX_MULT, b[0], c[3], H.DENSE_RMS_FRAC,
                                                                      c[0], e[4], H.DENSE_EXCEED_SHARE, e[0]), flush=True)
            if stats:
                error_stats(case)
    H.SOFT_REPORT = None
    H.STATS = None
    H.DENSE_REPORT = None


if __name__ == "__main__":
    main()
