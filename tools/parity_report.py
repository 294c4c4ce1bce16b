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


def main():
    args = sys.argv[1:]
    routes = None
    if "--routes" in args:
        args.remove("--routes")
        routes = list(ROUTES)
    cases = args or (H.REAL_CASES + H.CASES)
    for case in cases:
        for wino in (routes or ([-1, 0] if case in H.REAL_CASES else T.WINO_VARIANTS)):
            H.SOFT_REPORT = []
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
    H.SOFT_REPORT = None


if __name__ == "__main__":
    main()
