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


def main():
    cases = sys.argv[1:] or (H.REAL_CASES + H.CASES)
    for case in cases:
        for wino in ([-1, 0] if case in H.REAL_CASES else T.WINO_VARIANTS):
            H.SOFT_REPORT = []
            mp = pytest.MonkeyPatch()
            t0 = time.time()
            err = ""
            try:
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
            print("PARITY %-18s wino=%2d  %5.1fs  %s%s" % (case, wino, time.time() - t0, line or "ok (all ratios <= 1)", err), flush=True)
    H.SOFT_REPORT = None


if __name__ == "__main__":
    main()
