#!/usr/bin/env python
"""run bench.py with Winograd thresholds overridden:  python tools/bench_knobs.py MIN MIN2 MIN4 WGRAD4 -- <bench args>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ic_gan_amd.ops as o
k = sys.argv.index("--")
o.WINOGRAD_MIN_CHANNELS, o.WINOGRAD2_MIN_CHANNELS, o.WINOGRAD4_MIN_CHANNELS, o.WINOGRAD4_WGRAD_MIN_CHANNELS = map(int, sys.argv[1:k])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[k + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
