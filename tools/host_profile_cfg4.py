"""Host-side profile of the cfg4 (StyleGAN2, --fp16) iteration (tools only): cProfile over the bench's timed iterations, functions by OWN time
(the iteration is host-bound: ~1 200 launches in 35 ms wall against 29 ms of kernels, profiles/r06_cfg4_host.txt).
    python tools/host_profile_cfg4.py [iterations]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = sys.argv[1] if len(sys.argv) > 1 else "32"
sys.argv = ["bench.py", "--workload", "cfg4", "--fp16", "--steps", steps, "--warmup", "4", "--no-cpu-baseline"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print("\n".join(l[:190] for l in s.getvalue().splitlines()))
