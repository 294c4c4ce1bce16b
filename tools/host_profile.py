"""Host-side profile of the cfg3 step (tools only): cProfile over K bench steps -> cumulative time per function, to see what the host does
between the loss read-back of one step and the first kernel of the next (the device idles there: tools/gpu_step_trace.sh).
    python tools/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = sys.argv[1] if len(sys.argv) > 1 else "6"
sys.argv = ["bench.py", "--steps", steps, "--warmup", "2", "--no-cpu-baseline", "--no-uninstrumented-leg", "--no-kernel-timer", "--init", "N02"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
    st.print_stats(70)
    out = s.getvalue()
    print("\n".join(l[:200] for l in out.splitlines()))
