"""Lab: where the HOST time of a FixMatch+NTM iteration (or the supervised step) goes -- cProfile over 10 iterations with the GPU
queue kept short (a synchronize per iteration, so the profile shows issue cost, not back-pressure)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

WL = os.environ.get("WL", "fixmatch")
POINTS = int(os.environ.get("POINTS", "24000"))
sys.argv = ["bench.py", "--workload", WL, "--points", str(POINTS), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-dense-reference"]
# reuse bench.py's construction of the step by running it once with a hook that hands the step function back
captured = {}
orig = bench.timed_steps


def grab(step, steps, dev, rehearsal):
    captured.setdefault("step", step)          # the first timed loop is the workload itself
    return orig(step, steps, dev, rehearsal)


bench.timed_steps = grab
try:
    bench.main()
except SystemExit:
    pass
step = captured["step"]
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
