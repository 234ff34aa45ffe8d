"""graph_parts.py for the FixMatch+NTM iteration: M alone, P alone, P beside M, the full replayed call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GEOT_GRAPH_LAUNCH", "fast")
import geot_amd
import torch
from test_graph_step_gpu import _fix_batch, DEV
from geot_amd import train_step as ts, graph_step as gs, tuning
print("TunableOp file:", tuning.enable(path=os.environ.get("TUNE_FILE")))
torch.manual_seed(0)
step = ts.build_fixmatch(DEV, use_ddp=False)
call = gs.GraphedFixMatchStep(step)
b = [_fix_batch(3, 24000), _fix_batch(400, 24000)]


def timed(fn, k=20):
    fn(); fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / k


i = [0]
def full():
    cur, nxt = b[i[0] % 2], b[(i[0] + 1) % 2]; i[0] += 1
    call(cur[0], cur[1], next_batches=nxt)
def nolook():
    cur = b[i[0] % 2]; i[0] += 1
    call(cur[0], cur[1])
for _ in range(4):
    full()
print("nodes", call.node_types)
print("full call, look-ahead      %.3f ms" % timed(full))
print("full call, no look-ahead   %.3f ms" % timed(nolook))
for _ in range(3):
    full()
M, P = call.graphs["M"][0], call.graphs["P"][0]
print("M replay alone             %.3f ms" % timed(M.replay))
print("P replay alone             %.3f ms" % timed(P.replay))
side = torch.cuda.Stream()
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        P.replay()
    M.replay()
    torch.cuda.current_stream().wait_stream(side)
print("P beside M, bare replays   %.3f ms" % timed(both))
