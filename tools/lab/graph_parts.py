"""What the replayed step's wall time is made of: M1+M2 alone, P alone, the full call with and without look-ahead, the eager
iteration over the same static geometry.  usage: graph_parts.py [clouds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GEOT_GRAPH_LAUNCH", "fast")
import geot_amd
import torch
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd import train_step as ts, graph_step as gs
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if os.environ.get("BLAS"):                      # BLAS=cublas (= rocBLAS) | cublaslt (= hipBLASLt), with TUNED=0
    print("preferred blas:", torch.backends.cuda.preferred_blas_library(os.environ["BLAS"]))
if os.environ.get("TUNED", "1") == "1":        # the GEMM selection bench.py runs with (geot_amd/tuning)
    from geot_amd import tuning
    print("TunableOp file:", tuning.enable(path=os.environ.get("TUNE_FILE")))
torch.manual_seed(0)
m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = ts.SupervisedStep(m)
call = gs.GraphedSupervisedStep(step)
b = _sup_batches(B, 24000)


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
    call(cur[0], cur[1], cur[2], next_pos=nxt[0])
def nolook():
    cur = b[i[0] % 2]; i[0] += 1
    call(cur[0], cur[1], cur[2])
for _ in range(4):
    full()
print("clouds %d  split %s  nodes %s" % (B, call.split, call.node_types))
print("full call, look-ahead      %.3f ms" % timed(full))
print("full call, no look-ahead   %.3f ms" % timed(nolook))
for _ in range(3):
    full()
names = [n for n in ("M", "M1", "M2") if n in call.graphs]
def m_only():
    for n in names:
        call.graphs[n][0].replay()
print("%s replays alone        %.3f ms" % ("+".join(names), timed(m_only)))
print("P replay alone             %.3f ms" % timed(lambda: call.graphs["P"][0].replay()))
side = torch.cuda.Stream()
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call.graphs["P"][0].replay()
    m_only()
    torch.cuda.current_stream().wait_stream(side)
print("P beside M, bare replays   %.3f ms" % timed(both))
if len(names) == 2:
    def between():
        call.graphs["M1"][0].replay()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call.graphs["P"][0].replay()
        call.graphs["M2"][0].replay()
        torch.cuda.current_stream().wait_stream(side)
    print("M1, P beside M2, bare      %.3f ms" % timed(between))
# eager over a static geometry
pre = step.lookahead_work(b[0][0])
def eager():
    step.iteration(b[0][0], b[0][1], b[0][2], pre, None)
print("eager iteration, static geometry (no P)   %.3f ms" % timed(eager))
# which part of P costs M its time: the long FPS alone on the side stream, beside the bare M replays
from geot_amd.pointops.functions import pointops as pops
xyz = b[0][0].reshape(-1, 3).contiguous()
gf = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream()
s2.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s2):
    pops.furthestsampling_uniform(xyz, B, 24000, 8192)
torch.cuda.current_stream().wait_stream(s2)
torch.cuda.synchronize()
with torch.cuda.graph(gf):
    idx = pops.furthestsampling_uniform(xyz, B, 24000, 8192)
print("FPS-8192 graph alone       %.3f ms" % timed(gf.replay))
def fps_beside():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gf.replay()
    m_only()
    torch.cuda.current_stream().wait_stream(side)
print("FPS-8192 beside M, bare    %.3f ms" % timed(fps_beside))
# (tried: M's replays on a priority -1 stream -- 63 ms per replay instead of 34.5, with or without the FPS beside it: not a path)
