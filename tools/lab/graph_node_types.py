"""Node types of the captured P / M graphs, through hipGraphGetNodes on torch's raw graph (keep_graph=True).
usage: graph_node_types.py model|fixmatch"""
import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd
import torch
from test_graph_step_gpu import _sup_batches, _fix_batch, SMALL, DEV
from geot_amd import train_step as ts, graph_step as gs
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
NAMES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord",
         8: "extSemSignal", 9: "extSemWait", 10: "memAlloc", 11: "memFree", 12: "memcpyFromSymbol", 13: "memcpyToSymbol"}
made = []
real = torch.cuda.CUDAGraph
class Keep(real):
    def __new__(cls, keep_graph=False):
        g = super().__new__(cls, keep_graph=True)
        made.append(g)
        return g
    def __init__(self, keep_graph=False):
        super().__init__(keep_graph=True)
torch.cuda.CUDAGraph = Keep
if sys.argv[1] == "model":
    torch.manual_seed(0)
    m = PointTransformer_seg_T(**SMALL).to(DEV)
    call = gs.GraphedSupervisedStep(ts.SupervisedStep(m))
    b = _sup_batches(2, 6000)
    for i in range(4):
        call(b[i % 2][0], b[i % 2][1], b[i % 2][2], next_pos=b[(i + 1) % 2][0])
else:
    torch.manual_seed(5)
    step = ts.build_fixmatch(DEV, seg_cfg=SMALL, cfg=dict(ts.NTM_CFG, threed_k=8), use_ddp=False)
    call = gs.GraphedFixMatchStep(step)
    b = [_fix_batch(3), _fix_batch(400)]
    for i in range(4):
        call(b[i % 2][0], b[i % 2][1], next_batches=b[(i + 1) % 2])
torch.cuda.synchronize()
for name, g in zip(("P", "M"), made):
    raw = ctypes.c_void_p(g.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
    kinds = collections.Counter()
    for nd in nodes:
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t)) == 0
        kinds[NAMES.get(t.value, t.value)] += 1
    print(sys.argv[1], name, "nodes:", n.value, dict(kinds))
