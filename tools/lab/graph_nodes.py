"""Node types of the captured P / M graphs (hipGraphDebugDotPrint): which are NOT kernel launches?  usage: graph_nodes.py model|fixmatch"""
import os, sys, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_graph_step_gpu import _sup_batches, _fix_batch, SMALL, DEV
from geot_amd import train_step as ts, graph_step as gs
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
orig = torch.cuda.CUDAGraph.__new__
made = []
class DbgGraph(torch.cuda.CUDAGraph):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k); return g
def patched_run(self, name, fn, _orig=gs._Graphed._run):
    had = name in self.graphs
    real = torch.cuda.CUDAGraph
    if not had and self._eager_runs[name] >= self.warmup:
        class G(real):
            pass
        def mk():
            g = real(); g.enable_debug_mode(); made.append((name, g)); return g
        torch.cuda.CUDAGraph = mk
    try:
        return _orig(self, name, fn)
    finally:
        torch.cuda.CUDAGraph = real
gs._Graphed._run = patched_run
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
for name, g in made:
    path = "/tmp/graph_%s.dot" % name
    g.debug_dump(path)
    text = open(path).read()
    labels = re.findall(r'label="([^"]*)"', text)
    kinds = collections.Counter()
    for l in labels:
        k = l.split("\\n")[0].split("(")[0].strip()
        kinds["KERNEL" if "kernel" in l.lower() or "Cijk" in l or "<" in l else k[:40]] += 1
    print(name, "nodes:", len(labels))
    for k, v in kinds.most_common(12):
        print("   %5d  %s" % (v, k))
    ms = [l for l in labels if "memset" in l.lower()]
    print("   memset nodes: %d" % len(ms), ms[:3])
    mc = [l for l in labels if "memcpy" in l.lower()]
    print("   memcpy nodes: %d" % len(mc), mc[:2])
