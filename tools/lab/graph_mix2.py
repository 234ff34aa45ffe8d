"""After look x6 + eager x13 + ONE graph call: which parameters / buffers / AdamW moments differ from the all-eager run?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_graph_step_gpu import _sup_batches, DEV, _state
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd import train_step as ts, graph_step as gs
batches = _sup_batches(8, 24000)
torch.manual_seed(0)
init = PointTransformer_seg_T(**TOOTH_SEG_CFG).state_dict()
n_e = int(sys.argv[1]) if len(sys.argv) > 1 else 13
plan = [("g", True)] * 6 + [("e", True)] * n_e + [("g", True)] * 1
states = {}
for mode in ("eager", "mixed"):
    m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV); m.load_state_dict(init)
    step = ts.SupervisedStep(m)
    graphed = gs.GraphedSupervisedStep(step)
    torch.manual_seed(7)
    for i, (how, look) in enumerate(plan):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]
        call = graphed if (how == "g" and mode == "mixed") else step
        loss = float(call(cur[0], cur[1], cur[2], next_pos=nxt[0] if look else None))
    torch.cuda.synchronize()
    states[mode] = _state(step)
    names = [n for n, _ in m.named_parameters()]
a, b = states["eager"], states["mixed"]
bad = [k for k in a if not torch.equal(a[k], b[k])]
print("differing entries: %d of %d" % (len(bad), len(a)))
pn = [n for n, p in m.named_parameters() if p.requires_grad]
for k in bad[:60]:
    d = (a[k].double() - b[k].double()).abs().max().item()
    label = k
    if k.startswith("opt0."):
        idx = int(k.split(".")[1])
    print("  %-60s max|diff| %.3g  numel %d" % (label, d, a[k].numel()))
