"""graph(look-ahead) -> eager -> graph(plain) on one SupervisedStep vs the same calls all eager: where do they part?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_graph_step_gpu import _sup_batches, SMALL, DEV, _state
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd import train_step as ts, graph_step as gs
full = len(sys.argv) > 1 and sys.argv[1] == "full"
cfg = TOOTH_SEG_CFG if full else SMALL
batches = _sup_batches(8 if full else 2, 24000 if full else 6000)
torch.manual_seed(0)
init = PointTransformer_seg_T(**cfg).state_dict()
if "tune" in sys.argv:
    from geot_amd import tuning
    tuning.enable(tune=False, path=None)
plans = {"A": [("g", False)] * 8, "B": [("g", True)] * 6 + [("g", False)] * 6,
         "C": [("g", True)] * 6 + [("e", True)] * 3 + [("g", False)] * 6,
         "D": [("g", True)] * 6 + [("e", False)] * 3 + [("g", False)] * 6,
         "E": [("g", False)] * 4 + [("e", True)] * 3 + [("g", False)] * 6}
plans.update({"F": [("g", True)] * 16 + [("e", True)] * 3 + [("g", False)] * 6,
              "G": [("g", True)] * 6 + [("e", True)] * 13 + [("g", False)] * 6,
              "H": [("g", True)] * 16 + [("g", False)] * 6,
              "I": [("g", True)] * 6 + [("e", False)] * 13 + [("g", False)] * 6,
              "J": [("g", True)] * 6 + [("e", True)] * 13 + [("g", True)] * 6})
plan = plans[sys.argv[2]]
runs = {}
for mode in ("eager", "mixed"):
    m = PointTransformer_seg_T(**cfg).to(DEV); m.load_state_dict(init)
    step = ts.SupervisedStep(m)
    graphed = gs.GraphedSupervisedStep(step)
    torch.manual_seed(7)
    losses = []
    for i, (how, look) in enumerate(plan):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]
        call = graphed if (how == "g" and mode == "mixed") else step
        losses.append(float(call(cur[0], cur[1], cur[2], next_pos=nxt[0] if look else None)))
    runs[mode] = losses
for i, (a, b) in enumerate(zip(runs["eager"], runs["mixed"])):
    print(sys.argv[2], i, plan[i], a, b, "" if a == b else "<-- differs")
