"""A long run of graph replays vs the same run eagerly: do they stay bit-identical?  usage: graph_long.py <clouds> <iterations>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd
import torch
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd import train_step as ts, graph_step as gs
B, N = int(sys.argv[1]), int(sys.argv[2])
FLOOD = int(os.environ.get("FLOOD", "0"))      # eager launches between replays every 10th iteration (the packet-capture hazard)
flood_buf = torch.randn(1 << 16, device=DEV)
print("launch mode:", geot_amd.GRAPH_LAUNCH, "| packet capture off:", geot_amd.graph_replay_is_safe(), "| flood", FLOOD, flush=True)
batches = _sup_batches(B, 24000)
torch.manual_seed(0)
init = PointTransformer_seg_T(**TOOTH_SEG_CFG).state_dict()
res = {}
for mode in ("eager", "graph"):
    m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV); m.load_state_dict(init)
    step = ts.SupervisedStep(m)
    graphed = gs.GraphedSupervisedStep(step)
    call = graphed if mode == "graph" else step
    torch.manual_seed(7)
    losses = []
    for i in range(N):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]
        out = call(cur[0], cur[1], cur[2], next_pos=nxt[0])
        if i % 10 == 0 or i == N - 1:
            losses.append((i, float(out)))
            if mode == "graph":
                for _ in range(FLOOD):
                    flood_buf.mul_(1.0)
    res[mode] = losses
bad = [(a[0], a[1], b[1]) for a, b in zip(res["eager"], res["graph"]) if a[1] != b[1]]
print("long run B=%d N=%d: first difference" % (B, N), bad[:2], "| last", res["eager"][-1], res["graph"][-1], flush=True)
