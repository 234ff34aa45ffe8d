"""A long run of FixMatch+NTM iterations replayed from hipGraphs vs the same run eagerly at the bench's sizes: bit-identical?
usage: graph_long_fixmatch.py <points> <iterations>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd
import torch
from test_graph_step_gpu import _fix_batch, DEV, _state
from geot_amd import train_step as ts, graph_step as gs
n, iters = int(sys.argv[1]), int(sys.argv[2])
FLOOD = int(os.environ.get("FLOOD", "0"))      # eager launches between replays every 5th iteration (the packet-capture hazard)
flood_buf = torch.randn(1 << 16, device=DEV)
print("launch mode:", geot_amd.GRAPH_LAUNCH, "| packet capture off:", geot_amd.graph_replay_is_safe(), "| flood", FLOOD, flush=True)
batches = [_fix_batch(3, n), _fix_batch(400, n), _fix_batch(900, n)]
res = {}
for mode in ("eager", "graph"):
    torch.manual_seed(5)
    step = ts.build_fixmatch(DEV, use_ddp=False)
    call = gs.GraphedFixMatchStep(step) if mode == "graph" else step
    torch.manual_seed(11)
    losses = []
    for i in range(iters):
        k = i % 3 if i % 7 else 2          # mostly alternating, now and then an out-of-turn batch
        cur = batches[i % 2] if i % 7 else batches[2]
        nxt = batches[(i + 1) % 2]
        out = call(cur[0], cur[1], next_batches=nxt)
        if i % 5 == 0 or i == iters - 1:
            losses.append((i, float(out["loss"]), float(out["threed"])))
            if mode == "graph":
                for _ in range(FLOOD):
                    flood_buf.mul_(1.0)
    torch.cuda.synchronize()
    res[mode] = (losses, _state(step))
bad = [(a, b) for a, b in zip(res["eager"][0], res["graph"][0]) if a != b]
sa, sb = res["eager"][1], res["graph"][1]
diff = [k for k in sa if not torch.equal(sa[k], sb[k])]
print("fixmatch long run n=%d iters=%d: loss mismatches %s | differing state entries %d of %d | last %s" % (n, iters, bad[:2], len(diff), len(sa), res["graph"][0][-1]), flush=True)
