"""Run a sequence of FixMatch runs in ONE process -- e<look> = eager, g<look> = graphed -- to bisect a capture fault.
usage: graph_bisect.py g1,g0"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faulthandler; faulthandler.enable()
import gc
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_graph_step_gpu import _fix_batch, SMALL, DEV
from geot_amd import train_step as ts, graph_step as gs
cfg = dict(ts.NTM_CFG, threed_k=8)
batches = [_fix_batch(3), _fix_batch(400)]
for run in sys.argv[1].split(","):
    look = run[1] == "1"
    torch.manual_seed(5)
    step = ts.build_fixmatch(DEV, seg_cfg=SMALL, cfg=cfg, use_ddp=False)
    call = gs.GraphedFixMatchStep(step, warmup=2) if run[0] == "g" else step
    for i in range(5):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]
        res = call(cur[0], cur[1], next_batches=nxt if look else None)
        torch.cuda.synchronize()
    print(run, "done", float(res["loss"]), flush=True)
    if "keep" not in sys.argv:
        del step, call, res
        gc.collect()
print("OK", sys.argv[1:], flush=True)
