#!/usr/bin/env python
"""Per-section GPU time of one configs[2] step (HIP events from module hooks; forward sections, backward as a
whole, optimiser): where the step's time goes, to decide what to fuse next.  Lab tool, not product."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd.train_step import SupervisedStep
from geot_amd.synth import make_batch, region_labels

B = int(os.environ.get("B", "8"))
dev = torch.device("cuda:0")
xyz_np, _ = make_batch(B, 24000)
pos = torch.from_numpy(xyz_np).to(dev)
target = torch.from_numpy(region_labels(xyz_np)).to(dev)
cls = torch.zeros(B, 1, dtype=torch.long, device=dev)
torch.manual_seed(0)
model = PointTransformer_seg_T(**TOOTH_SEG_CFG, dense=os.environ.get("GEOT_DENSE", "factored"),
                               overlap=os.environ.get("OVERLAP", "1") == "1").to(dev)
step = SupervisedStep(model)
for _ in range(3):
    step(pos, cls, target)
rec = {}


def hook(name, mod):
    def pre(m, i):
        e = torch.cuda.Event(enable_timing=True); e.record(); m._e0 = e

    def post(m, i, o):
        e = torch.cuda.Event(enable_timing=True); e.record(); rec.setdefault(name, []).append((m._e0, e))
    mod.register_forward_pre_hook(pre)
    mod.register_forward_hook(post)


for name in ("group_divider", "encoder", "reduce_dim", "pos_embed", "blocks", "propogation_2", "propogation_1",
             "dgcnn_pro_2", "dgcnn_pro_1", "propogation_0", "seg_head"):
    hook(name, getattr(model, name))
# SupervisedStep pieces
ev = lambda: torch.cuda.Event(enable_timing=True)
tot = {"fwd": [], "loss": [], "bwd": [], "opt": []}
STEPS = 5
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    e = [ev() for _ in range(5)]
    e[0].record()
    model.train()
    logits = model(pos, pos.transpose(1, 2).contiguous(), cls)[0]
    e[1].record()
    loss = step.criterion(logits, target)
    e[2].record()
    loss.backward()
    e[3].record()
    step.optimizer.step(); step.optimizer.zero_grad(set_to_none=True)
    e[4].record()
    for k, (a, b) in zip(tot, zip(e[:-1], e[1:])):
        tot[k].append((a, b))
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / STEPS * 1e3
print("wall %.2f ms/step (%.1f clouds/s)" % (wall, B / wall * 1e3))
for k, v in tot.items():
    print("%-16s %8.2f ms" % (k, np.mean([a.elapsed_time(b) for a, b in v])))
print("-- forward sections")
for k, v in rec.items():
    print("%-16s %8.2f ms" % (k, np.mean([a.elapsed_time(b) for a, b in v])))
