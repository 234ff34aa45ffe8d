"""Lab: are three supervised training steps of the configured model the same bits in every process?  Prints the losses (hex) and a
checksum of all parameters; run it several times and compare the lines."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd.synth import make_batch, region_labels  # noqa: E402
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG  # noqa: E402
from geot_amd.train_step import SupervisedStep  # noqa: E402
from geot_amd import tuning  # noqa: E402

DEV = torch.device("cuda:0")
B = int(os.environ.get("B", "2"))
tuning.enable()
xyz, _ = make_batch(B, 24000, start_index=20)
xyz2, _ = make_batch(B, 24000, start_index=40)
pos, pos2 = torch.from_numpy(xyz).to(DEV), torch.from_numpy(xyz2).to(DEV)
t1, t2 = torch.from_numpy(region_labels(xyz)).to(DEV), torch.from_numpy(region_labels(xyz2)).to(DEV)
cls = torch.zeros(B, 1, dtype=torch.long, device=DEV)
torch.manual_seed(11)
model = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = SupervisedStep(model, lr=1e-3)
losses = []
for it in range(3):
    cur, nxt = ((pos, t1), (pos2, t2)) if it % 2 == 0 else ((pos2, t2), (pos, t1))
    losses.append(float(step(cur[0], cls, cur[1], next_pos=nxt[0])))
chk = sum(int(p.detach().view(torch.int32).long().sum()) for p in model.parameters())
print("losses", " ".join(np.float32(v).tobytes().hex() for v in losses), " parameter bits", chk)
