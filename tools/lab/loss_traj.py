"""Loss of the bench's supervised step over many iterations (does the synthetic run stay finite?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd import train_step as ts, tuning
from geot_amd.synth import make_batch, region_labels
dev = torch.device("cuda:0")
tuning.enable(tune=False, path=None)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(1609)
model = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(dev)
tr = ts.SupervisedStep(model)
bs = []
for start in (0, 100003):
    x = make_batch(B, 24000, start_index=start)[0]
    bs.append((torch.from_numpy(x).to(dev), torch.from_numpy(np.random.default_rng(1609).integers(0, 2, size=(B, 1))).to(dev), torch.from_numpy(region_labels(x)).to(dev)))
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    c, n = bs[i % 2], bs[(i + 1) % 2]
    loss = tr(c[0], c[1], c[2], next_pos=n[0])
    if i % 4 == 0 or not torch.isfinite(loss):
        mx = max(float(p.detach().abs().max()) for p in model.parameters())
        print(i, float(loss), "max|w| %.3g" % mx, flush=True)
    if not torch.isfinite(loss):
        break
