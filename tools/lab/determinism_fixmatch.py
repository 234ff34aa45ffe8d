"""Lab: two FixMatch+NTM iterations of a small configuration; prints the loss components (hex) and a checksum of the student's and
the predictor's parameters.  Run it in several processes and compare.  GEOT_NTM_GRAD / other switches apply."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import train_step as ts  # noqa: E402
from geot_amd.synth import make_batch, region_labels  # noqa: E402

DEV = torch.device("cuda:0")
SMALL = dict(trans_dim=384, depth=3, num_heads=4, group_size=32, num_group=128, encoder_dims=256, nclasses=17,
             drop_path_rate=0.0, downsample_targets=[2048, 1024, 512], extract_layers=[1, 2, 3])
N = int(os.environ.get("POINTS", "6000"))          # POINTS=24000 FULL=1: the configured model at configs[4]'s sizes
if os.environ.get("FULL"):
    from geot_amd.openpoints.models.backbone.transformer import TOOTH_SEG_CFG
    SMALL = dict(TOOTH_SEG_CFG)
torch.manual_seed(2)
trainer = ts.build_fixmatch(DEV, seg_cfg=SMALL, use_ddp=False)
xyz = make_batch(2, N, start_index=0)[0]
pos = torch.from_numpy(xyz).to(DEV)
target = torch.from_numpy(region_labels(xyz)).to(DEV)
xu = torch.from_numpy(make_batch(2, N, start_index=50)[0]).to(DEV)
xs = (xu * 1.1).contiguous()
z = torch.zeros(2, 1, dtype=torch.long, device=DEV)
data = {"pos": pos, "x": pos.transpose(1, 2).contiguous(), "cls": z, "y": target}
data_u = {"pos_w": xu, "x_w": xu.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": xs,
          "x_s": xs.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": xu}
out = [trainer(data, data_u) for _ in range(2)]
hexes = " | ".join(" ".join("%s=%s" % (k, np.float32(float(v)).tobytes().hex()) for k, v in o.items()) for o in out)
chk = lambda m: sum(int(p.detach().contiguous().view(torch.int32).long().sum()) for p in m.parameters())
print(hexes, " student", chk(trainer.model), " predictor", chk(trainer.T_predictor), " ema_t", int(trainer.ema_t.view(torch.int32).long().sum()))
