"""Run the composite workloads a few times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch, make_logits  # noqa: E402
from geot_amd import workloads as wl  # noqa: E402

B = int(os.environ.get("B", "8"))
which = os.environ.get("WHICH", "backbone")
xyz_np = make_batch(B, 24000)[0]
xyz = torch.from_numpy(xyz_np).cuda()
if which == "backbone":
    hot = wl.BackboneHotPath().cuda()
    tokens = torch.randn(B, 384, 512, device="cuda")
    for _ in range(4):
        wl.backbone_hotpath_step(hot, xyz, tokens)
else:
    nt = wl.NtmHotPath().cuda()
    pw = torch.from_numpy(make_logits(xyz_np, 0)).cuda()
    ps = torch.from_numpy(make_logits(xyz_np, 1, sharp=3.0)).cuda()
    for _ in range(4):
        wl.ntm_step(nt, xyz, pw, ps)
torch.cuda.synchronize()
print("done")
