"""Runs the HBM-bound kernels of the path at B clouds a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch, make_logits  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402
from geot_amd import ntm  # noqa: E402
from geot_amd.openpoints.models.backbone.transformer_ops import graph_feature  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402

B, N, DEV = int(os.environ.get("B", "8")), 24000, "cuda"
xyz_np = make_batch(B, N)[0]
xyz = torch.from_numpy(xyz_np).to(DEV)
known = xyz[:, :8192].contiguous()
_, i3 = p2.three_nn(xyz, known)
w = torch.rand(B, N, 3, device=DEV); w = w / w.sum(2, keepdim=True)
f384 = torch.randn(B, 384, 8192, device=DEV)
g384 = torch.randn(B, 384, N, device=DEV)
feats = torch.randn(B, 64, N, device=DEV)
c6000 = p2.furthest_point_sampling(xyz, 6000)
new_xyz = p2.gather_points(xyz.transpose(1, 2).contiguous(), c6000).transpose(1, 2).contiguous()
bq = p2.ball_query(new_xyz, xyz, 0.1, 32)
go = torch.randn(B, 64, 6000, 32, device=DEV)
xq = torch.randn(B, 384, 8192, device=DEV)
_, kidx = knn_sorted(known, known, 4)
C = 17
prob = torch.softmax(torch.from_numpy(make_logits(xyz_np, 1)).to(DEV), 1)
cm = torch.softmax(torch.randn(C, C, device=DEV), 1)
pred = ntm.Ins_T_mean(nclasses=C).to(DEV)
for _ in range(3):
    with torch.no_grad():
        p2.three_interpolate(f384, i3, w)
        p2.three_interpolate_grad(g384, i3, w, 8192)
        p2.group_points(feats, bq)
        p2.group_points_grad(go, bq, N)
        graph_feature(xq, xq, kidx)
        insT = pred(prob, cm)
        ntm.correct_logits(prob, insT, cm, 0.9)
torch.cuda.synchronize()
print("done")
