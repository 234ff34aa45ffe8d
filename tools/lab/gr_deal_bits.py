"""Lab: bit pattern of the fused / plain row gathers' outputs on a fixed input (compare between a -DGEOT_GR_LAB_SHARES build and the
default: the deal changes which workgroup walks a list, not the order inside it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import fused_norm as fn  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.pointnet2 import pointnet2_utils as pu  # noqa: E402

DEV = torch.device("cuda:0")
for b, n, m, c in ((8, 24000, 8192, 384), (3, 5000, 77, 260), (1, 1000, 5, 256)):
    pos = torch.from_numpy(make_batch(b, n, start_index=7)[0]).to(DEV)
    unknown, known = pos, pos[:, :m].contiguous()
    d2, idx = pu._ext.three_nn(unknown, known)
    w = pu._ext.fp_weights(d2)
    torch.manual_seed(3)
    a = torch.randn(b, m, c, device=DEV, requires_grad=True)
    skip = torch.randn(b, 3, n, device=DEV)
    wb = torch.randn(c, 3, device=DEV, requires_grad=True)
    bn = torch.nn.BatchNorm1d(c).to(DEV)
    rix = fn.ReverseIndex(idx, w, m, fn.local_spatial_order(known))
    z = fn.fp_stage_cl(bn, a, idx, w, skip, wb, True, fn.local_spatial_order(unknown), rix)
    (z * torch.randn_like(z)).sum().backward()
    g = torch.randn(b, n, c, device=DEV)
    plain = rix.gather(g)
    print("b=%d n=%5d m=%4d c=%4d  fused dA bits %d  plain gather bits %d" %
          (b, n, m, c, int(a.grad.view(torch.int32).long().sum()), int(plain.view(torch.int32).long().sum())))
