"""Lab: bit sums of every tensor of one fp_stage_cl forward + backward on a fixed input; run it in several processes and compare."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import fused_norm as fn  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.pointnet2 import pointnet2_utils as pu  # noqa: E402

DEV = torch.device("cuda:0")


def bits(t):
    return int(t.detach().contiguous().view(torch.int32).long().sum())


b, n, m, c = 8, 24000, 8192, 384
pos = torch.from_numpy(make_batch(b, n, start_index=7)[0]).to(DEV)
unknown, known = pos, pos[:, :m].contiguous()
d2, idx = pu._ext.three_nn(unknown, known)
w = pu._ext.fp_weights(d2)
torch.manual_seed(3)
a = torch.randn(b, m, c, device=DEV, requires_grad=True)
skip = torch.randn(b, 3, n, device=DEV)
wb = torch.randn(c, 3, device=DEV, requires_grad=True)
up = torch.randn(b, n, c, device=DEV)
bn = torch.nn.BatchNorm1d(c).to(DEV)
ou, ok = fn.local_spatial_order(unknown), fn.local_spatial_order(known)
rix = fn.ReverseIndex(idx, w, m, ok)
out = ["idx %d w %d order_u %d order_k %d rix %d" % (bits(idx), bits(w), bits(ou), bits(ok), bits(rix.ws[:int(rix.ws_ints) - 8])),
       "a %d skip %d wb %d up %d pos %d d2 %d" % (bits(a), bits(skip), bits(wb), bits(up), bits(pos), bits(d2))]
with torch.no_grad():
    y0, part0 = fn.fp_front_cl(a.detach(), idx, w, skip, wb.detach(), ou, rix)
    y1, part1 = fn.fp_front_cl(a.detach(), idx, w, skip, wb.detach(), None, rix)
    out.append("y(Morton) %d partial %d (%d floats)   y(memory order) %d partial %d" % (bits(y0), bits(part0), part0.numel(), bits(y1), bits(part1)))
for fused in (True, False):
    a.grad = wb.grad = None
    bn.zero_grad()
    bn.running_mean.zero_(); bn.running_var.fill_(1.0)
    if fused:
        z = fn.fp_stage_cl(bn, a, idx, w, skip, wb, True, ou, rix)
    else:
        y, part = fn.fp_front_cl(a, idx, w, skip, wb, ou, rix)
        z = fn.bn_act_cl(bn, y, relu=True, partial=part)
    (z * up).sum().backward()
    out.append("%s: z %d running_mean %d running_var %d dA %d dWb %d dgamma %d dbeta %d" %
               ("fused   " if fused else "two-node", bits(z), bits(bn.running_mean), bits(bn.running_var), bits(a.grad), bits(wb.grad),
                bits(bn.weight.grad), bits(bn.bias.grad)))
print("\n".join(out))
