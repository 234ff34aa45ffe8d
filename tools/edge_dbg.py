import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geot_amd.openpoints.models.backbone.transformer_ops import edgeconv_tail
dev = torch.device("cuda:0")
b, c, nq, nk, k, groups = 2, 64, 1000, 700, 4, 4
g = torch.Generator(device="cpu").manual_seed(c + nq)
p0 = torch.randn(b, c, nk, generator=g).to(dev)
q0 = torch.randn(b, c, nq, generator=g).to(dev)
idx = torch.randint(0, nk, (b, nq, k), generator=g).to(torch.int32).to(dev)
norm = torch.nn.GroupNorm(groups, c).to(dev)
with torch.no_grad():
    norm.weight.copy_(torch.randn(c, generator=g)); norm.bias.copy_(torch.randn(c, generator=g))
up = torch.randn(b, c, nq, generator=g).to(dev)
res = []
for mode in ("f64", "torch", "fused"):
    dt = torch.float64 if mode == "f64" else torch.float32
    p, q = p0.clone().to(dt).requires_grad_(True), q0.clone().to(dt).requires_grad_(True)
    n2 = torch.nn.GroupNorm(groups, c).to(dev).to(dt); n2.load_state_dict(norm.state_dict())
    if mode == "fused":
        out = edgeconv_tail(p, q, idx, n2, 0.2)
    else:
        y = torch.gather(p, 2, idx.long().reshape(b, 1, nq * k).expand(-1, c, -1)).view(b, c, nq, k) + q.unsqueeze(-1)
        out = torch.nn.functional.leaky_relu(n2(y), 0.2).max(dim=-1)[0]
    (out * up.to(dt)).sum().backward()
    res.append((out.detach().double(), p.grad.double(), q.grad.double()))
for name, i in (("out", 0), ("dP", 1), ("dQ", 2)):
    ref = res[0][i]
    for m, r in (("torch32", res[1][i]), ("fused", res[2][i])):
        d = (r - ref).abs()
        print(name, m, "max abs err", float(d.max()), "scale", float(ref.abs().max()), "n>1e-4*scale", int((d > 1e-4 * ref.abs().max()).sum()))
d = (res[2][1] - res[0][1]).abs()
flat = d.flatten().topk(5)
for v, pos in zip(flat.values.tolist(), flat.indices.tolist()):
    bi, cc, n = pos // (c * nk), (pos // nk) % c, pos % nk
    deg = int((idx[bi] == n).sum())
    print("err", v, "at", (bi, cc, n), "ref", float(res[0][1][bi, cc, n]), "fused", float(res[2][1][bi, cc, n]), "torch", float(res[1][1][bi, cc, n]), "indeg", deg)
