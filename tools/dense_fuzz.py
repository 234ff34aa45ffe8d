"""Randomised check of the small kernels around the dense layers (profiles/DESIGN_r01_r03.md 4.10) against their torch compositions in fp64:
res_ln, qkv_split, softmax_last, linear (bias gradient), add_last_broadcast, thin_mm, bn_relu_max, bn_act(pre_bias) and the
two Poly-1 focal losses -- random shapes incl. row counts that do not fill a workgroup, all optional inputs on / off.

    python tools/dense_fuzz.py [--cases 60] [--seed 0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd import fused_norm as fn  # noqa: E402
from geot_amd.openpoints.loss import Poly1FocalLoss, Poly1FocalLoss_U_corr  # noqa: E402

DEV = "cuda:0"
worst = {}


def close(name, a, b, tol=3e-5):
    err = float((a.double() - b.double()).abs().max()) / (float(b.double().abs().max()) + 1e-9)
    worst[name] = max(worst.get(name, 0.0), err)
    assert err <= tol, (name, err)


def grads(out, inputs, ups):
    loss = sum((o * u.to(o.dtype)).sum() for o, u in zip(out, ups))
    return torch.autograd.grad(loss, inputs, allow_unused=True)


def case(rng):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)                      # noqa: E731
    b, n = int(rng.integers(1, 5)), int(rng.integers(1, 300))
    c = int(rng.choice([128, 256, 384, 512]))
    # ---- res_ln
    use_y, use_e = bool(rng.integers(2)), bool(rng.integers(2))
    use_s = use_y and bool(rng.integers(2))
    x0, y0, e0 = r(b, n, c), r(b, n, c), r(b, n, c)
    s0 = (torch.rand(b, 1, 1, generator=g) > 0.3).float().to(DEV) / 0.7
    ups = [r(b, n, c), r(b, n, c)]
    res = []
    for fused in (False, True):
        dt = torch.float32 if fused else torch.float64
        ln = torch.nn.LayerNorm(c).to(DEV).to(dt)
        with torch.no_grad():
            ln.weight.copy_(torch.linspace(0.5, 1.5, c)); ln.bias.copy_(torch.linspace(-1, 1, c))
        x = x0.detach().clone().to(dt).requires_grad_(True)
        y = y0.detach().clone().to(dt).requires_grad_(True) if use_y else None
        e = e0.detach().clone().to(dt).requires_grad_(True) if use_e else None
        s = s0.to(dt) if use_s else None
        if fused:
            t, z = fn.res_ln(x, y, s, e, ln)
        else:
            t = x
            if y is not None:
                t = t + (y if s is None else y * s)
            if e is not None:
                t = t + e
            z = ln(t)
        gs = grads([t, z], [v for v in (x, y, e, ln.weight, ln.bias) if v is not None], ups)
        res.append([t.detach(), z.detach()] + list(gs))
    for i, (a, f) in enumerate(zip(*res)):
        close("res_ln[%d]" % i, f, a)
    # ---- qkv_split + softmax_last
    h, d = int(rng.choice([1, 2, 4])), int(rng.choice([4, 24, 96]))
    nn_ = int(rng.choice([64, 128, 256, 512]))
    q0 = r(b, nn_, 3 * h * d)
    upa = r(b * h, nn_, nn_)
    res = []
    for fused in (False, True):
        dt = torch.float32 if fused else torch.float64
        x = q0.detach().clone().to(dt).requires_grad_(True)
        if fused:
            q, k, v = fn.qkv_split(x, h, 0.31)
            a = fn.softmax_last(torch.bmm(q, k.transpose(1, 2)))
        else:
            q, k, v = x.view(b, nn_, 3, h, d).permute(2, 0, 3, 1, 4).reshape(3, b * h, nn_, d).unbind(0)
            a = torch.bmm(q * 0.31, k.transpose(1, 2)).softmax(-1)
        o = torch.bmm(a, v)
        gs = grads([a, o], [x], [upa, r(b * h, nn_, d)] if False else [upa, torch.ones_like(o)])
        res.append([a.detach(), o.detach(), gs[0]])
    for i, (a, f) in enumerate(zip(*res)):
        close("attention[%d]" % i, f, a, 1e-4)
    # ---- linear
    cout = int(rng.integers(1, 300))
    lin64 = torch.nn.Linear(c, cout).to(DEV).double()
    lin32 = torch.nn.Linear(c, cout).to(DEV)
    lin32.load_state_dict(lin64.state_dict())
    up = r(b, n, cout)
    x64 = x0.double().requires_grad_(True)
    x32 = x0.clone().requires_grad_(True)
    g64 = grads([lin64(x64)], [x64, lin64.weight, lin64.bias], [up])
    g32 = grads([fn.linear(lin32, x32)], [x32, lin32.weight, lin32.bias], [up])
    for i, (a, f) in enumerate(zip(g64, g32)):
        close("linear[%d]" % i, f, a, 1e-4)
    # ---- add_last_broadcast, thin_mm
    rows, ns = int(rng.integers(1, 400)), int(rng.choice([4, 8, 16, 32, 64]))
    a0, p0, up = r(rows, ns), r(rows), r(rows, ns)
    a = a0.clone().requires_grad_(True); p = p0.clone().requires_grad_(True)
    ga, gp = grads([fn.add_last_broadcast(a, p)], [a, p], [up])
    close("add_last_broadcast", gp, up.double().sum(-1))
    j, l = int(rng.integers(1, 9)), int(rng.integers(1, 9000))
    w0, xx, up = r(rows, j), r(j, l), r(rows, l)
    w = w0.clone().requires_grad_(True)
    gw, = grads([fn.thin_mm(w, xx)], [w], [up])
    close("thin_mm", gw, up.double() @ xx.double().t())
    # ---- bn_relu_max, bn_act(pre_bias)
    cb, grp = int(rng.integers(1, 40)), int(rng.integers(1, 60))
    y0 = r(b, cb, grp * ns) * 2 + 0.3
    bias0 = r(cb)
    upm, upb = r(b, cb, grp), r(b, cb, grp * ns)
    res = []
    for fused in (False, True):
        dt = torch.float32 if fused else torch.float64
        bn = torch.nn.BatchNorm1d(cb).to(DEV).to(dt)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(-1, 1, cb) if cb > 1 else torch.ones(1)); bn.bias.copy_(torch.linspace(-0.5, 0.7, cb) if cb > 1 else torch.zeros(1))
        yv = y0.detach().clone().to(dt).requires_grad_(True)
        out = fn.bn_relu_max(bn, yv, ns) if fused else torch.relu(bn(yv)).view(b, cb, grp, ns).max(-1)[0]
        gs = grads([out], [yv, bn.weight, bn.bias], [upm])
        bn2 = torch.nn.BatchNorm1d(cb).to(DEV).to(dt)
        bv = bias0.detach().clone().to(dt).requires_grad_(True)
        yv2 = y0.detach().clone().to(dt).requires_grad_(True)
        out2 = fn.bn_act(bn2, yv2, relu=True, pre_bias=bv) if fused else torch.relu(bn2(yv2 + bv.view(1, -1, 1)))
        gs2 = grads([out2], [yv2, bn2.weight, bn2.bias], [upb])
        res.append([out.detach()] + list(gs) + [out2.detach()] + list(gs2) + [bn2.running_mean, bn2.running_var])
    if b * grp * ns > 1:
        for i, (a, f) in enumerate(zip(*res)):
            close("bn[%d]" % i, f, a, 2e-4)
    # ---- losses
    cl, npts = int(rng.integers(2, 20)), int(rng.integers(1, 3000))
    lg = r(b, cl, npts) * 4
    lab = torch.randint(0, cl, (b, npts), generator=g).to(DEV)
    conf = torch.rand(b, npts, generator=g).to(DEV)
    for cls, extra in ((Poly1FocalLoss, ()), (Poly1FocalLoss_U_corr, (conf, 0.5))):
        crit = cls(alpha=float(rng.choice([0.25, -1.0])), gamma=float(rng.choice([2.0, 1.5])))
        res = []
        for dev, dt in (("cpu", torch.float64), (DEV, torch.float32)):
            x = lg.detach().to(dev).to(dt).requires_grad_(True)
            loss = crit(x, lab.to(dev), *[v.to(dev) if torch.is_tensor(v) else v for v in extra])
            gx, = torch.autograd.grad(loss, [x])
            res.append((loss.detach().cpu(), gx.cpu()))
        close(cls.__name__, res[1][0], res[0][0], 1e-5)
        close(cls.__name__ + ".grad", res[1][1], res[0][1], 1e-4)


def main():
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    for i in range(a.cases):
        case(rng)
        if (i + 1) % 10 == 0:
            print("case %d ok" % (i + 1), flush=True)
    torch.cuda.synchronize()
    print("dense_fuzz: %d cases; worst relative errors: " % a.cases + ", ".join("%s %.1e" % kv for kv in sorted(worst.items())))


if __name__ == "__main__":
    main()
