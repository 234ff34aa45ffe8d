"""Randomised check of the remaining op families against their oracles: the channels-last pointops
(grouping / interpolation / subtraction / aggregation, forward and backward), the DGCNN graph feature, and the
dataloader's grid_subsampling (bit-exact against the numpy restatement that is pinned to the reference's C++).

    python tools/misc_fuzz.py [--cases 40] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.ext import pointops_cuda as pops  # noqa: E402
from geot_amd.openpoints.dataset import grid_subsampling  # noqa: E402
from oracle import capi, np_data  # noqa: E402  (checkers)

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def zeros(shape):
    return torch.zeros(shape, dtype=torch.float32, device=DEV)


def close(name, got, want, info, tol=1e-4):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    if not err <= tol:
        print("MISMATCH %s rel err %.3g %s" % (name, err, info))
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    for case in range(a.cases):
        rng = np.random.default_rng(15485863 * a.seed + case)
        n = int(rng.choice([1, 7, 100, 1000, 5000]))
        m = int(rng.choice([1, 33, 500, 3000]))
        ns = int(rng.choice([1, 3, 8, 16, 31]))
        c = int(rng.choice([1, 3, 16, 32, 48, 64]))
        wc = int(rng.choice([d for d in (1, 2, 4, 8, 16) if c % d == 0]))
        info = "case %d seed %d: n %d m %d ns %d c %d w_c %d" % (case, a.seed, n, m, ns, c, wc)
        f = lambda *s: rng.standard_normal(s).astype(np.float32)
        # grouping: input (n,c), idx (m,ns)
        inp, idx = f(n, c), rng.integers(0, n, (m, ns)).astype(np.int32)
        out = zeros((m, ns, c))
        pops.grouping_forward_cuda(m, ns, c, dev(inp), dev(idx), out)
        if not np.array_equal(out.cpu().numpy(), capi.grouping_cl(inp, idx)):
            print("MISMATCH grouping_cl", info); sys.exit(1)
        go, gin = f(m, ns, c), zeros((n, c))
        pops.grouping_backward_cuda(m, ns, c, dev(go), dev(idx), gin)
        close("grouping_cl grad", gin.cpu().numpy(), capi.grouping_cl_grad(go, idx, n), info)
        # interpolation: input (m,c) -> (n,c) with k = 3 weights
        src, ii, ww = f(m, c), rng.integers(0, m, (n, 3)).astype(np.int32), rng.random((n, 3)).astype(np.float32)
        out = zeros((n, c))
        pops.interpolation_forward_cuda(n, c, 3, dev(src), dev(ii), dev(ww), out)
        close("interpolation_cl", out.cpu().numpy(), capi.interpolation_cl(src, ii, ww), info, 1e-6)
        go, gin = f(n, c), zeros((m, c))
        pops.interpolation_backward_cuda(n, c, 3, dev(go), dev(ii), dev(ww), gin)
        close("interpolation_cl grad", gin.cpu().numpy(), capi.interpolation_cl_grad(go, ii, ww, m), info)
        # subtraction: in1 (n,c), in2 (n,c), idx (n,ns)
        a1, a2, si = f(n, c), f(n, c), rng.integers(0, n, (n, ns)).astype(np.int32)
        out = zeros((n, ns, c))
        pops.subtraction_forward_cuda(n, ns, c, dev(a1), dev(a2), dev(si), out)
        if not np.array_equal(out.cpu().numpy(), capi.subtraction_cl(a1, a2, si)):
            print("MISMATCH subtraction_cl", info); sys.exit(1)
        go, g1, g2 = f(n, ns, c), zeros((n, c)), zeros((n, c))
        pops.subtraction_backward_cuda(n, ns, c, dev(si), dev(go), g1, g2)
        w1, w2 = capi.subtraction_cl_grad(si, go)
        close("subtraction_cl grad1", g1.cpu().numpy(), w1, info)
        close("subtraction_cl grad2", g2.cpu().numpy(), w2, info)
        # aggregation: input (n,c), position (n,ns,c), weight (n,ns,w_c), idx (n,ns)
        pos, wt = f(n, ns, c), f(n, ns, wc)
        out = zeros((n, c))
        pops.aggregation_forward_cuda(n, ns, c, wc, dev(a1), dev(pos), dev(wt), dev(si), out)
        close("aggregation_cl", out.cpu().numpy(), capi.aggregation_cl(a1, pos, wt, si), info, 1e-5)
        go = f(n, c)
        gi, gp, gw = zeros((n, c)), zeros((n, ns, c)), zeros((n, ns, wc))
        pops.aggregation_backward_cuda(n, ns, c, wc, dev(a1), dev(pos), dev(wt), dev(si), dev(go), gi, gp, gw)
        wi_, wp_, ww_ = capi.aggregation_cl_grad(a1, pos, wt, si, go)
        close("aggregation_cl grad_in", gi.cpu().numpy(), wi_, info)
        close("aggregation_cl grad_pos", gp.cpu().numpy(), wp_, info)
        close("aggregation_cl grad_w", gw.cpu().numpy(), ww_, info)
        # grid_subsampling: bit-exact
        npts = int(rng.choice([1, 2, 50, 3000, 40000]))
        fd, ld = int(rng.choice([0, 1, 5])), int(rng.choice([0, 1, 2]))
        dl = float(rng.choice([0.01, 0.07, 0.3, 2.0, 50.0]))
        p = (rng.standard_normal((npts, 3)) * rng.choice([0.1, 1.0, 30.0]) + rng.choice([0.0, 100.0])).astype(np.float32)
        if npts > 10:
            p[rng.integers(0, npts, npts // 10)] = p[rng.integers(0, npts, npts // 10)]
        ft = f(npts, fd) if fd else None
        lb = rng.integers(-3, 4, (npts, ld)).astype(np.int32) if ld else None
        res = grid_subsampling(p, features=ft, labels=lb, sampleDl=dl)
        res = list(res) if isinstance(res, tuple) else [res]
        want = np_data.grid_subsampling(p, ft, lb, dl)
        ginfo = info + " | grid n %d fdim %d ldim %d dl %g" % (npts, fd, ld, dl)
        if not np.array_equal(res.pop(0), want["points"]):
            print("MISMATCH grid_subsampling points", ginfo); sys.exit(1)
        if fd and not np.array_equal(res.pop(0), want["features"]):
            print("MISMATCH grid_subsampling features", ginfo); sys.exit(1)
        if ld and not np.array_equal(res.pop(0), want["labels"]):
            print("MISMATCH grid_subsampling labels", ginfo); sys.exit(1)
        if case % 10 == 9:
            print("case %d ok" % case, flush=True)
    print("misc_fuzz: %d cases (channels-last pointops fwd/bwd, grid_subsampling) match their oracles" % a.cases)


if __name__ == "__main__":
    main()
