"""Randomised check of the gather-type gradients (gather / group / three_interpolate, plain and through the
reverse-index path) against the oracle's sequential sums: random channel counts, source / target sizes around the
planning thresholds (LDS rows, part lengths, 16-channel switch), duplicate-heavy index lists (hubs).

    python tools/grad_fuzz.py [--cases 60] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402
from oracle import capi  # noqa: E402  (checker)

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(name, got, want, info):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    if not err <= 2e-4:
        print("MISMATCH %s rel err %.3g %s" % (name, err, info))
        sys.exit(1)
    return err


def idx_list(rng, shape, upper):
    kind = rng.integers(0, 3)
    if kind == 0:
        return rng.integers(0, upper, shape).astype(np.int32)
    if kind == 1:                                   # hubs: most entries point at a few targets
        hubs = rng.integers(0, upper, 4)
        a = hubs[rng.integers(0, 4, shape)]
        mask = rng.random(shape) < 0.2
        a[mask] = rng.integers(0, upper, int(mask.sum()))
        return a.astype(np.int32)
    return (np.arange(int(np.prod(shape))).reshape(shape) % upper).astype(np.int32)   # every target equally often


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    worst = 0.0
    for case in range(a.cases):
        rng = np.random.default_rng(7919 * a.seed + case)
        b = int(rng.integers(1, 4))
        c = int(rng.choice([1, 3, 15, 16, 17, 33, 64, 100, 384]))
        n = int(rng.choice([5, 100, 1000, 4096, 8192, 12000, 24000, 36864, 40000]))
        m = int(rng.choice([1, 50, 512, 3000, 8192]))
        ns = int(rng.choice([1, 3, 16, 32]))
        if b * c * max(n, m * ns) > 60_000_000:      # keep the oracle's part quick
            c = min(c, 16)
        info = "case %d seed %d: b %d c %d n %d m %d ns %d" % (case, a.seed, b, c, n, m, ns)
        for impl in ("auto", "plain"):
            os.environ.pop("GEOT_GATHER_IMPL", None)
            if impl == "plain":
                os.environ["GEOT_GATHER_IMPL"] = "plain"
            # gather_points_grad: grad_out (b,c,m), idx (b,m) -> (b,c,n)
            gi = idx_list(rng, (b, m), n)
            go = rng.standard_normal((b, c, m)).astype(np.float32)
            worst = max(worst, close("gather_grad/" + impl, p2.gather_points_grad(dev(go), dev(gi), n).cpu().numpy(),
                                     capi.gather_points_grad(go, gi, n), info))
            # group_points_grad: grad_out (b,c,m,ns), idx (b,m,ns) -> (b,c,n)
            gg = idx_list(rng, (b, m, ns), n)
            go4 = rng.standard_normal((b, c, m, ns)).astype(np.float32)
            worst = max(worst, close("group_grad/" + impl, p2.group_points_grad(dev(go4), dev(gg), n).cpu().numpy(),
                                     capi.group_points_grad(go4, gg, n), info))
            # three_interpolate_grad: grad_out (b,c,n), idx/weight (b,n,3) over m sources -> (b,c,m)
            ti = idx_list(rng, (b, n, 3), m)
            tw = rng.random((b, n, 3)).astype(np.float32)
            go3 = rng.standard_normal((b, c, n)).astype(np.float32)
            worst = max(worst, close("interp_grad/" + impl,
                                     p2.three_interpolate_grad(dev(go3), dev(ti), dev(tw), m).cpu().numpy(),
                                     capi.three_interpolate_grad(go3, ti, tw, m), info))
        os.environ.pop("GEOT_GATHER_IMPL", None)
        if case % 10 == 9:
            print("case %d ok (worst rel err %.2g)" % (case, worst), flush=True)
    print("grad_fuzz: %d cases, worst relative error %.2g (limit 2e-4)" % (a.cases, worst))


if __name__ == "__main__":
    main()
