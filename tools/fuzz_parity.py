"""Randomised differential run of the index-producing ops against the CPU oracle.

    python tools/fuzz_parity.py [--cases 60] [--seed 0]

Random batch sizes, point counts (1 .. a few thousand, ragged segments for the offset-batched ops), sample
counts above and below n, radii / nsample / k over their whole range, point sets with duplicates, lattice
ties, collinear and coplanar clouds, clouds with fewer valid / distinct points than samples; every implementation switch (GEOT_FPS_IMPL multi|single|basic,
GEOT_NN_IMPL grid|wave|basic).  Every output must be bit-identical to the oracle's.  Prints one line per op
and exits non-zero on the first mismatch (the failing case is printed with its seed).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.ext import pointnet2_ext as p2, pointnet2_batch_cuda as p2b, pointops_cuda as pops  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402
from oracle import capi  # noqa: E402  (checker)

DEV = "cuda:0"


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dt) if dt is not None else t).to(DEV)


def cloud(rng, b, n):
    kind = rng.integers(0, 8)
    if kind == 0:
        p = rng.random((b, n, 3))
    elif kind == 1:                                   # lattice: exact distance ties
        p = rng.integers(0, 6, (b, n, 3)) * 0.125
    elif kind == 2:                                   # duplicates
        p = rng.random((b, n, 3))
        p[:, rng.integers(0, n, max(1, n // 4))] = p[:, rng.integers(0, n, max(1, n // 4))]
    elif kind == 3:                                   # a line
        p = rng.random((b, n, 1)) * np.array([1.0, 0.5, -0.25])
    elif kind == 4:                                   # a plane + an outlier
        p = rng.random((b, n, 3)) * np.array([1.0, 1.0, 0.0])
        p[:, 0] = 40.0
    elif kind == 5:                                   # tight clusters around the origin (K1 origin-skip rule)
        p = rng.standard_normal((b, n, 3)) * 0.01
    elif kind == 6:                                   # almost everything AT the origin, a handful of valid points:
        p = np.zeros((b, n, 3))                       # fewer valid points than samples (exhaustion, repeated picks)
        far = rng.choice(n, min(n, int(rng.integers(1, 60))), replace=False)
        p[:, far] = rng.random((b, len(far), 3)) + 0.5
    else:                                             # few distinct locations, each many times over
        d = int(rng.integers(2, 50))
        p = rng.random((b, d, 3))[:, rng.integers(0, d, n)]
    return p.astype(np.float32)


def ragged(rng, b, lo, hi):
    sizes = rng.integers(lo, hi + 1, b)
    return sizes, np.cumsum(sizes).astype(np.int32)


def check(name, got, want, info):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print("MISMATCH %s %s first at %s: got %s want %s" % (name, info, bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    for case in range(a.cases):
        rng = np.random.default_rng(a.seed * 100003 + case)
        b = int(rng.integers(1, 4))
        n = int(rng.choice([1, 2, 3, 7, 63, 64, 65, 200, 513, 1025, 2500, 5000]))
        xyz = cloud(rng, b, n)
        info = "case %d seed %d b %d n %d" % (case, a.seed, b, n)
        # ---- FPS, the three rule sets, the three kernels
        m = int(rng.integers(1, n + 6)) if (n < 600 or (n <= 2500 and rng.random() < 0.3)) else int(rng.integers(1, 400))
        for impl in ("multi", "single", "basic"):
            os.environ["GEOT_FPS_IMPL"] = impl
            check("fps_k1/" + impl, p2.furthest_point_sampling(dev(xyz), m).cpu().numpy(), capi.fps_dense(xyz, m, 512, True), info)
            out = torch.full((b, m), -7, dtype=torch.int32, device=DEV)
            tmp = torch.full((b, n), 1e10, dtype=torch.float32, device=DEV)
            p2b.furthest_point_sampling_wrapper(b, n, m, dev(xyz), tmp, out)
            check("fps_k1p/" + impl, out.cpu().numpy(), capi.fps_dense(xyz, m, 1024, False), info)
            sizes, off = ragged(rng, b, 1, n)
            flat = np.concatenate([xyz[i, :sizes[i]] for i in range(b)])
            msz = np.array([int(rng.integers(1, s + 3)) if (s < 600 or (s <= 2500 and rng.random() < 0.3)) else int(rng.integers(1, 300))
                            for s in sizes])
            noff = np.cumsum(msz).astype(np.int32)
            w = (0.5 + rng.random(len(flat))).astype(np.float32) if rng.random() < 0.5 else None
            idx = torch.full((int(noff[-1]),), -7, dtype=torch.int32, device=DEV)
            tmp = torch.full((len(flat),), 1e10, dtype=torch.float32, device=DEV)
            if w is None:
                pops.furthestsampling_cuda(b, int(sizes.max()), dev(flat), dev(off), dev(noff), tmp, idx)
            else:
                pops.furthestsampling_weights_cuda(b, int(sizes.max()), dev(flat), dev(off), dev(noff), dev(w), tmp, idx)
            check("fps_k2/" + impl, idx.cpu().numpy(), capi.fps_offset(flat, off, noff, w), info)
        os.environ.pop("GEOT_FPS_IMPL")
        # ---- neighbour queries
        nq = int(rng.choice([1, 5, 64, 300, 1500]))
        q = cloud(rng, b, nq) if rng.random() < 0.5 else xyz[:, rng.integers(0, n, nq)] + \
            (rng.standard_normal((b, nq, 3)) * 0.02).astype(np.float32)
        q = np.ascontiguousarray(q, dtype=np.float32)
        radius = float(rng.choice([0.01, 0.05, 0.1, 0.3, 2.0]))
        ns = int(rng.choice([1, 2, 16, 32, 64]))
        k = int(min(n, rng.choice([1, 2, 3, 4, 8, 16, 33, 48])))
        for impl in ("grid", "wave", "basic"):
            os.environ["GEOT_NN_IMPL"] = impl
            check("ball_query/" + impl, p2.ball_query(dev(q), dev(xyz), radius, ns).cpu().numpy(),
                  capi.ball_query(q, xyz, radius, ns), info + " r %g ns %d" % (radius, ns))
            d2, idx = p2.three_nn(dev(q), dev(xyz))
            wd, wi = capi.three_nn(q, xyz)
            if n >= 3:
                check("three_nn idx/" + impl, idx.cpu().numpy(), wi, info)
                check("three_nn d2/" + impl, d2.cpu().numpy(), wd, info)
            kd, ki = knn_sorted(dev(q), dev(xyz), k)
            oi, od = capi.knn_sorted(q, xyz, k)
            check("knn idx/" + impl, ki.cpu().numpy(), oi, info + " k %d" % k)
            check("knn d2/" + impl, kd.cpu().numpy(), od, info + " k %d" % k)
        os.environ.pop("GEOT_NN_IMPL")
        # ---- pointops heap kNN, ragged segments
        sizes, off = ragged(rng, b, 1, n)
        flat = np.concatenate([xyz[i, :sizes[i]] for i in range(b)])
        qs, qoff = ragged(rng, b, 1, nq)
        qflat = np.concatenate([q[i, :qs[i]] for i in range(b)])
        kk = int(rng.choice([1, 2, 5, 16]))
        idx = torch.zeros((len(qflat), kk), dtype=torch.int32, device=DEV)
        d2 = torch.zeros((len(qflat), kk), dtype=torch.float32, device=DEV)
        pops.knnquery_cuda(len(qflat), kk, dev(flat), dev(qflat), dev(off), dev(qoff), idx, d2)
        wi, wd = capi.knnquery_heap(kk, flat, qflat, off, qoff)
        check("knnquery idx", idx.cpu().numpy(), wi, info + " k %d" % kk)
        check("knnquery d2", d2.cpu().numpy(), wd, info + " k %d" % kk)
        bi = torch.zeros((len(qflat), ns), dtype=torch.int32, device=DEV)
        pops.ballquery_cuda(len(qflat), radius, ns, dev(flat), dev(qflat), dev(off), dev(qoff), bi)
        check("ballquery_offset", bi.cpu().numpy(), capi.ballquery_offset(radius, ns, flat, qflat, off, qoff), info)
        # ---- value ops on the oracle's indices
        c = int(rng.choice([1, 3, 16, 33]))
        feats = rng.standard_normal((b, c, n)).astype(np.float32)
        gi = rng.integers(0, n, (b, nq)).astype(np.int32)
        check("gather", p2.gather_points(dev(feats), dev(gi)).cpu().numpy(), capi.gather_points(feats, gi), info)
        gg = rng.integers(0, n, (b, nq, ns)).astype(np.int32)
        check("group", p2.group_points(dev(feats), dev(gg)).cpu().numpy(), capi.group_points(feats, gg), info)
        wt = rng.random((b, nq, 3)).astype(np.float32)
        ti = rng.integers(0, n, (b, nq, 3)).astype(np.int32)
        check("interpolate", p2.three_interpolate(dev(feats), dev(ti), dev(wt)).cpu().numpy(),
              capi.three_interpolate(feats, ti, wt), info)
        if case % 10 == 9:
            print("case %d ok" % case, flush=True)
    print("fuzz: %d cases, every op and implementation bit-identical to the oracle" % a.cases)


if __name__ == "__main__":
    main()
