"""One-off soak: grid kNN / three_nn / ball query against the brute-force kernels on many clouds."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_cloud  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--trials", type=int, default=12)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
bad = 0
for trial in range(args.trials):
    n = int(rng.choice([5000, 12000, 24000]))
    B = 6
    clouds = []
    for i in range(B):
        x = make_cloud(n, 1000 * args.seed + trial * 10 + i, dup_frac=0.02 * (i % 2))[0]
        if i == 1:
            x = x * np.array([1.0, 1.0, 0.02], np.float32)          # nearly flat
        if i == 2:
            x = (x * 64).round() / 64                               # quantised: many exact ties
        if i == 3:
            x[: n // 3] *= 0.05                                      # a dense core
        if i == 4:                                                   # few distinct locations: more coincident points
            d = int(rng.integers(20, 400))                           # than any k, every distance tied many times
            x = x[rng.integers(0, d, n)]
        if i == 5:                                                   # one tight blob + a few far outliers: one cell
            x = x * 1e-3                                             # holds nearly everything
            x[rng.integers(0, n, 5)] += 3.0
        clouds.append(np.ascontiguousarray(x, dtype=np.float32))
    ref = torch.from_numpy(np.stack(clouds)).cuda()
    q = torch.cat([ref[:, : n // 2], ref[:, : 500] * 1.7 + 0.01], 1).contiguous()
    for k in (3, 8, 33, 48, 64):
        os.environ["GEOT_NN_IMPL"] = "grid"
        dg, ig = knn_sorted(q, ref, k)
        os.environ["GEOT_NN_IMPL"] = "wave"
        db, ib = knn_sorted(q, ref, k)
        if not (torch.equal(ig, ib) and torch.equal(dg, db)):
            bad += 1
            print("kNN MISMATCH trial", trial, "k", k, (ig != ib).sum().item())
    for r, ns in ((0.05, 16), (0.1, 32), (0.25, 64)):
        os.environ["GEOT_NN_IMPL"] = "grid"
        a = p2.ball_query(q, ref, r, ns)
        os.environ["GEOT_NN_IMPL"] = "wave"
        b_ = p2.ball_query(q, ref, r, ns)
        if not torch.equal(a, b_):
            bad += 1
            print("ball MISMATCH trial", trial, r, ns, (a != b_).sum().item())
    print("trial", trial, "n", n, "ok so far" if bad == 0 else "BAD", flush=True)
del os.environ["GEOT_NN_IMPL"]
print("mismatches:", bad)
