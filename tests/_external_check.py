"""Child process of tests/test_external_vectors.py: with GEOT_DISTANCE set, compare the oracle (and, with --gpu, the
HIP library of the same arithmetic) with one file of reference-CUDA vectors (tools/emit_reference_vectors.py).
Prints one line per mismatch and "external ok" when every index agrees."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402

path, gpu = sys.argv[1], "--gpu" in sys.argv
g = np.load(path, allow_pickle=False)
bad = []


def check(what, got, want):
    if not np.array_equal(np.asarray(got), np.asarray(want)):
        bad.append(what)
        print("MISMATCH", what, float((np.asarray(got) != np.asarray(want)).mean()))


if gpu:
    import torch
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.openpoints.models.layers import subsample
    from geot_amd.pointops.functions import pointops
    dev = torch.device("cuda:0")
for name in [str(c) for c in g["cases"]]:
    xyz, m = g[name + "_xyz"], int(g[name + "_m"])
    b, n, _ = xyz.shape
    k1 = capi.fps_dense(xyz, m, 512, True)
    check(name + " K1 oracle", k1, g[name + "_fps_k1"])
    if name + "_fps_k1p" in g.files:
        check(name + " K1' oracle", capi.fps_dense(xyz, m, 1024, False), g[name + "_fps_k1p"])
    off, noff = (np.arange(1, b + 1) * n).astype(np.int32), (np.arange(1, b + 1) * m).astype(np.int32)
    k2 = capi.fps_offset(xyz.reshape(-1, 3), off, noff)
    check(name + " K2 oracle", xyz.reshape(-1, 3)[k2].reshape(b, m, 3), g[name + "_fps_k2_xyz"])
    centres = np.take_along_axis(xyz, g[name + "_fps_k1"][..., None].astype(np.int64).repeat(3, -1), 1)
    check(name + " ball oracle", capi.ball_query(centres, xyz, 0.1, 32), g[name + "_ball_r0.1_ns32"])
    d2, i3 = capi.three_nn(xyz, centres)
    check(name + " three_nn oracle", i3, g[name + "_three_nn_idx"])
    ki, _ = capi.knnquery_heap(5, xyz.reshape(-1, 3), centres.reshape(-1, 3), off, noff)
    check(name + " knn heap oracle", ki - np.repeat(np.arange(b) * n, m)[:, None], g[name + "_knn5_idx"].reshape(-1, 5))
    if gpu:
        x, c = torch.from_numpy(xyz).to(dev), torch.from_numpy(centres).to(dev)
        check(name + " K1 hip", pu.furthest_point_sample(x, m).cpu().numpy(), g[name + "_fps_k1"])
        if name + "_fps_k1p" in g.files:
            check(name + " K1' hip", subsample.furthest_point_sample(x, m).cpu().numpy(), g[name + "_fps_k1p"])
        check(name + " K2 hip", pointops.fps(x, m).cpu().numpy(), g[name + "_fps_k2_xyz"])
        check(name + " ball hip", pu.ball_query(0.1, 32, x, c).cpu().numpy(), g[name + "_ball_r0.1_ns32"])
        check(name + " three_nn hip", pu.three_nn(x, c)[1].cpu().numpy(), g[name + "_three_nn_idx"])
        check(name + " knn heap hip", pointops.knn(c, x, 5)[0].cpu().numpy(), g[name + "_knn5_idx"])
print("external ok" if not bad else "external FAILED: %d" % len(bad))
sys.exit(0 if not bad else 1)
