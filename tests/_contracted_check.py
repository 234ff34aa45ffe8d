"""Run by tests/test_contracted_gpu.py in a child process with GEOT_DISTANCE=fma (or fma_xy): the index-producing
ops of the contracted-distance library against the oracle twin built with the same GEOT_DISTANCE_MODE -- bit for bit,
as tests/test_parity_gpu.py does for the default build -- and a check that the mode really is a different arithmetic
(some squared distances differ in the last bit from the exact build's formula)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geot_amd import _lib  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.pointnet2 import pointnet2_utils as pu  # noqa: E402
from geot_amd.openpoints.models.layers import subsample  # noqa: E402
from geot_amd.pointops.functions import pointops  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402
from oracle import capi  # noqa: E402

mode = os.environ["GEOT_DISTANCE"]
assert _lib.DISTANCE == mode and capi.DISTANCE == mode and _lib.load().geot_distance_mode() == {"fma": 1, "fma_xy": 2}[mode]
dev = torch.device("cuda:0")
checked = []
for n, m, dup in ((4096, 1024, 0.01), (24000, 2048, 0.0)):
    xyz = make_batch(2, n, start_index=21, dup_frac=dup)[0]
    x = torch.from_numpy(xyz).to(dev)
    # FPS: K1 (cap 512, origin skip), K1' (cap 1024), K2 (offset-batched)
    assert np.array_equal(pu.furthest_point_sample(x, m).cpu().numpy(), capi.fps_dense(xyz, m, 512, True))
    assert np.array_equal(subsample.furthest_point_sample(x, m).cpu().numpy(), capi.fps_dense(xyz, m, 1024, False))
    off = (np.arange(1, 3) * n).astype(np.int32)
    want = capi.fps_offset(xyz.reshape(-1, 3), off, (np.arange(1, 3) * m).astype(np.int32)).reshape(2, m)
    assert np.array_equal(pointops.fps(x, m).cpu().numpy(), xyz.reshape(-1, 3)[want])
    centres = torch.from_numpy(np.take_along_axis(xyz, capi.fps_dense(xyz, m, 512, True)[..., None].astype(np.int64).repeat(3, -1), 1)).to(dev)
    c_np = centres.cpu().numpy()
    # ball query, three_nn, sorted kNN (brute force or grid, whatever the sizes select), pointops heap kNN
    assert np.array_equal(pu.ball_query(0.1, 32, x, centres).cpu().numpy(), capi.ball_query(c_np, xyz, 0.1, 32))
    d, i3 = pu.three_nn(x, centres)
    wd2, wi = capi.three_nn(xyz, c_np)
    assert np.array_equal(i3.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), np.sqrt(wd2))
    d2, ki = knn_sorted(centres, x, 16)
    wi, wd = capi.knn_sorted(c_np, xyz, 16)
    assert np.array_equal(ki.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    d2, ki = knn_sorted(x, x, 9)
    wi, wd = capi.knn_sorted(xyz, xyz, 9)
    assert np.array_equal(ki.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    idx, dist = pointops.knn(centres, x, 5)
    wi, wd = capi.knnquery_heap(5, xyz.reshape(-1, 3), c_np.reshape(-1, 3), off, (np.arange(1, 3) * m).astype(np.int32))
    assert np.array_equal(idx.cpu().numpy().reshape(-1, 5) + np.repeat(np.arange(2) * n, m)[:, None], wi)
    # the mode is a different arithmetic: the exact formula disagrees with these squared distances somewhere
    a, b = xyz[:, :, None, :], c_np[np.arange(2)[:, None, None], capi.three_nn(xyz, c_np)[1]]
    diff = (a - b).astype(np.float32)
    exact = ((diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]).astype(np.float32) + diff[..., 2] * diff[..., 2]).astype(np.float32)
    checked.append(int((exact != wd2).sum()))
assert sum(checked) > 0, "contracted mode produced the exact-mode distances everywhere: is the macro applied?"
print("contracted parity ok:", mode, "three_nn squared distances differing from the un-contracted formula:", checked)
