"""Mirror of openpoints/dataset/grid_sample.py:4-23 (``grid_subsampling`` over the C++ extension
``openpoints.cpp.subsampling.grid_subsampling.compute``, wrapper.cpp:58-285) on the GPU.

Same call, same return shape conventions (points, then features and/or labels when given).  numpy inputs
are uploaded and numpy comes back, as the reference does; CUDA tensors stay on the device.  Rows come out by
ascending voxel key -- the reference emits the same rows in its hash map's iteration order -- and a label
tie resolves to the smallest tied label (see include/geot_hip.h).  There is no CPU path.
"""
import numpy as np
import torch

from ... import _lib
from ...ext._common import call, f32, i32, need, ptr


def _to_dev(a, dtype, device):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(device=device, dtype=dtype).contiguous()


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1, verbose=0, device=None):
    """points (N,3) float; features (N,d) float, optional; labels (N,) or (N,l) int, optional.
    -> subsampled points [, features][, labels] (labels come back (M,l), l = 1 for a flat input: wrapper.cpp
    returns a 2-D array there too)."""
    as_numpy = isinstance(points, np.ndarray)
    if device is None:
        device = points.device if isinstance(points, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())
    pts = f32(_to_dev(points, torch.float32, device), "points", 2)
    need(pts.shape[1] == 3, "points must be (N, 3)")
    n = pts.shape[0]
    need(n >= 1, "points must not be empty")
    need(float(sampleDl) > 0, "sampleDl must be positive")
    feats = _to_dev(features, torch.float32, device)
    labs = _to_dev(labels, torch.int32, device)
    if feats is not None:
        feats = f32(feats.reshape(n, -1), "features", 2)
    if labs is not None:
        labs = i32(labs.reshape(n, -1), "labels", 2)
    fdim = 0 if feats is None else feats.shape[1]
    ldim = 0 if labs is None else labs.shape[1]
    nbytes = int(_lib.load().geot_grid_subsampling_ws_bytes(n))
    need(nbytes > 0, "grid_subsampling needs a GPU (workspace query failed)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    out_p = torch.empty((n, 3), dtype=torch.float32, device=device)
    out_f = torch.empty((n, fdim), dtype=torch.float32, device=device) if fdim else None
    out_l = torch.empty((n, ldim), dtype=torch.int32, device=device) if ldim else None
    count = torch.zeros(1, dtype=torch.int32, device=device)
    call("geot_grid_subsampling", device, n, fdim, ldim, float(sampleDl), ptr(pts), ptr(feats), ptr(labs), ptr(out_p),
         ptr(out_f), ptr(out_l), ptr(count), ptr(ws), nbytes)
    m = int(count.item())          # the output size is data-dependent: one read-back, as any caller needs it
    res = [out_p[:m]] + ([out_f[:m]] if fdim else []) + ([out_l[:m]] if ldim else [])
    if as_numpy:
        res = [r.cpu().numpy() for r in res]
    return res[0] if len(res) == 1 else tuple(res)
