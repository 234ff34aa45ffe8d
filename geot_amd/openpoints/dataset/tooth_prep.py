"""The per-scan preparation of openpoints/dataset/tooth_semi/tooth_dataset.py:108-147 on the GPU: ``pc_norm``
(centroid, max-norm scale), the random-choice gather of ``sample_points_num`` vertices with their labels, and
the class-weight histogram -- so that a raw scan uploaded once (~1e5 vertices) turns into the training sample
without a CPU pass per item.

The random indices themselves stay the caller's (the reference draws them with ``np.random.choice``,
tooth_dataset.py:134-135; pass the same array for identical samples).  fp32 results agree with the numpy code
to 1e-5 relative (the centroid is accumulated in fp64 here; numpy keeps a running fp32 sum).
"""
import numpy as np
import torch

from ... import _lib
from ...ext._common import call, f32, i32, need, ptr


def _stats(points):
    n = points.shape[0]
    stats = torch.empty(4, dtype=torch.float32, device=points.device)
    nbytes = int(_lib.load().geot_pc_norm_ws_bytes())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    call("geot_pc_norm_stats", points.device, n, ptr(points), ptr(stats), ptr(ws), nbytes)
    return stats


def pc_norm(pc):
    """pc (N,3) CUDA float -> (normalised (N,3), centroid (3,), scale ()) -- tooth_dataset.py:108-114."""
    pc = f32(pc, "pc", 2)
    need(pc.shape[1] == 3 and pc.shape[0] >= 1, "pc must be (N>=1, 3)")
    stats = _stats(pc)
    out = torch.empty_like(pc)
    hist = torch.empty(1, dtype=torch.int32, device=pc.device)
    call("geot_cloud_sample", pc.device, pc.shape[0], pc.shape[0], 0, ptr(pc), None, None, ptr(stats), ptr(out), None,
         None, ptr(hist))
    return out, stats[:3], stats[3]


def prepare_sample(points, labels, selected_idxs, num_classes=17, check=True):
    """points (N,3) float, labels (N,) int32, selected_idxs (m,) int64 (CUDA tensors or numpy).
    -> dict(pos (m,3) float32, y (m,) int64, class_weights (num_classes,), center (3,), scale ())
    = tooth_dataset.py:129-147 for one scan.  check=True reads back the out-of-range flag (numpy raises
    IndexError for a bad index; one host sync)."""
    dev = points.device if isinstance(points, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())

    def up(a, dt):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=dev, dtype=dt).contiguous()

    pts = f32(up(points, torch.float32), "points", 2)
    need(pts.shape[1] == 3 and pts.shape[0] >= 1, "points must be (N>=1, 3)")
    labs = i32(up(labels, torch.int32), "labels", 1)
    need(labs.shape[0] == pts.shape[0], "labels must have one entry per point")
    sel = up(selected_idxs, torch.int64)
    need(sel.dim() == 1, "selected_idxs must be 1-D")
    n, m = pts.shape[0], sel.shape[0]
    stats = _stats(pts)
    pos = torch.empty((m, 3), dtype=torch.float32, device=dev)
    y = torch.empty(m, dtype=torch.int64, device=dev)
    w = torch.empty(num_classes, dtype=torch.float32, device=dev)
    hist = torch.empty(num_classes + 1, dtype=torch.int32, device=dev)
    call("geot_cloud_sample", dev, n, m, int(num_classes), ptr(pts), ptr(labs), ptr(sel), ptr(stats), ptr(pos), ptr(y),
         ptr(w), ptr(hist))
    if check and int(hist[num_classes].item()):
        raise IndexError("selected_idxs holds an index outside [0, %d)" % n)
    return {"pos": pos, "y": y, "class_weights": w, "center": stats[:3], "scale": stats[3]}
