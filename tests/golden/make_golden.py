"""Regenerate the committed golden vectors (run in the BUILD container only).

    python tests/golden/make_golden.py

* config1_*.npz   -- BASELINE.json configs[0]: one 4096-point cloud, FPS -> 1024 under the
                     three reference tie rules + ball_query(r=0.1, nsample=32) from the
                     K1 centres.  Expected outputs come from the C oracle and are
                     asserted equal to the independent numpy restatement before writing.
* knn_point_ref.npz -- outputs of the REFERENCE's own torch-only ``knn_point``
                     (openpoints/models/layers/knn.py:7-20), imported by file path from
                     /root/reference (read-only; the source itself never enters this repo).
* shared_mlp_ref.npz -- outputs + parameters of the reference's ``SharedMLP``
                     (pointnet2/pytorch_utils.py:8-33), imported by file path, on a seeded
                     grouped tensor: pins layer order / bias / BN placement of the SA MLP.
The fixtures hold inputs and expected outputs only.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from geot_amd.synth import make_batch  # noqa: E402
from oracle import capi, np_ref  # noqa: E402

REF = "/root/reference"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def config1(tag, dup_frac):
    xyz, _ = make_batch(1, 4096, dup_frac=dup_frac)
    k1 = capi.fps_dense(xyz, 1024, 512, True)
    k1p = capi.fps_dense(xyz, 1024, 1024, False)
    flat = xyz.reshape(-1, 3)
    k2 = capi.fps_offset(flat, np.array([4096]), np.array([1024]))
    assert (k1 == np_ref.fps_dense(xyz, 1024, 512, True)).all()
    assert (k1p == np_ref.fps_dense(xyz, 1024, 1024, False)).all()
    assert (k2 == np_ref.fps_offset(flat, np.array([4096]), np.array([1024]))).all()
    centres = np.take_along_axis(xyz, k1[..., None].astype(np.int64).repeat(3, -1), 1)
    bq = capi.ball_query(centres, xyz, 0.1, 32)
    assert (bq == np_ref.ball_query(centres, xyz, 0.1, 32)).all()
    np.savez_compressed(os.path.join(HERE, "config1_%s.npz" % tag), xyz=xyz, fps_k1=k1, fps_k1p=k1p,
                        fps_k2=k2, ball_query=bq, radius=np.float32(0.1), nsample=np.int32(32))


def knn_point_ref():
    import torch
    mod = _load(os.path.join(REF, "openpoints/models/layers/knn.py"), "geot_ref_knn")
    xyz, _ = make_batch(2, 2048, start_index=10, origin_pts=0)
    q = torch.from_numpy(xyz)
    dist, idx = mod.knn_point(9, q, q)
    sub = q[:, :256].contiguous()
    dist2, idx2 = mod.knn_point(33, sub, q)
    np.savez_compressed(os.path.join(HERE, "knn_point_ref.npz"), xyz=xyz, k=np.int32(9),
                        idx=idx.numpy().astype(np.int32), dist=dist.numpy(),
                        idx_sub33=idx2.numpy().astype(np.int32), dist_sub33=dist2.numpy())


def shared_mlp_ref():
    import torch
    mod = _load(os.path.join(REF, "pointnet2/pytorch_utils.py"), "geot_ref_pt_utils")
    torch.manual_seed(1609)
    mlp = mod.SharedMLP([6, 16, 16, 32], bn=True).eval()
    with torch.no_grad():
        for m in mlp.modules():  # non-trivial BN statistics so the eval-mode fold is exercised
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.5, 0.5)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
        x = torch.randn(2, 6, 24, 8)
        y = mlp(x)
        pooled = torch.nn.functional.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1)
    sd = {k.replace(".", "__"): v.numpy() for k, v in mlp.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "shared_mlp_ref.npz"), x=x.numpy(), y=y.numpy(),
                        pooled=pooled.numpy(), keys=np.array(list(mlp.state_dict().keys())), **sd)


if __name__ == "__main__":
    capi.build()
    config1("plain", 0.0)
    config1("dup1pct", 0.01)
    knn_point_ref()
    shared_mlp_ref()
    print("golden vectors written to", HERE)
