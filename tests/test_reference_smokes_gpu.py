"""The reference's own executable checks -- its `__main__` smoke blocks -- run through this
package with the same shapes and seeds, and checked against the oracle instead of just printed:

* pointnet2/pointnet2_modules.py:725-744   SA-MSG fwd + bwd on randn(2,9,3), npoint 2, radii [5, 10], nsamples [6, 3]
* openpoints/models/layers/subsample.py:159-185   gather_operation == torch.gather, B 2, N 10000, npoint 4096
* openpoints/models/layers/group.py:355-415   FPS 40960 -> 10000, QueryAndGroup(0.1, 16) on 2 x 40960 points
"""
import numpy as np
import pytest
import torch

from oracle import capi, np_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_sa_msg_smoke_forward_backward():
    from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(1)
    xyz = torch.randn(2, 9, 3).to(DEV).requires_grad_(True)
    feats = torch.randn(2, 9, 6).to(DEV).requires_grad_(True)
    mod = PointnetSAModuleMSG(npoint=2, radii=[5.0, 10.0], nsamples=[6, 3], mlps=[[9, 3], [9, 6]]).to(DEV)
    new_xyz, new_features = mod(xyz, feats)
    assert new_xyz.shape == (2, 2, 3) and new_features.shape == (2, 3 + 6, 2)
    new_features.backward(torch.ones_like(new_features))
    assert xyz.grad is not None and torch.isfinite(xyz.grad).all() and torch.isfinite(feats.grad).all()
    assert feats.grad.abs().sum() > 0
    # the sampled centres are the oracle's FPS picks (origin-skip rule, <= 512-thread tie rule) of each cloud
    want = capi.fps_dense(xyz.detach().cpu().numpy(), 2, 512, True)
    got = torch.gather(xyz.detach(), 1, torch.from_numpy(want).long().to(DEV).unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(new_xyz.detach(), got)
    # radii this large take every point: each scale groups the first nsample indices 0..ns-1 (ball query order)
    for radius, ns in ((5.0, 6), (10.0, 3)):
        idx = np_ref.ball_query(got.cpu().numpy(), xyz.detach().cpu().numpy(), radius, ns)
        d = np.linalg.norm(xyz.detach().cpu().numpy()[:, None, :, :] - got.cpu().numpy()[:, :, None, :], axis=-1)
        if (d < radius).all():
            assert np.array_equal(idx, np.broadcast_to(np.arange(ns, dtype=idx.dtype), idx.shape))


def test_gather_operation_smoke_equals_torch_gather():
    from geot_amd.openpoints.models.layers.subsample import furthest_point_sample, gather_operation
    torch.manual_seed(0)
    B, C, N, npoint = 2, 3, 10000, 4096
    points = torch.randn(B, N, C, device=DEV)
    idx = furthest_point_sample(points, npoint)
    assert idx.shape == (B, npoint) and np.array_equal(idx.cpu().numpy(),
                                                        capi.fps_dense(points.cpu().numpy(), npoint, 1024, False))
    feats = points.transpose(1, 2).contiguous()
    a = gather_operation(feats, idx)
    b = torch.gather(feats, 2, idx.long().unsqueeze(1).expand(-1, C, -1))
    assert torch.equal(a, b)                      # the reference prints torch.allclose(...)
    q = torch.gather(points, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(a.transpose(1, 2), q)


def test_query_and_group_smoke_on_two_clouds_of_40960():
    from geot_amd.openpoints.models.layers.group import QueryAndGroup
    from geot_amd.openpoints.models.layers.subsample import furthest_point_sample
    torch.manual_seed(0)
    B, N, K, npoints = 2, 40960, 16, 10000
    points = torch.randn(B, N, 3, device=DEV)
    idx = furthest_point_sample(points, npoints).to(torch.int64)
    assert np.array_equal(idx.cpu().numpy(), capi.fps_dense(points.cpu().numpy(), npoints, 1024, False))
    query = torch.gather(points, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    grouped_xyz, grouped_feats = QueryAndGroup(0.1, K)(query, points)
    assert grouped_xyz.shape == (B, 3, npoints, K) and grouped_feats is None
    want_idx = capi.ball_query(query.cpu().numpy(), points.cpu().numpy(), 0.1, K)
    rel = np_ref.group_points(points.transpose(1, 2).contiguous().cpu().numpy(), want_idx) \
        - query.transpose(1, 2).cpu().numpy()[:, :, :, None]
    assert np.array_equal(grouped_xyz.cpu().numpy(), rel)
