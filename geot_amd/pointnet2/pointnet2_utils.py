"""autograd.Function front ends over ``pointnet2._ext`` -- same names, argument
order and return conventions as the reference's pointnet2/pointnet2_utils.py
(FurthestPointSampling :48-77, GatherOperation :80-114, ThreeNN :117-146,
ThreeInterpolate :149-203, GroupingOperation :206-254, BallQuery :257-288,
QueryAndGroup :291-373, GroupAll :376-422), so openpoints.models / the decoder
modules can call them unchanged.  Index-producing ops are non-differentiable;
gather / group / three_interpolate have backward passes.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ..ext import pointnet2_ext as _ext


class RandomDropout(nn.Module):
    """pointnet2_utils.py:37-45 -- whole-feature dropout with a random rate in [0, p)."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p
        self.inplace = inplace

    def forward(self, X):
        theta = torch.empty(1).uniform_(0, self.p)[0].item()
        return torch.nn.functional.dropout(X, theta, self.training, self.inplace) * (1 - theta) \
            if self.training else X


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) f32 -> (B,npoint) i32, first pick = index 0; points with
        |p|^2 <= 1e-3 are never picked (reference quirk, sampling_gpu.cu:103-104)."""
        inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) i32 -> (B,C,npoint)."""
        ctx.for_backwards = (idx, features.size(1), features.size(2))
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, _, n = ctx.for_backwards
        return _ext.gather_points_grad(grad_out.contiguous(), idx, n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2 distance, idx (B,n,3) i32)."""
        dist2, idx = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx (B,n,3) i32, weight (B,n,3) -> (B,c,n)."""
        ctx.three_interpolate_for_backward = (idx, weight, features.size(2))
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, m), None, None


three_interpolate = ThreeInterpolate.apply


class FPInterpolateConcat(Function):
    """PointnetFPModule front end (pointnet2_modules.py:619-626) as one op: three_nn -> inverse-distance weights
    -> three_interpolate -> cat([interpolated, unknow_feats], 1), without the (B,n,3) weight temporaries and
    without the concat's copy of the interpolated half: the interpolation kernel
    writes straight into the wide tensor, its gradient kernel reads straight from the wide gradient."""

    @staticmethod
    def forward(ctx, unknown, known, unknow_feats, known_feats, skip_first=False):
        """-> cat([interpolated, unknow_feats], 1), or cat([unknow_feats, interpolated], 1) with skip_first
        (openpoints' PointNetFPModule order, pointnetv2.py:141-142); unknow_feats may be None."""
        dist2, idx = _ext.three_nn(unknown.contiguous(), known.contiguous())
        weight = _ext.fp_weights(dist2)
        b, c, m = known_feats.shape
        n = unknown.shape[1]
        cs = 0 if unknow_feats is None else unknow_feats.shape[1]
        off = cs if skip_first else 0                      # first channel of the interpolated block
        wide = torch.empty((b, c + cs, n), dtype=torch.float32, device=known_feats.device)
        _ext.three_interpolate_into(known_feats.contiguous(), idx, weight, wide, off)
        if cs:
            (wide[:, :cs] if skip_first else wide[:, c:]).copy_(unknow_feats)
        ctx.fp = (idx, weight, c, m, cs, off)
        return wide

    @staticmethod
    def backward(ctx, g):
        idx, weight, c, m, cs, off = ctx.fp
        g = g.contiguous()
        g_known = _ext.three_interpolate_grad_from(g, c, idx, weight, m, off) if ctx.needs_input_grad[3] else None
        g_skip = None
        if cs and ctx.needs_input_grad[2]:
            g_skip = g[:, :cs] if off else g[:, c:]
        return None, None, g_skip, g_known, None


fp_interpolate_concat = FPInterpolateConcat.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) i32 -> (B,C,npoint,nsample)."""
        ctx.for_backwards = (idx, features.size(2))
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        return _ext.group_points_grad(grad_out.contiguous(), idx, n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) i32: the first
        nsample indices (ascending) with d^2 < radius^2, first hit pre-filled."""
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """ball_query + grouping (pointnet2_utils.py:291-373).  Returns
    (B, 3+C, npoint, nsample) (+ grouped_xyz / unique counts when asked)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = None
        if self.sample_uniformly:
            # Host-side resampling of duplicate slots (pointnet2_utils.py:333-342).
            unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
            for b in range(idx.shape[0]):
                for r in range(idx.shape[1]):
                    uniq = torch.unique(idx[b, r, :])
                    nu = uniq.shape[0]
                    unique_cnt[b, r] = nu
                    pick = torch.randint(0, nu, (self.nsample - nu,), dtype=torch.long)
                    idx[b, r, :] = torch.cat((uniq, uniq[pick]))
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius
        if features is not None:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """pointnet2_utils.py:376-422: one group holding every point."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        else:
            new_features = grouped_xyz
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features
