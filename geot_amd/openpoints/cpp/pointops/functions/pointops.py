"""Mirror of openpoints/cpp/pointops/functions/pointops.py (offset-batched, channels-last ops):
FurthestSampling :10-29, KNNQuery :32-50, BallQuery :53-70, Grouping :73-103, querygroup :106-148,
queryandgroup :151-172, Subtraction :175-206, Aggregation :209-242, interpolation :245-259,
Interpolation :262-299 -- same names, arguments and outputs, over the HIP ``pointops_cuda``."""
import torch
from torch.autograd import Function

from .....ext import pointops_cuda
from .....pointops.functions.pointops import _segments


def _new(shape, dtype, like, zero=True):
    return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=like.device)


class FurthestSampling(Function):
    @staticmethod
    def forward(ctx, xyz, offset, new_offset):
        """xyz (n,3), offset (b), new_offset (b) -> idx (m) int32, global indices."""
        assert xyz.is_contiguous()
        n_max, m_total = _segments(offset, new_offset)
        idx = _new((m_total,), torch.int32, xyz)
        tmp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
        pointops_cuda.furthestsampling_cuda(offset.shape[0], n_max, xyz, offset, new_offset, tmp, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


furthestsampling = FurthestSampling.apply


class KNNQuery(Function):
    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz, offset, new_offset):
        """-> (idx (m,nsample) int32 global, dist (m,nsample) L2)."""
        if new_xyz is None:
            new_xyz = xyz
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        m = new_xyz.shape[0]
        idx = _new((m, nsample), torch.int32, xyz)
        dist2 = _new((m, nsample), torch.float32, xyz)
        pointops_cuda.knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2)
        ctx.mark_non_differentiable(idx)
        return idx, torch.sqrt(dist2)

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


knnquery = KNNQuery.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz, offset, new_offset):
        """-> idx (m,nsample) int32 global; queries without a neighbour get zeros (reference quirk)."""
        if new_xyz is None:
            new_xyz = xyz
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        m = new_xyz.shape[0]
        idx = _new((m, nsample), torch.int32, xyz)
        pointops_cuda.ballquery_cuda(m, radius, nsample, xyz, new_xyz, offset, new_offset, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None, None


ballquery = BallQuery.apply


class Grouping(Function):
    @staticmethod
    def forward(ctx, input, idx):
        """input (n,c), idx (m,nsample) -> (m,nsample,c)."""
        assert input.is_contiguous() and idx.is_contiguous()
        m, nsample = idx.shape
        n, c = input.shape
        out = _new((m, nsample, c), torch.float32, input, zero=False)
        pointops_cuda.grouping_forward_cuda(m, nsample, c, input, idx, out)
        ctx.n = n
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        idx, = ctx.saved_tensors
        m, nsample, c = grad_output.shape
        grad_input = _new((ctx.n, c), torch.float32, grad_output)
        pointops_cuda.grouping_backward_cuda(m, nsample, c, grad_output.contiguous(), idx, grad_input)
        return grad_input, None


grouping = Grouping.apply


def querygroup(nsample, xyz, new_xyz, feat, offset, new_offset, radius=None, query_method='knn',
               normalize_dp=False, idx=None):
    """kNN- or ball-query then gather relative xyz and features: -> (grouped_xyz (m,ns,3), grouped_feat (m,ns,c))."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is not None:
        return None
    if nsample is None:
        return xyz.transpose(1, 2).unsqueeze(2), (feat.unsqueeze(2) if feat is not None else None)
    if query_method in ('knn', 'knnquery'):
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    else:
        idx = ballquery(radius, nsample, xyz, new_xyz, offset, new_offset)
    flat = idx.flatten().long()
    m = new_xyz.shape[0]
    grouped_xyz = xyz[flat, :].view(m, nsample, 3)
    grouped_xyz -= new_xyz.unsqueeze(1)
    if normalize_dp:
        scale = (grouped_xyz.norm(dim=-1, p=2, keepdim=True).max(dim=-1, keepdim=True)[0] + 1.0e-8) \
            if query_method == 'knn' else radius
        grouped_xyz /= scale
    grouped_feat = feat[flat, :].view(m, nsample, feat.shape[1]) if feat is not None else None
    return grouped_xyz, grouped_feat


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True):
    """-> (m, nsample, 3+c) (or (m,nsample,c) without xyz)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    m, c = new_xyz.shape[0], feat.shape[1]
    flat = idx.view(-1).long()
    grouped_xyz = xyz[flat, :].view(m, nsample, 3) - new_xyz.unsqueeze(1)
    grouped_feat = feat[flat, :].view(m, nsample, c)
    return torch.cat((grouped_xyz, grouped_feat), -1) if use_xyz else grouped_feat


class Subtraction(Function):
    @staticmethod
    def forward(ctx, input1, input2, idx):
        """input1/2 (n,c), idx (n,nsample) -> input1[:,None,:] - input2[idx] (n,nsample,c)."""
        assert input1.is_contiguous() and input2.is_contiguous()
        n, c = input1.shape
        nsample = idx.shape[-1]
        out = _new((n, nsample, c), torch.float32, input1, zero=False)
        pointops_cuda.subtraction_forward_cuda(n, nsample, c, input1, input2, idx, out)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        idx, = ctx.saved_tensors
        n, nsample, c = grad_output.shape
        g1 = _new((n, c), torch.float32, grad_output)
        g2 = _new((n, c), torch.float32, grad_output)
        pointops_cuda.subtraction_backward_cuda(n, nsample, c, idx, grad_output.contiguous(), g1, g2)
        return g1, g2, None


subtraction = Subtraction.apply


class Aggregation(Function):
    @staticmethod
    def forward(ctx, input, position, weight, idx):
        """input (n,c), position (n,ns,c), weight (n,ns,c'), idx (n,ns) -> (n,c)."""
        assert input.is_contiguous() and position.is_contiguous() and weight.is_contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        out = _new((n, c), torch.float32, input)
        pointops_cuda.aggregation_forward_cuda(n, nsample, c, w_c, input, position, weight, idx, out)
        ctx.save_for_backward(input, position, weight, idx)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, position, weight, idx = ctx.saved_tensors
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        gi, gp, gw = torch.zeros_like(input), torch.zeros_like(position), torch.zeros_like(weight)
        pointops_cuda.aggregation_backward_cuda(n, nsample, c, w_c, input, position, weight, idx,
                                                grad_output.contiguous(), gi, gp, gw)
        return gi, gp, gw, None


aggregation = Aggregation.apply


def _idw(dist):
    r = 1.0 / (dist + 1e-8)
    return r / torch.sum(r, dim=1, keepdim=True)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """Inverse-distance kNN interpolation, torch gather form (pointops.py:245-259)."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
    weight = _idw(dist)
    new_feat = torch.zeros((new_xyz.shape[0], feat.shape[1]), dtype=torch.float32, device=feat.device)
    for i in range(k):
        new_feat += feat[idx[:, i].long(), :] * weight[:, i].unsqueeze(-1)
    return new_feat


class Interpolation(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and input.is_contiguous()
        idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
        weight = _idw(dist).contiguous()
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        out = _new((n, c), torch.float32, input)
        pointops_cuda.interpolation_forward_cuda(n, c, k, input, idx, weight, out)
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        idx, weight = ctx.saved_tensors
        n, c = grad_output.shape
        gi = _new((ctx.m, c), torch.float32, grad_output)
        pointops_cuda.interpolation_backward_cuda(n, c, ctx.k, grad_output.contiguous(), idx, weight, gi)
        return None, None, gi, None, None, None


interpolation2 = Interpolation.apply
