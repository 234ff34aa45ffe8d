"""Mirror of openpoints/models/layers/group.py: KNN :12-28, DenseDilated / DilatedKNN :31-73, GroupingOperation :76-117,
torch_grouping_operation :120-137, GatherOperation :140-174, BallQuery :177-203, QueryAndGroup :206-255,
GroupAll :258-275, KNNGroup :275-320, get_aggregation_feautres, create_grouper :336-352."""
import copy
import logging

import torch
import torch.nn as nn
from torch.autograd import Function

from ...cpp import pointnet2_cuda
from .subsample import GatherOperation, gather_operation  # noqa: F401  (same op, defined twice in the reference)
from .knn import _knn, DenseDilated  # noqa: F401  (group.py:31-54 repeats knn.py's class)


class KNN(nn.Module):
    """group.py:12-28: forward(support (B,N,C), query (B,M,C)) -> (dist, idx (B,M,K) int32)."""

    def __init__(self, neighbors, transpose_mode=True):
        super().__init__()
        self.neighbors = neighbors

    @torch.no_grad()
    def forward(self, support, query):
        dist, idx = _knn(query, support, self.neighbors)      # 3-D: grid / wave kernel; C <= 32: N-D wave kernel
        return dist.transpose(1, 2).contiguous(), idx


class DilatedKNN(nn.Module):
    """group.py:57-73: kNN with k * dilation neighbours, every dilation-th kept."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)
        self.knn = KNN(k * self.dilation, transpose_mode=True)

    def forward(self, query):
        _, idx = self.knn(query, query)
        return self._dilated(idx)


class GroupingOperation(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)."""
        assert features.is_contiguous() and idx.is_contiguous()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, nfeatures, nsample), dtype=torch.float32, device=features.device)
        pointnet2_cuda.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2_cuda.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous().float(), idx,
                                                 grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


def torch_grouping_operation(features, idx):
    """Pure-torch equivalent (group.py:120-137), kept for callers that want int64 indices."""
    all_idx = idx.reshape(idx.shape[0], -1)
    all_idx = all_idx.unsqueeze(1).expand(-1, features.shape[1], -1)
    grouped = features.gather(2, all_idx.long())
    return grouped.reshape(idx.shape[0], features.shape[1], idx.shape[1], idx.shape[2])


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample) int32."""
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2_cuda.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """forward(query_xyz (B,npoint,3), support_xyz (B,N,3), features (B,C,N))
    -> (grouped_xyz (B,3,npoint,nsample), grouped_features (B,C,npoint,nsample) or None)."""

    def __init__(self, radius, nsample, relative_xyz=True, normalize_dp=False, normalize_by_std=False,
                 normalize_by_allstd=False, normalize_by_allstd2=False, return_only_idx=False, **kwargs):
        super().__init__()
        self.radius, self.nsample = radius, nsample
        self.normalize_dp = normalize_dp
        self.normalize_by_std = normalize_by_std
        self.normalize_by_allstd = normalize_by_allstd
        self.normalize_by_allstd2 = normalize_by_allstd2
        assert self.normalize_dp + self.normalize_by_std + self.normalize_by_allstd < 2
        self.relative_xyz = relative_xyz
        self.return_only_idx = return_only_idx

    def forward(self, query_xyz, support_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, support_xyz, query_xyz)
        if self.return_only_idx:
            return idx
        grouped_xyz = grouping_operation(support_xyz.transpose(1, 2).contiguous(), idx)
        if self.relative_xyz:
            grouped_xyz = grouped_xyz - query_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_dp:
                grouped_xyz /= self.radius
        grouped_features = grouping_operation(features, idx) if features is not None else None
        return grouped_xyz, grouped_features


class GroupAll(nn.Module):
    def forward(self, new_xyz, xyz, features=None):
        return xyz.transpose(1, 2).unsqueeze(2), (features.unsqueeze(2) if features is not None else None)


class KNNGroup(nn.Module):
    def __init__(self, nsample, relative_xyz=True, normalize_dp=False, return_only_idx=False, **kwargs):
        super().__init__()
        self.nsample = nsample
        self.knn = KNN(nsample, transpose_mode=True)
        self.relative_xyz = relative_xyz
        self.normalize_dp = normalize_dp
        self.return_only_idx = return_only_idx

    def forward(self, query_xyz, support_xyz, features=None):
        _, idx = self.knn(support_xyz, query_xyz)
        if self.return_only_idx:
            return idx
        idx = idx.int()
        grouped_xyz = grouping_operation(support_xyz.transpose(1, 2).contiguous(), idx)
        if self.relative_xyz:
            grouped_xyz -= query_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_dp:
            grouped_xyz /= torch.amax(torch.sqrt(torch.sum(grouped_xyz ** 2, dim=1)), dim=(1, 2)).view(-1, 1, 1, 1)
        return grouped_xyz, (grouping_operation(features, idx) if features is not None else None)


def get_aggregation_feautres(p, dp, f, fj, feature_type='dp_fj'):
    if feature_type == 'dp_fj':
        return torch.cat([dp, fj], 1)
    df = fj - f.unsqueeze(-1)
    if feature_type == 'dp_fj_df':
        return torch.cat([dp, fj, df], 1)
    if feature_type == 'pi_dp_fj_df':
        return torch.cat([p.transpose(1, 2).unsqueeze(-1).expand(-1, -1, -1, df.shape[-1]), dp, fj, df], 1)
    if feature_type == 'dp_df':
        return torch.cat([dp, df], 1)
    return fj


def create_grouper(group_args):
    args = copy.deepcopy(dict(group_args))
    method = args.pop('NAME', 'ballquery')
    radius = args.pop('radius', 0.1)
    nsample = args.pop('nsample', 20)
    logging.info(group_args)
    if nsample is None:
        return GroupAll()
    if method == 'ballquery':
        return QueryAndGroup(radius, nsample, **args)
    if method == 'knn':
        return KNNGroup(nsample, **args)
    raise ValueError("unknown grouper %r" % method)
