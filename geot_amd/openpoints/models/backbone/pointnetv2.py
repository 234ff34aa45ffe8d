"""Mirror of the openpoints-flavoured SetAbstraction / FeaturePropagation modules
(openpoints/models/backbone/pointnetv2.py:17-100 PointNetSAModuleMSG, :103-146 PointNetFPModule):
FPS (K1', no origin skip) -> gather -> per-scale LocalAggregation -> concat, and
three_nn -> inverse-distance weights -> three_interpolate -> concat skip -> Conv1d stack.
Same constructor arguments, forward signatures and return values."""
import copy
from typing import List

import torch
import torch.nn as nn

from ..layers.subsample import furthest_point_sample, random_sample
from ..layers.upsampling import three_interpolation
from ....pointnet2.pointnet2_utils import fp_interpolate_concat
from ..layers.local_aggregation import LocalAggregation, create_convblock1d


class PointNetSAModuleMSG(nn.Module):
    def __init__(self, stride: int, radii: List[float], nsamples: List[int], channel_list: List[List[int]],
                 aggr_args: dict, group_args: dict, conv_args: dict, norm_args: dict, act_args: dict,
                 sampler='fps', use_res=False, query_as_support=False, voxel_size=0.1, **kwargs):
        super().__init__()
        self.stride = stride
        self.blocks = len(channel_list)
        self.query_as_support = query_as_support
        if 'fps' in sampler.lower() or 'furthest' in sampler.lower():
            self.sample_fn = furthest_point_sample
        elif 'random' in sampler.lower():
            self.sample_fn = random_sample
        else:
            raise NotImplementedError("sampler %r" % sampler)
        channel_list = [list(c) for c in channel_list]
        self.local_aggregations = nn.ModuleList()
        for i in range(len(radii)):
            channels = channel_list[i]
            if i > 0 and query_as_support:
                channels[0] = channel_list[i - 1][-1]
            ga = copy.deepcopy(dict(group_args))
            ga['radius'], ga['nsample'] = radii[i], nsamples[i]
            self.local_aggregations.append(
                LocalAggregation(channels, aggr_args, conv_args, norm_args, act_args, ga, use_res))

    def forward(self, support_xyz, support_features=None, query_xyz=None):
        """support_xyz (B,N,3), support_features (B,C,N) -> (query_xyz (B,N/stride,3), (B, sum C_out, N/stride))."""
        new_features_list = []
        if query_xyz is None and self.stride > 1:
            idx = self.sample_fn(support_xyz, support_xyz.shape[1] // self.stride).long()
            query_xyz = torch.gather(support_xyz, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
        else:
            query_xyz = support_xyz
            idx = None
        for i in range(self.blocks):
            new_features = self.local_aggregations[i](query_xyz, support_xyz, support_features, query_idx=idx)
            new_features_list.append(new_features)
            if self.query_as_support:
                support_xyz = query_xyz
                support_features = new_features
                idx = None
        return query_xyz, torch.cat(new_features_list, dim=1)


class PointNetFPModule(nn.Module):
    def __init__(self, mlp: List[int], norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'}):
        super().__init__()
        self.convs = nn.Sequential(*[create_convblock1d(mlp[i], mlp[i + 1], norm_args=norm_args, act_args=act_args)
                                     for i in range(len(mlp) - 1)])
        self.fused_front_end = True     # False: the reference's op-by-op chain (kept as the parity baseline)

    def forward(self, unknown, known, unknow_feats, known_feats):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n) or None, known_feats (B,C2,m) -> (B,mlp[-1],n)."""
        if known is not None and self.fused_front_end and known.shape[1] > 0:
            # three_nn -> weights -> interpolate -> cat([skip, interpolated]) as one op (same values)
            return self.convs(fp_interpolate_concat(unknown, known, unknow_feats, known_feats, True))
        if known is not None:
            interpolated_feats = three_interpolation(unknown, known, known_feats)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            new_features = torch.cat([unknow_feats, interpolated_feats], dim=1)
        else:
            new_features = interpolated_feats
        return self.convs(new_features)
