"""Mirror of the local aggregations the reference's openpoints PointNet++ / ASSANet encoder is built from
(openpoints/models/layers/local_aggregation.py:12-29 CHANNEL_MAP, :32-138 ASSA, :140-242 ConvPool, :246-288
LocalAggregation; conv blocks openpoints/models/layers/conv.py:24-103 in their default
'conv-norm-act' order): the callers of the grouping operators (QueryAndGroup / KNNGroup / GroupAll).

In eval mode without autograd, a ('dp_fj', max-reduction, conv->BN->ReLU, ball-query) ConvPool runs
the fused HIP SetAbstraction kernel (geot_amd/csrc/sa_mlp.hip): the (B, 3+C, npoint, nsample) grouped
tensor and every intermediate activation stay on chip.  Otherwise it composes the individual ops in
the reference's order.
"""
from typing import List

import torch
import torch.nn as nn

from .group import create_grouper, get_aggregation_feautres, ball_query, QueryAndGroup

CHANNEL_MAP = {
    'fj': lambda x: x,
    'df': lambda x: x,
    'assa': lambda x: x * 3,
    'assa_dp': lambda x: x * 3 + 3,
    'dp_fj': lambda x: 3 + x,
    'pj': lambda x: x,
    'dp': lambda x: 3,
    'pi_dp': lambda x: x + 3,
    'pj_dp': lambda x: x + 3,
    'dp_fj_df': lambda x: x * 2 + 3,
    'dp_fi_df': lambda x: x * 2 + 3,
    'pi_dp_fj_df': lambda x: x * 2 + 6,
    'pj_dp_fj_df': lambda x: x * 2 + 6,
    'pj_dp_df': lambda x: x + 6,
    'dp_df': lambda x: x + 3,
}


def create_norm(norm_args, channels, dimension=None):
    """norm_args: None | str | {'norm': name, **kwargs}; names bn / bn1d / bn2d / in / in1d / in2d."""
    if norm_args is None:
        return None
    args = dict(norm_args) if not isinstance(norm_args, str) else {'norm': norm_args}
    name = args.pop('norm', None)
    if name is None:
        return None
    name = name.lower()
    if dimension is not None and name in ('bn', 'in'):
        name += dimension
    table = {'bn1d': nn.BatchNorm1d, 'bn2d': nn.BatchNorm2d, 'bn': nn.BatchNorm2d,
             'in1d': nn.InstanceNorm1d, 'in2d': nn.InstanceNorm2d, 'in': nn.InstanceNorm2d}
    if name not in table:
        raise NotImplementedError("norm %r is not on the GeoT path" % name)
    return table[name](channels, **args)


def create_act(act_args):
    """act_args: None | str | {'act': name, **kwargs}; relu / leakyrelu / gelu / silu; inplace by default."""
    if act_args is None:
        return None
    args = dict(act_args) if not isinstance(act_args, str) else {'act': act_args}
    name = args.pop('act', None)
    if name is None:
        return None
    name = name.lower()
    if name == 'relu':
        return nn.ReLU(inplace=args.get('inplace', True))
    if name == 'leakyrelu':
        return nn.LeakyReLU(args.get('negative_slope', 0.01), inplace=args.get('inplace', True))
    if name == 'gelu':
        return nn.GELU()
    if name in ('silu', 'swish'):
        return nn.SiLU(inplace=args.get('inplace', True))
    raise NotImplementedError("activation %r is not on the GeoT path" % name)


def _convblock(conv_cls, dimension, cin, cout, norm_args, act_args, **kwargs):
    kwargs = dict(kwargs)
    bias = kwargs.pop('bias', True)
    kwargs.pop('order', None)
    norm = create_norm(norm_args, cout, dimension=dimension)
    layers = [conv_cls(cin, cout, kwargs.pop('kernel_size', 1), bias=False if norm is not None else bias, **kwargs)]
    if norm is not None:
        layers.append(norm)
    act = create_act(act_args)
    if act is not None:
        layers.append(act)
    return nn.Sequential(*layers)


def create_convblock1d(cin, cout, norm_args=None, act_args=None, **kwargs):
    return _convblock(nn.Conv1d, '1d', cin, cout, norm_args, act_args, **kwargs)


def create_convblock2d(cin, cout, norm_args=None, act_args=None, **kwargs):
    return _convblock(nn.Conv2d, '2d', cin, cout, norm_args, act_args, **kwargs)


class ConvPool(nn.Module):
    def __init__(self, channels: List[int], conv_args=None, norm_args=None, act_args=None, group_args=None,
                 feature_type='dp_fj', reduction='mean', use_res=False, use_pooled_as_identity=False,
                 fused_eval=True, **kwargs):
        super().__init__()
        conv_args = dict(conv_args or {})
        channels = list(channels)
        skip_channel = channels[0]
        self.use_res = use_res
        self.use_pooled_as_identity = use_pooled_as_identity
        self.fused_eval = fused_eval
        if use_res:
            self.skipconv = create_convblock1d(skip_channel, channels[-1], norm_args=None, act_args=None,
                                               **conv_args) if skip_channel != channels[-1] else nn.Identity()
        self.feature_type = feature_type
        channels[0] = CHANNEL_MAP[feature_type](channels[0])
        convs = [create_convblock2d(channels[i], channels[i + 1], norm_args=norm_args, act_args=act_args, **conv_args)
                 for i in range(len(channels) - 2)]
        convs.append(create_convblock2d(channels[-2], channels[-1], norm_args=norm_args,
                                        act_args=None if use_res else act_args, **conv_args))
        self.act = create_act(act_args)
        self.convs = nn.Sequential(*convs)
        self.grouper = create_grouper(group_args)
        self.reduction = reduction
        if reduction == 'max':
            self.reduction_layer = lambda x: torch.max(x, dim=-1, keepdim=False)[0]
        elif reduction in ('avg', 'mean'):
            self.reduction_layer = lambda x: torch.mean(x, dim=-1, keepdim=False)
        elif reduction == 'sum':
            self.reduction_layer = lambda x: torch.sum(x, dim=-1, keepdim=False)
        else:
            raise NotImplementedError('reduction %s not implemented' % reduction)

    def _fused_ok(self, features):
        from ....sa_fused import fused_sa_available
        g = self.grouper
        return (self.fused_eval and not self.training and not torch.is_grad_enabled() and features is not None
                and self.feature_type == 'dp_fj' and self.reduction == 'max' and not self.use_res
                and isinstance(g, QueryAndGroup) and g.relative_xyz and not g.return_only_idx
                and not (g.normalize_by_std or g.normalize_by_allstd or g.normalize_by_allstd2)
                and fused_sa_available(self.convs, g.nsample))

    def forward(self, query_xyz, support_xyz, features, query_idx=None):
        if self._fused_ok(features):
            from ....sa_fused import fused_group_mlp_max
            g = self.grouper
            idx = ball_query(g.radius, g.nsample, support_xyz, query_xyz)
            return fused_group_mlp_max(support_xyz.contiguous(), query_xyz.contiguous(), features.contiguous(), idx,
                                       self.convs, 1.0 / g.radius if g.normalize_dp else 1.0)
        dp, fj = self.grouper(query_xyz, support_xyz, features)
        neighbor_dim = 3
        identity = 0
        if 'df' in self.feature_type or self.use_res:
            if self.use_pooled_as_identity:
                features = torch.max(fj, dim=-1, keepdim=False)[0]
            elif query_idx is not None:
                if query_xyz.shape[1] != support_xyz.shape[1]:
                    features = torch.gather(features, -1, query_idx.unsqueeze(1).expand(-1, features.shape[1], -1))
            elif dp.shape[2] == 1:
                neighbor_dim = 2
            if self.use_res and neighbor_dim != 2:
                identity = self.skipconv(features)
        fj = get_aggregation_feautres(query_xyz, dp, features, fj, feature_type=self.feature_type)
        out_features = self.reduction_layer(self.convs(fj))
        if self.use_res:
            out_features = self.act(out_features + identity)
        return out_features


class ASSA(nn.Module):
    """Anisotropic separable set abstraction (local_aggregation.py:32-138): point-wise convolutions, grouping, the three
    relative coordinates times every grouped channel reduced over the neighbourhood, point-wise convolutions, residual.
    `convs` holds both conv stacks under the reference's indices (same state_dict keys).  The reduction never needs the
    (B, 3 C, npoint, nsample) product the reference materialises when it is a sum or a mean: it is one contraction over the
    neighbourhood of the grouper's two outputs (channel a C + c = coordinate a times feature c, the reference's view)."""

    def __init__(self, channels: List[int], conv_args=None, norm_args=None, act_args=None, group_args=None,
                 feature_type='dp_fj', reduction='mean', use_res=True, use_inverted_dims=False):
        super().__init__()
        conv_args = dict(conv_args or {})
        channels = list(channels)
        self.feature_type, self.use_res, self.reduction = feature_type, use_res, reduction
        self.num_preconv = n_pre = -(-(len(channels) - 1) // 2)
        if feature_type == 'assa' and not use_inverted_dims:
            channels[n_pre] = -(-channels[n_pre] // 3)
        convs = [create_convblock1d(channels[i], channels[i + 1], norm_args=norm_args, act_args=act_args, **conv_args)
                 for i in range(n_pre)]
        skip_channels = channels[n_pre]
        channels[n_pre] = CHANNEL_MAP[feature_type](channels[n_pre])
        last = len(channels) - 2
        convs += [create_convblock1d(channels[i], channels[i + 1], norm_args=norm_args,
                                     act_args=None if use_res and i == last else act_args, **conv_args)
                  for i in range(n_pre, len(channels) - 1)]
        self.act = create_act(act_args)
        self.convs = nn.Sequential(*convs)
        if use_res:
            self.skip_layer = nn.Identity() if skip_channels == channels[-1] else nn.Conv1d(skip_channels, channels[-1], 1, bias=False)
        self.grouper = create_grouper(group_args)
        if reduction not in ('max', 'avg', 'mean', 'sum'):
            raise NotImplementedError('reduction %s not implemented' % reduction)

    def forward(self, query_xyz, support_xyz, features, query_idx=None):
        features = self.convs[:self.num_preconv](features)
        dp, fj = self.grouper(query_xyz, support_xyz, features)          # (B, 3, P, S), (B, C, P, S)
        if self.use_res and query_idx is not None:
            features = torch.gather(features, -1, query_idx.unsqueeze(1).expand(-1, features.shape[1], -1))
        b, c, npoint, nsample = fj.shape
        if self.reduction == 'max':
            prod = (fj.unsqueeze(1) * dp.unsqueeze(2)).reshape(b, 3 * c, npoint, nsample)
            out = prod.max(dim=-1)[0]
        else:
            out = torch.einsum('baps,bcps->bacp', dp, fj).reshape(b, 3 * c, npoint)
            if self.reduction != 'sum':
                out = out / nsample
        out = self.convs[self.num_preconv:](out)
        if self.use_res:
            out = self.act(out + self.skip_layer(features))
        return out


class LocalAggregation(nn.Module):
    def __init__(self, channels: List[int], aggr_args: dict, conv_args=None, norm_args=None, act_args=None,
                 group_args=None, use_res=False):
        super().__init__()
        aggr_args = dict(aggr_args or {})
        aggr_type = aggr_args.get('NAME', 'convpool').lower()
        feature_type, reduction = aggr_args.get('feature_type', 'dp_fj'), aggr_args.get('reduction', 'max')
        if aggr_type == 'convpool':
            self.SA_CONFIG_operator = ConvPool(channels, conv_args, norm_args, act_args, group_args, feature_type, reduction,
                                               use_res, aggr_args.get('use_pooled_as_identity', False))
        elif aggr_type == 'assa':
            self.SA_CONFIG_operator = ASSA(channels, conv_args, norm_args, act_args, group_args, feature_type, reduction,
                                           use_res, aggr_args.get('use_inverted_dims', False))
        else:
            raise NotImplementedError('LocalAggregation %s not implemented' % aggr_type)

    def forward(self, query_xyz, support_xyz, support_features, query_idx=None):
        return self.SA_CONFIG_operator(query_xyz, support_xyz, support_features, query_idx)
