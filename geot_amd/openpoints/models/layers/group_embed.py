"""Patch embeddings on the HIP sampling / grouping operators: SubsampleGroup, PointPatchEmbed, P3Embed -- the callers of
furthest_point_sample / QueryAndGroup / KNNGroup in openpoints/models/layers/group_embed.py:14-286 (same constructor
arguments, attribute names, outputs and state_dict keys, so a reference checkpoint loads).  The convolution stacks of the two
embeddings are one builder here; everything index-producing goes through geot_amd's operators."""
import math

import torch
from torch import nn

from .subsample import furthest_point_sample, random_sample
from .group import KNNGroup, QueryAndGroup, get_aggregation_feautres
from .local_aggregation import CHANNEL_MAP, create_convblock2d

_MEAN_POOLS = ('mean', 'avg', 'meanpool', 'avgpool')


def _sampler(name):
    """group_embed.py:39-44, 90-93: 'fps' (also furthest / farthest) or anything holding 'random' / 'rs'."""
    low = name.lower()
    if 'fps' in low or 'furthest' in low or 'farthest' in low:
        return furthest_point_sample
    if 'random' in low or 'rs' in low:
        return random_sample
    raise NotImplementedError(f'{low} is not implemented. Only support fps, random')


def _grouper(kind, group_size, radius, **kwargs):
    low = kind.lower()
    if 'ball' in low or 'query' in low:
        return QueryAndGroup(radius=radius, nsample=group_size, **kwargs)
    if 'knn' in low:
        return KNNGroup(group_size, **kwargs)
    raise NotImplementedError(f'{low} is not implemented. Only support ballquery, knn')


def _pool(reduction):
    if reduction in _MEAN_POOLS:
        return lambda t: torch.mean(t, dim=-1, keepdim=True)
    return lambda t: torch.max(t, dim=-1, keepdim=True)[0]


def _centres(points, idx):
    return torch.gather(points, 1, idx.unsqueeze(-1).expand(-1, -1, 3))


def _centre_features(x, idx):
    return torch.gather(x, 2, idx.unsqueeze(1).expand(-1, x.shape[1], -1))


def _conv_pair(channels, layers, norm_args, act_args, conv_args, bare_tail):
    """Two stacks of 1x1 Conv2d blocks over `channels` (len layers + 1): the first half, then -- after the caller concatenates
    the pooled feature, which doubles channels[layers // 2] -- the second.  The LAST block of the first stack carries no
    norm / activation (group_embed.py:111-116, 242-247); `bare_tail`: nor does the last block of the second (:120-126),
    which P3Embed keeps (:251-257)."""
    half = layers // 2
    first = [create_convblock2d(channels[i], channels[i + 1],
                                norm_args=norm_args if i != half - 1 else None,
                                act_args=act_args if i != half - 1 else None, **conv_args) for i in range(half)]
    channels[half] *= 2
    second = []
    for i in range(half, layers):
        plain = bare_tail and i == layers - 1
        second.append(create_convblock2d(channels[i], channels[i + 1], norm_args=None if plain else norm_args,
                                         act_args=None if plain else act_args, **conv_args))
    return nn.Sequential(*first), nn.Sequential(*second)


class SubsampleGroup(nn.Module):
    """p (B,N,3)[, x (B,C,N)] -> (grouped_p (B,3,G,K), center_p (B,G,3)[, fj (B,C,G,K), center_x (B,C,G,1)])
    (group_embed.py:14-55)."""

    def __init__(self, num_groups=256, group_size=32, subsample='fps', group='ballquery', radius=0.1, **kwargs):
        super().__init__()
        self.num_groups, self.group_size = num_groups, group_size
        self.subsample, self.group = subsample, group
        self.grouper = _grouper(group, group_size, radius)

    def forward(self, p, x=None):
        idx = _sampler(self.subsample)(p, self.num_groups).to(torch.int64)
        center_p = _centres(p, idx)
        if x is None:
            return self.grouper(center_p, p)[0], center_p
        grouped_p, fj = self.grouper(center_p, p, x)
        return grouped_p, center_p, fj, _centre_features(x, idx).unsqueeze(-1)


class PointPatchEmbed(nn.Module):
    """One-stage patch embedding (group_embed.py:58-171): sample N * sample_ratio centres, group, conv stack 1, concat the
    group's pooled feature, conv stack 2, pool.  forward(p, x) -> ([p, center_p], [x, out_f (B, embed_dim, G)])."""

    def __init__(self, sample_ratio=0.0625, group_size=32, in_channels=3, layers=4, embed_dim=256, channels=None,
                 subsample='fps', group='ballquery', normalize_dp=False, radius=0.1, feature_type='dp_df',
                 relative_xyz=True, norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'},
                 conv_args={'order': 'conv-norm-act'}, reduction='max', **kwargs):
        super().__init__()
        self.sample_ratio, self.group_size, self.feature_type = sample_ratio, group_size, feature_type
        self.sample_fn = _sampler(subsample)
        self.group = group.lower()
        self.grouper = _grouper(self.group, group_size, radius, relative_xyz=relative_xyz, normalize_dp=normalize_dp)
        width_in = CHANNEL_MAP[feature_type](in_channels)
        if channels is None:
            channels = [width_in] + [embed_dim] * (layers // 2) + [embed_dim * 2] * (layers // 2 - 1) + [embed_dim]
        else:
            channels = [width_in] + list(channels) + [embed_dim]
            layers = len(channels) - 1
        self.conv1, self.conv2 = _conv_pair(channels, layers, norm_args, act_args, conv_args, bare_tail=True)
        self.pool = _pool(reduction)
        self.out_channels = channels[-1]
        self.channel_list = [in_channels, embed_dim]

    def forward(self, p, x=None):
        n = p.shape[1]
        idx = self.sample_fn(p, int(n * self.sample_ratio)).long()
        center_p = _centres(p, idx)
        dp, fj = self.grouper(center_p, p, x)
        kind = self.feature_type
        if kind == 'dp':
            fj = dp
        elif kind == 'dp_fj':
            fj = torch.cat([dp, fj], dim=1)
        elif kind in ('dp_df', 'df'):
            df = fj - _centre_features(x, idx).unsqueeze(-1)
            fj = torch.cat([dp, df], dim=1) if kind == 'dp_df' else df
        fj = self.conv1(fj)
        fj = torch.cat([self.pool(fj).expand(-1, -1, -1, self.group_size), fj], dim=1)
        return [p, center_p], [x, self.pool(self.conv2(fj)).squeeze(-1)]


class P3Embed(nn.Module):
    """Progressive patch embedding (group_embed.py:174-286): log_scale(1 / sample_ratio) stages, each sampling a quarter of
    the points it receives; the width doubles per stage and ends at embed_dim.  forward(p, f) -> (list of point sets,
    list of features), input first."""

    def __init__(self, sample_ratio=0.0625, scale=4, group_size=32, in_channels=3, layers=4, embed_dim=256,
                 subsample='fps', group='ballquery', normalize_dp=False, radius=0.1, feature_type='dp_df',
                 relative_xyz=True, norm_args={'norm': 'bn1d'}, act_args={'act': 'relu'},
                 conv_args={'order': 'conv-norm-act'}, reduction='max', **kwargs):
        super().__init__()
        self.sample_ratio, self.group_size, self.feature_type = sample_ratio, group_size, feature_type
        self.sample_fn = _sampler(subsample)
        self.group = group.lower()
        self.grouper = _grouper(self.group, group_size, radius, relative_xyz=relative_xyz, normalize_dp=normalize_dp)
        stages = int(math.log(1 / sample_ratio, scale))
        width = int(embed_dim // 2 ** (stages - 1))
        self.convs = nn.ModuleList()
        self.channel_list = [in_channels]
        channels = None
        for _ in range(stages):
            channels = ([CHANNEL_MAP[feature_type](in_channels)] + [width] * (layers // 2) + [width * 2] * (layers // 2 - 1)
                        + [width])
            self.convs.append(nn.ModuleList(_conv_pair(channels, layers, norm_args, act_args, conv_args, bare_tail=False)))
            self.channel_list.append(width)
            in_channels, width = width, width * 2
        self.pool = _pool(reduction)
        self.out_channels = channels[-1]

    def forward(self, p, f=None):
        n = p.shape[1]
        out_p, out_f = [p], [f]
        for first, second in self.convs:
            cur_p, cur_f = out_p[-1], out_f[-1]
            idx = self.sample_fn(cur_p, int(n // 4)).long()
            n = n // 4
            center_p = _centres(cur_p, idx)
            dp, fj = self.grouper(center_p, cur_p, cur_f)
            fj = first(get_aggregation_feautres(center_p, dp, _centre_features(cur_f, idx), fj, self.feature_type))
            fj = torch.cat([self.pool(fj).expand(-1, -1, -1, self.group_size), fj], dim=1)
            out_f.append(self.pool(second(fj)).squeeze(-1))
            out_p.append(center_p)
        return out_p, out_f
