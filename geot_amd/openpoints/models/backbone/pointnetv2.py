"""Mirror of the openpoints-flavoured SetAbstraction / FeaturePropagation modules
(openpoints/models/backbone/pointnetv2.py:17-100 PointNetSAModuleMSG, :103-146 PointNetFPModule):
FPS (K1', no origin skip) -> gather -> per-scale LocalAggregation -> concat, and
three_nn -> inverse-distance weights -> three_interpolate -> concat skip -> Conv1d stack;
and of the encoder / decoders that stack them (:149-345 PointNet2Encoder, :348-381 PointNet2Decoder, :384-512
PointNet2PartDecoder).
Same constructor arguments, forward signatures, return values and state_dict keys."""
import copy
import logging
from typing import List, Optional

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


def _as_dict(args):
    """group_args arrives as a dict or an attribute bag (the reference writes .radius / .nsample into an EasyDict)"""
    return copy.deepcopy(dict(args)) if args is not None else {}


class PointNet2Encoder(nn.Module):
    """Encoder of PointNet++ / ASSANet (pointnetv2.py:149-345): an optional stem (point-wise conv, optional local
    aggregation at full resolution), then one PointNetSAModuleMSG per stride.  radius / num_samples: a scalar (scaled per
    stage by radius_scaling / nsample_scaling, per block by block_radius_scaling) or per-stage lists; channel sizes from
    `mlps` or from width / layers / width_scaling.  forward -> (list of xyz, list of features), input first."""

    def __init__(self, in_channels: int, radius, num_samples, aggr_args: dict, group_args: dict, conv_args: dict,
                 norm_args: dict, act_args: dict, blocks: Optional[List] = None, mlps=None, width: Optional[int] = None,
                 strides: List[int] = [4, 4, 4, 4], layers: int = 3, width_scaling: int = 2, radius_scaling: int = 2,
                 block_radius_scaling: int = 1, nsample_scaling: int = 1, sampler: str = 'fps', use_res=False,
                 stem_conv=False, stem_aggr=False, double_last_channel=True, query_as_support=False, **kwargs):
        super().__init__()
        if kwargs:
            logging.warning("kwargs: %s are not used in PointNet2Encoder", kwargs)
        self.strides = list(strides)
        self.blocks = list(blocks) if mlps is None else [len(m) for m in mlps]
        self.radius = self._per_block(radius, radius_scaling, block_radius_scaling)
        self.num_samples = self._per_block(num_samples, nsample_scaling, 1)
        self.stem_conv, self.stem_aggr = stem_conv, stem_aggr
        if stem_conv:
            width = width if width is not None else mlps[0][0][0]
            self.conv1 = create_convblock1d(in_channels, width, norm_args=None, act_args=None)
            if stem_aggr:
                ga = _as_dict(group_args)
                ga['radius'], ga['nsample'] = self.radius[0][0], self.num_samples[0][0]
                self.stem = LocalAggregation([width] * (layers + 1), aggr_args, conv_args, norm_args, act_args, ga, use_res)
            in_channels = width
        if mlps is None:
            assert width is not None and layers is not None
            mlps = []
            for i, stride in enumerate(self.strides):
                grown = width * width_scaling if stride > 1 else width
                if double_last_channel:             # the first block of a stage ends on the stage's new width
                    mlps.append([[width] * (layers - 1) + [grown]] + [[grown] * layers] * (self.blocks[i] - 1))
                else:
                    mlps.append([[width] * layers] * self.blocks[i])
                width = grown
        self.mlps = mlps
        self.SA_modules = nn.ModuleList()
        self.channel_list = [in_channels]
        for k, stride in enumerate(self.strides):
            per_block = [[in_channels] + list(m) for m in mlps[k]]
            self.SA_modules.append(PointNetSAModuleMSG(
                stride=stride, radii=self.radius[k], nsamples=self.num_samples[k], channel_list=per_block,
                aggr_args=aggr_args, group_args=_as_dict(group_args), conv_args=conv_args, norm_args=norm_args,
                act_args=act_args, sampler=sampler, use_res=use_res, query_as_support=query_as_support))
            in_channels = sum(m[-1] for m in per_block)      # the blocks' outputs are concatenated
            self.channel_list.append(in_channels)
        self.out_channels = in_channels

    def _per_block(self, param, stage_scaling, block_scaling):
        """radius / nsample as [stage][block] (pointnetv2.py:290-307)"""
        if isinstance(param, (list, tuple)):
            full = []
            for i, value in enumerate(param):
                value = list(value) if isinstance(value, (list, tuple)) else [value]
                full.append(value + [value[-1]] * (self.blocks[i] - len(value)))
            return full
        full = []
        for i, stride in enumerate(self.strides):
            if stride == 1:
                full.append([param] * self.blocks[i])
            else:
                full.append([param] + [param * block_scaling] * (self.blocks[i] - 1))
                param = param * stage_scaling
        return full

    def _stem(self, xyz, features):
        if hasattr(xyz, 'keys'):
            xyz, features = xyz['pos'], xyz['x']
        if features is None:
            features = xyz.clone().transpose(1, 2).contiguous()
        xyz = xyz.contiguous()
        if self.stem_conv:
            features = self.conv1(features)
        if self.stem_aggr:
            features = self.stem(xyz, xyz, features)
        return xyz, features

    def forward_cls_feat(self, xyz, features=None):
        xyz, features = self._stem(xyz, features)
        for sa in self.SA_modules:
            xyz, features = sa(xyz, features)
        return features.squeeze(-1)

    def forward_seg_feat(self, xyz, features=None):
        xyz, features = self._stem(xyz, features)
        l_xyz, l_features = [xyz], [features]
        for sa in self.SA_modules:
            xyz, features = sa(l_xyz[-1], l_features[-1])
            l_xyz.append(xyz)
            l_features.append(features)
        return l_xyz, l_features

    def forward(self, xyz, features=None):
        return self.forward_seg_feat(xyz, features)


class PointNet2Decoder(nn.Module):
    """Decoder of PointNet++ (pointnetv2.py:348-381): one PointNetFPModule per encoder stage, coarsest first; stage k takes
    the skip of level k and the output of stage k + 1 (the encoder's last features for the coarsest)."""

    def __init__(self, encoder_channel_list: List[int], mlps=None, fp_mlps=None, decoder_layers=1, **kwargs):
        super().__init__()
        skips = list(encoder_channel_list)
        if fp_mlps is None:
            fp_mlps = [[mlps[0][0][0]] * (decoder_layers + 1)] + [[c] * (decoder_layers + 1) for c in skips[1:-1]]
        self.FP_modules = nn.ModuleList()
        for k, widths in enumerate(fp_mlps):
            below = fp_mlps[k + 1][-1] if k + 1 < len(fp_mlps) else skips[-1]
            self.FP_modules.append(PointNetFPModule([below + skips[k]] + list(widths)))
        self.out_channels = fp_mlps[0][-1]

    def forward(self, l_xyz, l_features):
        for i in range(len(self.FP_modules) - 1, -1, -1):
            l_features[i] = self.FP_modules[i](l_xyz[i], l_xyz[i + 1], l_features[i], l_features[i + 1])
        return l_features[0]


class PointNet2PartDecoder(nn.Module):
    """Part-segmentation decoder (pointnetv2.py:384-512): the FP stack of PointNet2Decoder, computed from the ENCODER's
    arguments (it rebuilds the encoder's channel table), with the object class as 16 one-hot channels beside the finest
    level's skip features.  forward(l_xyz, l_features, cls_label (B, 1) int64) -> (B, out_channels, N)."""
    NUM_OBJECT_CLASSES = 16

    def __init__(self, in_channels: int, radius, num_samples, group_args: dict, conv_args: dict, norm_args: dict,
                 act_args: dict, mlps=None, blocks: Optional[List] = None, width: Optional[int] = None, strides=[4, 4, 4, 4],
                 layers=3, fp_mlps=None, decoder_layers=1, decocder_aggr_args=None, width_scaling=2, radius_scaling=2,
                 nsample_scaling=1, use_res=False, stem_conv=False, double_last_channel=False, **kwargs):
        super().__init__()
        if kwargs:
            logging.warning("kwargs: %s are not used in PointNet2PartDecoder", kwargs)
        self.strides = list(strides)
        self.blocks = list(blocks) if mlps is None else [len(m) for m in mlps]
        if stem_conv:
            in_channels = width
        if mlps is None:
            assert width is not None and layers is not None
            mlps = []
            for i, stride in enumerate(self.strides):
                grown = width * (width_scaling if not double_last_channel else 2) if stride > 1 else width
                if double_last_channel:
                    mlps.append([[width] * (layers - 1) + [grown]] + [[grown] * layers] * (self.blocks[i] - 1))
                else:                               # (this table holds output widths only: the stage's new width throughout)
                    mlps.append([[grown] * layers] * self.blocks[i])
                width = grown
        self.mlps = mlps
        skips = [in_channels] + [sum(m[-1] for m in stage) for stage in mlps]
        if fp_mlps is None:
            fp_mlps = [[mlps[0][0][0]] * (decoder_layers + 1)] + [[c] * (decoder_layers + 1) for c in skips[1:-1]]
        skips[0] += self.NUM_OBJECT_CLASSES
        self.FP_modules = nn.ModuleList()
        for k, widths in enumerate(fp_mlps):
            below = fp_mlps[k + 1][-1] if k + 1 < len(fp_mlps) else skips[-1]
            self.FP_modules.append(PointNetFPModule([below + skips[k]] + list(widths)))
        self.out_channels = fp_mlps[0][-1]

    def forward(self, l_xyz, l_features, cls_label):
        for i in range(len(self.FP_modules) - 1, 0, -1):
            l_features[i] = self.FP_modules[i](l_xyz[i], l_xyz[i + 1], l_features[i], l_features[i + 1])
        b, n = l_xyz[0].shape[:2]
        one_hot = torch.zeros((b, self.NUM_OBJECT_CLASSES), device=l_xyz[0].device).scatter_(1, cls_label, 1)
        skip = torch.cat([one_hot.unsqueeze(-1).expand(-1, -1, n), l_features[0]], 1)
        l_features[0] = self.FP_modules[0](l_xyz[0], l_xyz[1], skip, l_features[1])
        return l_features[0]
