"""Set-abstraction / feature-propagation modules over the HIP ops -- the callers of the
hot path that BASELINE configs[1] measures (pointnet2/pointnet2_modules.py:
PointnetSAModuleVotes :273-380, PointnetSAModule(MSG) :75-158, PointnetSAModuleMSGVotes :500-579, PointnetFPModule :582-642,
PointnetLFPModuleMSG :644-722).

Same constructor arguments, forward signatures and return tuples as the reference.
In eval mode, with max pooling and a plain conv->BN->ReLU stack, the SA module runs the
fused HIP kernel (grouping + centre subtraction + MLP on fp32 MFMA + max over nsample,
geot_amd/csrc/sa_mlp.hip) so the (B, C, npoint, nsample) tensors never touch HBM; in
training mode (batch statistics) it composes the individual ops exactly as the reference.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


def _pool(new_features, grouped_xyz, pooling, sigma, nsample):
    if pooling == "max":
        new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)])
    elif pooling == "avg":
        new_features = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
    elif pooling == "rbf":
        rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (sigma ** 2) / 2)
        new_features = torch.sum(new_features * rbf.unsqueeze(1), -1, keepdim=True) / float(nsample)
    return new_features.squeeze(-1)


def _plain_post_act(mlp):
    """SharedMLP stages of the form conv(1x1, no bias) -> BatchNorm -> ReLU (what the SA modules build with bn=True)."""
    for stage in mlp.children():
        names = [n for n, _ in stage.named_children()]
        if names != ["conv", "bn", "activation"] or stage.conv.bias is not None or not isinstance(stage.activation, nn.ReLU):
            return False
    return len(list(mlp.children())) >= 1


def _sa_factored(xyz_flipped, new_xyz, features, idx, mlp, xyz_scale):
    """Training-mode body of a SetAbstraction module (group xyz, centre, group features, concat, SharedMLP, max:
    pointnet2_modules.py:31-72 + pointnet2_utils.py:314-373) with the first 1x1 convolution moved in front of the
    grouping: W1.[(xyz_j - c_i) s ; f_j] = (W1.[s xyz ; f])[j] - (W1x s).c_i = P[idx] + Q.  The GEMM runs over the N
    points instead of the npoint x nsample rows (32x fewer at nsample = 32) and neither grouped tensor of the
    reference ((B,3,np,ns), (B,C,np,ns)) nor their concatenation is built; BatchNorm + ReLU of every stage is one
    fused pass each way (fused_norm.bn_act).  Same function as the composed path (tests: 1e-4)."""
    from ..fused_norm import bn_act, max_last, add_last_broadcast, bn_relu_max
    stages = list(mlp.children())
    conv = stages[0].conv
    w = conv.weight.view(conv.out_channels, -1)
    pts = torch.cat([xyz_flipped * xyz_scale, features], dim=1)                    # (B, 3 + C, N)
    p = pt_utils.pointwise(w, pts)                                                 # (B, C1, N)
    q = pt_utils.pointwise(w[:, :3] * (-xyz_scale), new_xyz.transpose(1, 2).contiguous())   # (B, C1, np)
    b, c1, npoint = q.shape
    ns = idx.shape[2]
    y = add_last_broadcast(pointnet2_utils.grouping_operation(p.contiguous(), idx), q).view(b, c1, npoint * ns)
    y = bn_act(stages[0].bn.bn, y, relu=True)
    last = stages[-1]
    names = [n for n, _ in last.named_children()]
    if len(stages) > 1 and names == ["conv", "bn", "activation"] and isinstance(last.activation, nn.ReLU):
        # the last stage's BatchNorm -> ReLU -> max over nsample without the normalised (B, C, np, ns) tensor
        y = pt_utils.shared_mlp_nd(stages[1:-1], y)
        y = pt_utils.conv1x1(last.conv, y) if not isinstance(last.conv, (pt_utils.PointwiseConv1d, pt_utils.PointwiseConv2d)) \
            else last.conv(y)
        return bn_relu_max(last.bn.bn, y.view(b, y.shape[1], npoint * ns), ns)
    y = pt_utils.shared_mlp_nd(stages[1:], y)
    return max_last(y.view(b, y.shape[1], npoint, ns))


class PointnetSAModuleVotes(nn.Module):
    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pooling: str = "max", sigma: float = None,
                 normalize_xyz: bool = False, sample_uniformly: bool = False, ret_unique_cnt: bool = False,
                 fused_eval: bool = True):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else (radius / 2 if radius is not None else None)
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        self.fused_eval = fused_eval
        self.factored_train = True      # training: first 1x1 conv evaluated per POINT, then gathered (see _sa_factored)
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                         normalize_xyz=normalize_xyz,
                                                         sample_uniformly=sample_uniformly,
                                                         ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        mlp_spec = list(mlp)
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        """xyz (B,N,3), features (B,C,N) -> (new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds)."""
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        if self.npoint is None:
            inds = None                     # GroupAll: nothing is sampled (the reference would hand None to its FPS here)
        elif inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        else:
            assert inds.shape[1] == self.npoint
        new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous() \
            if self.npoint is not None else None

        from ..sa_fused import fused_sa_available, fused_group_mlp_max
        if (self.fused_eval and not self.training and not torch.is_grad_enabled() and self.npoint is not None
                and self.pooling == "max" and self.use_xyz and not self.sample_uniformly
                and features is not None and fused_sa_available(self.mlp_module, self.nsample)):
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
            new_features = fused_group_mlp_max(xyz, new_xyz, features.contiguous(), idx, self.mlp_module,
                                               1.0 / self.radius if self.normalize_xyz else 1.0)
            return new_xyz, new_features, inds

        if (self.factored_train and torch.is_grad_enabled() and self.npoint is not None and self.pooling == "max"
                and self.use_xyz and not self.sample_uniformly and not self.ret_unique_cnt and features is not None
                and xyz.is_cuda and _plain_post_act(self.mlp_module)):
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
            return new_xyz, _sa_factored(xyz_flipped, new_xyz, features, idx, self.mlp_module,
                                         1.0 / self.radius if self.normalize_xyz else 1.0), inds

        if not self.ret_unique_cnt:
            grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)
        else:
            grouped_features, grouped_xyz, unique_cnt = self.grouper(xyz, new_xyz, features)
        new_features = self.mlp_module(grouped_features)
        new_features = _pool(new_features, grouped_xyz, self.pooling, self.sigma, self.nsample)
        if not self.ret_unique_cnt:
            return new_xyz, new_features, inds
        return new_xyz, new_features, inds, unique_cnt


class PointnetSAModuleMSG(nn.Module):
    """Multi-scale grouping SA (pointnet2_modules.py:75-121 + base forward :31-72)."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                               sample_uniformly=sample_uniformly)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn))

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None):
        """-> (new_xyz (B,npoint,3), new_features (B, sum_k mlps[k][-1], npoint))."""
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        new_xyz = pointnet2_utils.gather_operation(
            xyz_flipped, pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        ).transpose(1, 2).contiguous() if self.npoint is not None else None
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            nf = mlp(grouper(xyz, new_xyz, features))
            outs.append(F.max_pool2d(nf, kernel_size=[1, nf.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModuleMSGVotes(PointnetSAModuleMSG):
    """Multi-scale grouping SA that takes / returns the sampled indices (pointnet2_modules.py:500-579): `inds` (B, npoint)
    int32 selects the centres when given (furthest point sampling otherwise) and comes back as the third result, so a
    caller can look up per-centre targets (votes).  Parameters / state_dict as PointnetSAModuleMSG."""

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        """-> (new_xyz (B,npoint,3), new_features (B, sum_k mlps[k][-1], npoint), inds (B,npoint))."""
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous() \
            if self.npoint is not None else None
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            nf = mlp(grouper(xyz, new_xyz, features))                              # (B, mlp[-1], npoint, nsample)
            outs.append(F.max_pool2d(nf, kernel_size=[1, nf.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1), inds


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation (pointnet2_modules.py:644-722): for every scale k, the features of the N1 source
    points are ball-grouped around the N2 target points (QueryAndGroup(radius_k, nsample_k)), run through mlps[k], max-pooled
    over the group, concatenated with the targets' own features and passed through the ONE shared `post_mlp`; the
    per-scale results are concatenated.  -> (B, n_scales * post_mlp[-1], N2)."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int], post_mlp: List[int],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.post_mlp = pt_utils.SharedMLP(list(post_mlp), bn=bn)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                               sample_uniformly=sample_uniformly))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn))

    def forward(self, xyz2: torch.Tensor, xyz1: torch.Tensor, features2: torch.Tensor, features1: torch.Tensor):
        """xyz2 (B,N2,3) targets, xyz1 (B,N1,3) sources, features2 (B,C2,N2) or None, features1 (B,C1,N1)."""
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            nf = mlp(grouper(xyz1, xyz2, features1))                               # (B, mlp[-1], N2, nsample)
            nf = F.max_pool2d(nf, kernel_size=[1, nf.size(3)]).squeeze(-1)         # (B, mlp[-1], N2)
            if features2 is not None:
                nf = torch.cat([nf, features2], dim=1)
            outs.append(self.post_mlp(nf.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """three_nn -> inverse-distance weights -> three_interpolate -> concat skip -> SharedMLP
    (pointnet2_modules.py:582-642); weights are (1/(d+1e-8)) normalised over the 3 neighbours."""

    def __init__(self, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)
        self.fused_front_end = True     # False: the reference's op-by-op chain (kept as the parity baseline)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None and self.fused_front_end and known.shape[1] > 0:
            # three_nn -> weights -> interpolate -> concat as one op (same values; pointnet2_utils.FPInterpolateConcat)
            new_features = pointnet2_utils.fp_interpolate_concat(unknown, known, unknow_feats, known_feats)
            return pt_utils.shared_mlp_nd(self.mlp.children(), new_features)
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        new_features = torch.cat([interpolated, unknow_feats], dim=1) if unknow_feats is not None else interpolated
        return pt_utils.shared_mlp_nd(self.mlp.children(), new_features)
