"""SharedMLP and friends -- structural mirror of pointnet2/pytorch_utils.py:8-117.

Layer order per stage is conv(1x1, bias = not bn) -> BatchNorm -> ReLU and the
state_dict keys are the reference's (``layer{i}.conv.weight``, ``layer{i}.bn.bn.*``)
so checkpoints load unchanged; pinned by tests/golden/shared_mlp_ref.npz, which was
produced by the reference class itself.
"""
import torch
import torch.nn as nn


def pointwise(w, x):
    """w (Cout,Cin) @ x (B,Cin,L) -> (B,Cout,L) as ONE strided-batched GEMM whose output is contiguous.
    (torch.matmul(2-D, 3-D) folds the batch into the rows instead: it copies x transposed and returns a transposed
    view, i.e. two strided copies of the largest tensors of the model per layer -- 15 ms of a 73 ms step, measured.)"""
    return torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)


def conv1x1(conv, x):
    """A kernel-size-1 convolution as ONE batched GEMM ``W (Cout,Cin) @ x (B,Cin,L)`` -> rocBLAS / hipBLASLt, for
    the forward, the data gradient and the weight gradient alike.  MIOpen's convolution path picks Winograd /
    per-sample solvers for these shapes on gfx950 (10 ms per 1x1 layer at 8 x 24 000 points, measured) and spends
    seconds in its find step; the arithmetic is the same dot products."""
    shp = x.shape
    y = pointwise(conv.weight.view(conv.out_channels, -1), x.reshape(shp[0], shp[1], -1))
    if conv.bias is not None:
        from ..fused_norm import add_channel_bias        # (its bias gradient: one fixed-order kernel instead of aten::sum)
        y = add_channel_bias(y, conv.bias)
    return y.view(shp[0], conv.out_channels, *shp[2:])


def batch_norm_nd(bn, x):
    """``bn(x)`` for a BatchNorm1d/2d module on an (B, C, *) tensor of any rank (same statistics, same running
    buffers as nn.modules.batchnorm._BatchNorm.forward): the (B,C,N,1) view BatchNorm2d insists on makes torch's
    layout heuristics take strided element-wise paths behind it.  SyncBatchNorm takes any rank itself."""
    if isinstance(bn, nn.SyncBatchNorm) or not isinstance(bn, nn.modules.batchnorm._BatchNorm):
        return bn(x)
    eaf = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            eaf = 1.0 / float(bn.num_batches_tracked)
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    keep = not bn.training or bn.track_running_stats
    return torch.nn.functional.batch_norm(x, bn.running_mean if keep else None, bn.running_var if keep else None,
                                          bn.weight, bn.bias, use_batch, eaf, bn.eps)


def shared_mlp_nd(layers, x, first_conv_done=False):
    """Run SharedMLP stages (conv -> BatchNorm -> activation, or pre-activation order) on (B, C, L) tensors; a
    BatchNorm followed by ReLU is one fused op (geot_amd/fused_norm.py: 2 + 2 passes instead of 5 + 8).
    first_conv_done: x already is the output of the first stage's convolution (the caller evaluated it in its own way)."""
    from ..fused_norm import bn_act
    for k, stage in enumerate(layers):
        mods = list(stage.named_children())
        i = 0
        while i < len(mods):
            name, mod = mods[i]
            if name == "conv":
                if k == 0 and first_conv_done:
                    i += 1
                    continue
                x = mod(x) if isinstance(mod, (PointwiseConv1d, PointwiseConv2d)) else conv1x1(mod, x)
            elif name == "bn":
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1][1], nn.ReLU)
                x = bn_act(mod.bn, x, relu=fuse)
                i += 1 if fuse else 0
            else:
                x = mod(x)
            i += 1
    return x


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d(kernel_size=1) with the same parameters / state_dict, evaluated by conv1x1()."""

    def forward(self, x):
        return conv1x1(self, x)


class PointwiseConv2d(nn.Conv2d):
    def forward(self, x):
        return conv1x1(self, x)


class _BN(nn.Sequential):
    def __init__(self, channels, cls):
        super().__init__()
        self.add_module("bn", cls(channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class _ConvStage(nn.Sequential):
    def __init__(self, cin, cout, conv_cls, bn_cls, bn, activation, preact=False):
        super().__init__()
        conv = conv_cls(cin, cout, kernel_size=1, stride=1, padding=0, bias=not bn)
        nn.init.kaiming_normal_(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        if preact:
            if bn:
                self.add_module("bn", _BN(cin, bn_cls))
            if activation is not None:
                self.add_module("activation", activation)
        self.add_module("conv", conv)
        if not preact:
            if bn:
                self.add_module("bn", _BN(cout, bn_cls))
            if activation is not None:
                self.add_module("activation", activation)


class Conv1d(_ConvStage):
    def __init__(self, cin, cout, *, bn=False, activation=nn.ReLU(inplace=True), preact=False):
        super().__init__(cin, cout, PointwiseConv1d, nn.BatchNorm1d, bn, activation, preact)


class Conv2d(_ConvStage):
    def __init__(self, cin, cout, *, bn=False, activation=nn.ReLU(inplace=True), preact=False):
        super().__init__(cin, cout, PointwiseConv2d, nn.BatchNorm2d, bn, activation, preact)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d stages over a (B, C, npoint, nsample) tensor."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = (not first) or (not preact) or (i != 0)
            self.add_module("layer%d" % i, Conv2d(args[i], args[i + 1], bn=plain and bn,
                                                  activation=activation if plain else None, preact=preact))
