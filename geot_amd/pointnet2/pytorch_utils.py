"""SharedMLP and friends -- structural mirror of pointnet2/pytorch_utils.py:8-117.

Layer order per stage is conv(1x1, bias = not bn) -> BatchNorm -> ReLU and the
state_dict keys are the reference's (``layer{i}.conv.weight``, ``layer{i}.bn.bn.*``)
so checkpoints load unchanged; pinned by tests/golden/shared_mlp_ref.npz, which was
produced by the reference class itself.
"""
import torch.nn as nn


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
        super().__init__(cin, cout, nn.Conv1d, nn.BatchNorm1d, bn, activation, preact)


class Conv2d(_ConvStage):
    def __init__(self, cin, cout, *, bn=False, activation=nn.ReLU(inplace=True), preact=False):
        super().__init__(cin, cout, nn.Conv2d, nn.BatchNorm2d, bn, activation, preact)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d stages over a (B, C, npoint, nsample) tensor."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = (not first) or (not preact) or (i != 0)
            self.add_module("layer%d" % i, Conv2d(args[i], args[i + 1], bn=plain and bn,
                                                  activation=activation if plain else None, preact=preact))
