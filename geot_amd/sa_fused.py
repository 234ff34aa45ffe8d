"""Host side of the fused SetAbstraction kernel (geot_amd/csrc/sa_mlp.hip): folds an
eval-mode SharedMLP (pointnet2/pytorch_utils.py:8-33: conv1x1 -> BatchNorm -> ReLU) into
the padded parameter block the kernel expects and launches it."""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .ext._common import f32, i32, same_device, need, call, ptr

_MAX_LAYERS = 4
_CACHE_ATTR = "_geot_sa_params"


def _pad_cols(c):
    return 32 if c <= 32 else 64 if c <= 64 else 128 if c <= 128 else 256


def _stages(mlp):
    """[(conv, bn or None, relu: bool)] or None if the stack is not a plain post-act SharedMLP."""
    out = []
    for stage in mlp.children():
        mods = list(stage.children())
        if not mods or not isinstance(mods[0], nn.Conv2d):
            return None
        conv, bn, relu = mods[0], None, False
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1:
            return None
        for m in mods[1:]:
            if isinstance(m, nn.Sequential) and len(m) == 1 and isinstance(m[0], nn.BatchNorm2d):
                bn = m[0]
            elif isinstance(m, nn.BatchNorm2d):
                bn = m
            elif isinstance(m, nn.ReLU):
                relu = True
            else:
                return None
        out.append((conv, bn, relu))
    return out


def fused_sa_available(mlp, nsample=32):
    """True when the fused kernel supports this stack AND this neighbourhood size (geot_sa_group_mlp_max takes
    nsample 8, 16 or a multiple of 32: a 32-row MFMA tile holds whole groups or a group holds whole tiles); callers
    compose grouper + SharedMLP + max otherwise."""
    if nsample not in (8, 16) and (nsample < 32 or nsample % 32):
        return False
    st = _stages(mlp)
    if not st or len(st) > _MAX_LAYERS:
        return False
    widths = [c.out_channels for c, _, _ in st]
    c_feat = st[0][0].in_channels - 3
    if c_feat < 0 or max(widths) > 256:
        return False
    arr = (ctypes.c_int * len(widths))(*widths)
    floats = _lib.load().geot_sa_param_floats(c_feat, len(widths), arr)
    if floats < 0:
        return False
    maxw = max([(3 + c_feat + 1) & ~1] + [_pad_cols(w) for w in widths[:-1]])   # the last layer is pooled from registers
    lds = 4 * (floats + 4 * (32 * (maxw + 1) + 4 * _pad_cols(widths[-1])))   # the smallest launch: 4 waves, 4 groups per tile
    return lds <= 160 * 1024


@torch.no_grad()
def pack_params(mlp):
    """-> (params f32 device tensor, widths list, relu_mask, c_feat); BN folded with running stats."""
    st = _stages(mlp)
    need(st is not None, "SharedMLP layout not supported by the fused SA kernel")
    key = tuple((p._version, p.data_ptr()) for p in mlp.parameters()) + \
        tuple((b._version, b.data_ptr()) for b in mlp.buffers())
    cached = getattr(mlp, _CACHE_ATTR, None)
    if cached is not None and cached[0] == key:
        return cached[1]
    dev = st[0][0].weight.device
    c_feat = st[0][0].in_channels - 3
    kp = (3 + c_feat + 1) & ~1
    blocks, widths, relu_mask = [], [], 0
    for l, (conv, bn, relu) in enumerate(st):
        w = conv.weight.detach().float().reshape(conv.out_channels, conv.in_channels)
        b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=dev)
        if bn is not None:
            scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
            w = w * scale[:, None]
            b = (b - bn.running_mean.float()) * scale + bn.bias.detach().float()
        cp = _pad_cols(conv.out_channels)
        wt = torch.zeros((kp, cp), dtype=torch.float32, device=dev)
        wt[:conv.in_channels, :conv.out_channels] = w.t()
        bb = torch.zeros(cp, dtype=torch.float32, device=dev)
        bb[:conv.out_channels] = b
        blocks += [wt.reshape(-1), bb]
        widths.append(conv.out_channels)
        relu_mask |= int(relu) << l
        kp = cp
    res = (torch.cat(blocks).contiguous(), widths, relu_mask, c_feat)
    setattr(mlp, _CACHE_ATTR, (key, res))
    return res


def fused_group_mlp_max(xyz, new_xyz, features, idx, mlp, xyz_scale=1.0):
    """xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N), idx (B,npoint,nsample) i32
    -> (B, C_out, npoint) = max_s MLP([ (xyz[idx]-new_xyz)*scale ; features[idx] ])."""
    params, widths, relu_mask, c_feat = pack_params(mlp)
    f32(xyz, "xyz", 3); f32(new_xyz, "new_xyz", 3); i32(idx, "idx", 3)
    dev = same_device(xyz, new_xyz, idx, params)
    b, n, _ = xyz.shape
    npoint, nsample = idx.shape[1], idx.shape[2]
    if c_feat:
        f32(features, "features", 3)
        need(tuple(features.shape) == (b, c_feat, n), "features must be (B, %d, N)" % c_feat)
    need(tuple(new_xyz.shape) == (b, npoint, 3), "new_xyz shape mismatch")
    out = torch.empty((b, widths[-1], npoint), dtype=torch.float32, device=dev)
    warr = (ctypes.c_int * len(widths))(*widths)
    call("geot_sa_group_mlp_max", dev, b, n, npoint, nsample, c_feat, ptr(xyz), ptr(new_xyz),
         ptr(features) if c_feat else None, ptr(idx), float(xyz_scale), len(widths), warr, relu_mask,
         ptr(params), ptr(out))
    return out
