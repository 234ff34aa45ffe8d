"""BatchNorm (+ ReLU) on channels-first (B, C, L) tensors over the streaming kernels of csrc/bnrelu.hip, and the
PointnetFPModule front end whose output arrives with its BatchNorm sums.

``bn_act(bn, x, relu)`` computes exactly what ``relu(bn(x))`` computes for an ``nn.BatchNorm1d/2d`` or
``nn.SyncBatchNorm`` module ``bn`` -- batch statistics, running-statistics update (momentum or cumulative average,
unbiased variance), ``num_batches_tracked``, eval mode on the running statistics, SyncBatchNorm's all-reduce of the
sums in both directions -- in 2 passes forward and 2 backward instead of 5 and 8 (csrc/bnrelu.hip).  The module is
only a parameter / buffer container here; tensors the kernels do not cover (CPU, other dtypes) take ``bn(x)``.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from .ext._common import call, ptr
from .pointnet2 import pointnet2_utils as pt_utils


def _sync_group(bn):
    if not isinstance(bn, nn.SyncBatchNorm) or not bn.training:
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group if dist.get_world_size(group) > 1 else None


class _BnActFn(Function):
    """out = act(x * scale + shift) with the full BatchNorm backward (the statistics' dependence on x included when
    `count` > 0: training mode; `count` = number of elements per channel over all ranks)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd, relu, count, group):
        b, c, l = x.shape
        scale = (gamma * rstd).contiguous()
        shift = (beta - mean * scale).contiguous()
        out = torch.empty_like(x)
        call("geot_bn_apply", x.device, b, c, l, int(relu), ptr(x), ptr(scale), ptr(shift), ptr(out))
        ctx.save_for_backward(x, gamma, scale, shift, mean, rstd)
        ctx.cfg = (bool(relu), count, group)      # count: python float, or a 0-dim device tensor under SyncBatchNorm
        return out

    @staticmethod
    def backward(ctx, dz):
        x, gamma, scale, shift, mean, rstd = ctx.saved_tensors
        relu, count, group = ctx.cfg
        b, c, l = x.shape
        dz = dz.contiguous()
        slices = int(_lib.load().geot_bn_slices(b, c, l))
        partial = torch.empty((b, c, slices, 2), dtype=torch.float32, device=x.device)
        call("geot_bn_bwd_reduce", x.device, b, c, l, int(relu), ptr(x), ptr(dz), ptr(scale), ptr(shift), ptr(mean),
             ptr(rstd), ptr(partial))
        sums = partial.sum(dim=(0, 2), dtype=torch.float64)                  # (C, 2): sum g, sum g * xhat (this rank)
        g_beta, g_gamma = sums[:, 0].float(), sums[:, 1].float()
        if torch.is_tensor(count) or count > 0:                              # batch statistics: they depend on x
            if group is not None:
                import torch.distributed as dist
                sums = sums.clone()
                dist.all_reduce(sums, group=group)                           # SyncBatchNorm: the means are over all ranks
            c1, c2 = (sums[:, 0] / count).float(), (sums[:, 1] / count).float()
        else:                                                                # running statistics: constants
            c1 = c2 = torch.zeros(c, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        call("geot_bn_bwd_apply", x.device, b, c, l, int(relu), ptr(x), ptr(dz), ptr(scale), ptr(shift), ptr(mean),
             ptr(rstd), ptr(scale), ptr(c1.contiguous()), ptr(c2.contiguous()), ptr(dx))
        return dx, g_gamma, g_beta, None, None, None, None, None


def _covered(bn, x):
    return (isinstance(bn, nn.modules.batchnorm._BatchNorm) and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 3 and x.shape[0] <= 65535 and x.shape[1] <= 65535 and x.numel() > 0)


def bn_act(bn, x, relu=True, partial=None):
    """act(bn(x)) for x (B, C, L) float32 on the GPU; `partial` (B, C, S, 2): per-slice (sum x, sum x^2) when the
    producer of x already formed them (fp_front)."""
    if not _covered(bn, x):
        from .pointnet2.pytorch_utils import batch_norm_nd
        y = batch_norm_nd(bn, x)
        return torch.relu(y) if relu else y
    x = x.contiguous()
    b, c, l = x.shape
    dev = x.device
    gamma = bn.weight if bn.weight is not None else torch.ones(c, device=dev)
    beta = bn.bias if bn.bias is not None else torch.zeros(c, device=dev)
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    if not use_batch:
        with torch.no_grad():
            mean = bn.running_mean.float()
            rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        return _BnActFn.apply(x, gamma, beta, mean, rstd, relu, 0.0, None)
    group = _sync_group(bn)
    with torch.no_grad():
        if partial is None:
            slices = int(_lib.load().geot_bn_slices(b, c, l))
            partial = torch.empty((b, c, slices, 2), dtype=torch.float32, device=dev)
            call("geot_bn_stats", dev, b, c, l, ptr(x), ptr(partial))
        sums = partial.sum(dim=(0, 2), dtype=torch.float64)                  # (C, 2)
        count = float(b * l)
        if group is not None:                  # SyncBatchNorm: sums and element counts of all ranks (counts may differ);
            import torch.distributed as dist   # the total stays on the device: no host synchronisation
            pack = torch.cat([sums.reshape(-1), torch.tensor([count], dtype=torch.float64, device=dev)])
            dist.all_reduce(pack, group=group)
            sums, count = pack[:-1].view(c, 2), pack[-1]
        mean64 = sums[:, 0] / count
        var64 = (sums[:, 1] / count - mean64 * mean64).clamp_min_(0.0)
        mean, rstd = mean64.float(), torch.rsqrt(var64 + bn.eps).float()
        if bn.training and bn.track_running_stats and bn.running_mean is not None:
            eaf = 0.0 if bn.momentum is None else bn.momentum
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
                if bn.momentum is None:
                    eaf = 1.0 / float(bn.num_batches_tracked)
            unbiased = var64 * (count / (count - 1.0)) if torch.is_tensor(count) else var64 * (count / max(count - 1.0, 1.0))
            bn.running_mean.mul_(1.0 - eaf).add_(mean64.to(bn.running_mean.dtype), alpha=eaf)
            bn.running_var.mul_(1.0 - eaf).add_(unbiased.to(bn.running_var.dtype), alpha=eaf)
    return _BnActFn.apply(x, gamma, beta, mean, rstd, relu, count, group)


class _FpFrontFn(Function):
    """y = three_interpolate(A, idx, w) + Wb @ skip, with the per-slice sums of y and y^2 (csrc/bnrelu.hip fp_front)."""

    @staticmethod
    def forward(ctx, a, idx, weight, skip, wb):
        b, c, m = a.shape
        n = idx.shape[1]
        cs = 0 if skip is None else skip.shape[1]
        lib = _lib.load()
        slices = int(lib.geot_fp_front_slices(b, c, m, n))
        y = torch.empty((b, c, n), dtype=torch.float32, device=a.device)
        partial = torch.empty((b, c, slices, 2), dtype=torch.float32, device=a.device)
        wbc = wb.contiguous() if cs else None
        call("geot_fp_front", a.device, b, c, m, n, cs, ptr(a), ptr(idx), ptr(weight), ptr(skip), ptr(wbc), ptr(y), ptr(partial))
        ctx.save_for_backward(idx, weight, skip, wbc)
        ctx.m = m
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, gy, _gp):
        idx, weight, skip, wb = ctx.saved_tensors
        gy = gy.contiguous()
        ga = pt_utils._ext.three_interpolate_grad(gy, idx, weight, ctx.m) if ctx.needs_input_grad[0] else None
        gskip = gwb = None
        if skip is not None:
            if ctx.needs_input_grad[3]:
                gskip = torch.bmm(wb.t().unsqueeze(0).expand(gy.shape[0], -1, -1), gy)
            if ctx.needs_input_grad[4]:
                gwb = torch.bmm(gy, skip.transpose(1, 2)).sum(0)
        return ga, None, None, gskip, gwb


def fp_front_eligible(a, skip):
    if not (a.is_cuda and a.dtype == torch.float32 and a.dim() == 3):
        return False
    cs = 0 if skip is None else skip.shape[1]
    return cs <= 8 and _lib.load().geot_fp_front_slices(a.shape[0], a.shape[1], a.shape[2], 1) > 0


def fp_front(a, idx, weight, skip, wb):
    """a (B,C,m) = W_a @ known_feats, idx / weight (B,n,3) from three_nn + the inverse-distance weights, skip (B,Cs,n)
    or None, wb (C,Cs) -> (y (B,C,n), partial): the first conv of a PointnetFPModule's SharedMLP, ready for bn_act."""
    return _FpFrontFn.apply(a.contiguous(), idx.contiguous(), weight.contiguous(),
                            None if skip is None else skip.contiguous().float(), wb)


class _SegmentMaxFn(Function):
    @staticmethod
    def forward(ctx, x, n):
        rows = x.numel() // n
        out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        arg = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
        call("geot_segment_max", x.device, rows, n, ptr(x), ptr(out), ptr(arg))
        ctx.save_for_backward(arg)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, dy):
        arg, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty(tuple(arg.shape) + (ctx.n,), dtype=torch.float32, device=dy.device)
        call("geot_segment_max_grad", dy.device, arg.numel(), ctx.n, ptr(dy), ptr(arg), ptr(dx))
        return dx, None


def max_last(x):
    """x.max(dim=-1)[0] for a contiguous float32 GPU tensor whose last dimension is a multiple of 4 and <= 256 (the
    group / nsample axis): one streaming kernel each way instead of torch's generic reduction + index scatter."""
    n = x.shape[-1]
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and 4 <= n <= 256 and n % 4 == 0 and x.numel() > 0):
        return x.max(dim=-1)[0]
    return _SegmentMaxFn.apply(x, n)
