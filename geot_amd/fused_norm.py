"""BatchNorm (+ ReLU) on channels-first (B, C, L) tensors over the streaming kernels of csrc/bnrelu.hip, and the
PointnetFPModule front end whose output arrives with its BatchNorm sums.

``bn_act(bn, x, relu)`` computes exactly what ``relu(bn(x))`` computes for an ``nn.BatchNorm1d/2d`` or
``nn.SyncBatchNorm`` module ``bn`` -- batch statistics, running-statistics update (momentum or cumulative average,
unbiased variance), ``num_batches_tracked``, eval mode on the running statistics, SyncBatchNorm's all-reduce of the
sums in both directions -- in 2 passes forward and 2 backward instead of 5 and 8 (csrc/bnrelu.hip).  The module is
only a parameter / buffer container here; tensors the kernels do not cover (CPU, other dtypes) take ``bn(x)``.

Further down: the small autograd ops that replace runs of torch launches around the dense layers (profiles/DESIGN_r01_r03.md 4.10) --
``max_last``, ``add_last_broadcast``, ``thin_mm``, ``linear`` (bias gradient as column sums), ``res_ln`` (residual add +
LayerNorm), ``qkv_split`` (attention head split) and ``softmax_last``.  Each falls back to the torch composition where its
kernel does not apply, so callers never branch.
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
    def forward(ctx, x, gamma, beta, mean, rstd, scale, shift, relu, count, group, pre_bias=None):
        b, c, l = x.shape
        if scale is None:
            scale = (gamma * rstd).contiguous()
            shift = (beta - mean * scale).contiguous()
        out = torch.empty_like(x)
        call("geot_bn_apply", x.device, b, c, l, int(relu), ptr(x), ptr(scale), ptr(shift), ptr(out))
        ctx.save_for_backward(x, gamma, scale, shift, mean, rstd)
        ctx.cfg = (bool(relu), count, group)      # count: python float, or a 1-element device double under SyncBatchNorm
        ctx.has_pre_bias = pre_bias is not None
        return out

    @staticmethod
    def backward(ctx, dz):
        x, gamma, scale, shift, mean, rstd = ctx.saved_tensors
        relu, count, group = ctx.cfg
        b, c, l = x.shape
        dev = x.device
        dz = dz.contiguous()
        slices = int(_lib.load().geot_bn_slices(b, c, l))
        partial = torch.empty((b, c, slices, 2), dtype=torch.float32, device=dev)
        call("geot_bn_bwd_reduce", dev, b, c, l, int(relu), ptr(x), ptr(dz), ptr(scale), ptr(shift), ptr(mean),
             ptr(rstd), ptr(partial))
        local = torch.empty((c, 2), dtype=torch.float64, device=dev)         # sum g, sum g * xhat (this rank)
        call("geot_bn_sums", dev, b, c, slices, ptr(partial), ptr(local))
        sums = local
        if group is not None:
            import torch.distributed as dist
            sums = local.clone()
            dist.all_reduce(sums, group=group)                               # SyncBatchNorm: the means are over all ranks
        coef = torch.empty((4, c), dtype=torch.float32, device=dev)          # g_gamma, g_beta, c1, c2
        on_dev = torch.is_tensor(count)                                      # batch statistics: they depend on x (count > 0)
        call("geot_bn_bwd_coef", dev, c, ptr(local), ptr(sums), 0.0 if on_dev else float(count), ptr(count) if on_dev else None,
             ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]))
        dx = torch.empty_like(x)
        call("geot_bn_bwd_apply", dev, b, c, l, int(relu), ptr(x), ptr(dz), ptr(scale), ptr(shift), ptr(mean),
             ptr(rstd), ptr(scale), ptr(coef[2]), ptr(coef[3]), ptr(dx))
        g_pre = None
        if ctx.has_pre_bias:    # d/d pre_bias = sum of dx over (b, l): exactly 0 under batch statistics, scale * sum g otherwise
            g_pre = scale * coef[1] if (not on_dev and float(count) == 0.0) else torch.zeros_like(scale)
        return dx, coef[0], coef[1], None, None, None, None, None, None, None, g_pre


def _covered(bn, x):
    return (isinstance(bn, nn.modules.batchnorm._BatchNorm) and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 3 and x.shape[0] <= 65535 and x.shape[1] <= 65535 and x.numel() > 0)


def _cl_stat_buffer(tiles, c, dev):
    """Uninitialised statistics buffer of a point-major producer: (tiles, 3, c) records + tiles counts (flat)."""
    return torch.empty(int(_lib.load().geot_cl_stat_floats(tiles, c)), dtype=torch.float32, device=dev)


def _batch_statistics(bn, x, gamma, beta, partial, pre_bias, cl=False):
    """Training-mode statistics of x (B, C, L) -- or, with cl, of the point-major x (B, L, C) -- for the module `bn`: the
    stats pass (unless `partial` came with x), the all-reduce under SyncBatchNorm, mean / rstd / scale / shift and the
    running-statistics update.
    -> (stats (4, C) = mean, rstd, scale, shift; count (python float, or a 1-element device double); group)."""
    if cl:
        b, l, c = x.shape
    else:
        b, c, l = x.shape
    dev = x.device
    group = _sync_group(bn)
    with torch.no_grad():
        sums = torch.empty((c, 2), dtype=torch.float64, device=dev)
        # statistics records: shifted sums (s1, s2, pivot, count) per slice / tile, rebuilt and added in fp64
        if cl:
            if partial is None:
                partial = _cl_stat_buffer(int(_lib.load().geot_cl_tiles(1, b * l, c)), c, dev)
                call("geot_bn_stats_cl", dev, b * l, c, ptr(x), ptr(partial))
            call("geot_bn_sums_shifted_cl", dev, partial.numel() // (3 * c + 1), c, ptr(partial), ptr(sums))
        else:
            if partial is None:
                slices = int(_lib.load().geot_bn_slices(b, c, l))
                partial = torch.empty((b, c, slices, 4), dtype=torch.float32, device=dev)
                call("geot_bn_stats", dev, b, c, l, ptr(x), ptr(partial))
            slices = partial.shape[2]
            # everything between the two passes in two launches (csrc/bnrelu.hip; ~13 torch launches per layer otherwise)
            call("geot_bn_sums_shifted", dev, b, c, slices, ptr(partial), ptr(sums))
        count = float(b * l)
        count_dev = None
        if group is not None:                  # SyncBatchNorm: sums and element counts of all ranks (counts may differ);
            import torch.distributed as dist   # the total stays on the device: no host synchronisation
            pack = torch.cat([sums.reshape(-1), torch.tensor([count], dtype=torch.float64, device=dev)])
            dist.all_reduce(pack, group=group)
            sums, count_dev = pack[:-1], pack[-1:]
            count = count_dev
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        eaf = 0.0
        if track:
            eaf = 0.0 if bn.momentum is None else bn.momentum
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
                if bn.momentum is None:
                    eaf = 1.0 / float(bn.num_batches_tracked)
        f32_running = track and bn.running_mean.dtype == torch.float32 and bn.running_var.dtype == torch.float32
        stats = torch.empty((4, c), dtype=torch.float32, device=dev)         # mean, rstd, scale, shift
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()   # named: raw pointers below
        p32 = None if pre_bias is None else pre_bias.detach().float().contiguous()
        call("geot_bn_finalize", dev, c, ptr(sums), 0.0 if count_dev is not None else count,
             ptr(count_dev) if count_dev is not None else None, float(bn.eps), float(eaf), ptr(g32), ptr(b32), ptr(p32),
             ptr(bn.running_mean) if f32_running else None, ptr(bn.running_var) if f32_running else None,
             ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]))
        if track and not f32_running:          # buffers in another precision: the O(c) update in torch
            n = count_dev[0] if count_dev is not None else count
            mean64 = sums.view(c, 2)[:, 0] / n
            var64 = (sums.view(c, 2)[:, 1] / n - mean64 * mean64).clamp_min_(0.0)
            if pre_bias is not None:
                mean64 = mean64 + pre_bias.detach().double()
            unbiased = var64 * (n / (n - 1.0)) if torch.is_tensor(n) else var64 * (n / max(n - 1.0, 1.0))
            bn.running_mean.mul_(1.0 - eaf).add_(mean64.to(bn.running_mean.dtype), alpha=eaf)
            bn.running_var.mul_(1.0 - eaf).add_(unbiased.to(bn.running_var.dtype), alpha=eaf)
    return stats, count, group


def bn_act(bn, x, relu=True, partial=None, pre_bias=None):
    """act(bn(x)) for x (B, C, L) float32 on the GPU; `partial` (B, C, S, 4): per-slice statistics records (shifted
    sums, pivot, count) when the producer of x already formed them (fp_front).  `pre_bias` (C,): act(bn(x + pre_bias[:, None])) without the add --
    the bias of the convolution in front: under batch statistics it cancels in the output (only the running mean sees
    it, and its gradient is exactly zero), under running statistics it folds into the shift."""
    if not _covered(bn, x):
        from .pointnet2.pytorch_utils import batch_norm_nd
        y = batch_norm_nd(bn, x if pre_bias is None else x + pre_bias.view(1, -1, 1))
        return torch.relu(y) if relu else y
    x = x.contiguous()
    b, c, l = x.shape
    dev = x.device
    gamma = bn.weight if bn.weight is not None else torch.ones(c, device=dev)
    beta = bn.bias if bn.bias is not None else torch.zeros(c, device=dev)
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    if not use_batch:
        with torch.no_grad():
            mean = bn.running_mean.float() if pre_bias is None else bn.running_mean.float() - pre_bias.detach().float()
            rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        return _BnActFn.apply(x, gamma, beta, mean, rstd, None, None, relu, 0.0, None, pre_bias)
    stats, count, group = _batch_statistics(bn, x, gamma, beta, partial, pre_bias)
    return _BnActFn.apply(x, gamma, beta, stats[0], stats[1], stats[2], stats[3], relu, count, group, pre_bias)


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
        partial = torch.empty((b, c, slices, 4), dtype=torch.float32, device=a.device)     # statistics records
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


# ---- the FP front end on point-major activations (csrc/channels_last.hip) ----------------------------------------------
# (B, N, C): a point's channels are one contiguous row.  Only the first stage of a PointnetFPModule runs in this layout
# -- the interpolation, its BatchNorm + ReLU and the gradients of both -- between two GEMMs that take the layout as a
# transpose flag: nothing is ever transposed in memory.
class _PointwiseToClFn(Function):
    """w (Cout, Cin), x (B, Cin, L) channels-first -> (B, L, Cout) point-major: the 1x1 convolution as x^T w^T."""

    @staticmethod
    def forward(ctx, w, x):
        ctx.save_for_backward(w, x)
        return torch.bmm(x.transpose(1, 2), w.t().unsqueeze(0).expand(x.shape[0], -1, -1))

    @staticmethod
    def backward(ctx, g):
        w, x = ctx.saved_tensors
        gw = gx = None
        if ctx.needs_input_grad[0]:
            gw = torch.bmm(g.transpose(1, 2), x.transpose(1, 2)).sum(0)
        if ctx.needs_input_grad[1]:
            gx = torch.bmm(w.t().unsqueeze(0).expand(x.shape[0], -1, -1), g.transpose(1, 2))
        return gw, gx


class _PointwiseFromClFn(Function):
    """w (Cout, Cin), z (B, L, Cin) point-major -> (B, Cout, L) channels-first; the gradient of z comes out point-major."""

    @staticmethod
    def forward(ctx, w, z):
        ctx.save_for_backward(w, z)
        return torch.bmm(w.unsqueeze(0).expand(z.shape[0], -1, -1), z.transpose(1, 2))

    @staticmethod
    def backward(ctx, g):
        w, z = ctx.saved_tensors
        gw = gz = None
        if ctx.needs_input_grad[0]:
            gw = torch.bmm(g, z).sum(0)
        if ctx.needs_input_grad[1]:
            gz = torch.bmm(g.transpose(1, 2), w.unsqueeze(0).expand(z.shape[0], -1, -1))
        return gw, gz


def pointwise_to_cl(w, x):
    return _PointwiseToClFn.apply(w, x.contiguous())


def pointwise_from_cl(w, z):
    return _PointwiseFromClFn.apply(w, z)


def fp_front_cl_eligible(a, skip):
    """a (B, C, m) channels-first (shape only): does the point-major front end cover it?  Wide layers only: below 256
    channels a row is shorter than a wave and the channels-first kernels are the better fit."""
    if not (a.is_cuda and a.dtype == torch.float32 and a.dim() == 3):
        return False
    b, c, m = a.shape
    cs = 0 if skip is None else skip.shape[1]
    return c >= 256 and cs <= 8 and _lib.load().geot_fp_front_cl_tiles(b, c, 16, cs) > 0


def local_spatial_order(pos):
    """(B, N, 3) -> int32 (B, N): the points of every cloud in Morton-cell order (per-cloud ids).  The sequence in which the
    point-major kernels take their rows: neighbouring rows of the sequence gather the same table rows."""
    from . import ntm
    b, n, _ = pos.shape
    order = ntm.spatial_order(pos.contiguous())
    if order is None:
        return None
    return (order.view(b, n) - torch.arange(b, device=pos.device, dtype=torch.int32).view(b, 1) * n).contiguous()


class ReverseIndex:
    """The (b, m)-target reverse index of idx (b, n, 3) + its inverse-distance weights for the point-major gradient
    (geot_rix_build): depends on the coordinates only, so the model builds it on its side stream with the index plan."""

    def __init__(self, idx, weight, m, order=None):
        b, n, nt = idx.shape
        self.b, self.n, self.m, self.nt, self.order = b, n, m, nt, order
        lib = _lib.load()
        self.ws_ints = int(lib.geot_rix_ws_ints(b, n, m, nt))
        self.ws = torch.empty(self.ws_ints, dtype=torch.int32, device=idx.device)
        call("geot_rix_build", idx.device, b, n, m, nt, ptr(idx), ptr(weight), ptr(order), ptr(self.ws), self.ws_ints)

    def gather(self, g_cl):
        b, n, c = g_cl.shape
        assert (b, n) == (self.b, self.n)
        out = torch.empty((b, self.m, c), dtype=torch.float32, device=g_cl.device)
        call("geot_gather_rows_csr_cl", g_cl.device, b, c, n, self.m, self.nt, ptr(g_cl), ptr(self.ws), ptr(self.order), ptr(out))
        return out


class _FpFrontClFn(Function):
    """y_cl = three_interpolate(A, idx, w) + skip^T Wb^T on point-major tensors, with the per-tile sums of y and y^2."""

    @staticmethod
    def forward(ctx, a_cl, idx, weight, skip, wb, order, rix):
        b, m, c = a_cl.shape
        n = idx.shape[1]
        cs = 0 if skip is None else skip.shape[1]
        tiles = int(_lib.load().geot_fp_front_cl_tiles(b, c, n, cs))
        y = torch.empty((b, n, c), dtype=torch.float32, device=a_cl.device)
        partial = _cl_stat_buffer(tiles, c, a_cl.device)
        wbc = wb.contiguous() if cs else None
        call("geot_fp_front_cl", a_cl.device, b, c, m, n, cs, ptr(a_cl), ptr(idx), ptr(weight), ptr(skip), ptr(wbc), ptr(order),
             ptr(y), ptr(partial))
        ctx.save_for_backward(idx, weight, skip, wbc)
        ctx.m, ctx.rix = m, rix
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, gy, _gp):
        idx, weight, skip, wb = ctx.saved_tensors
        gy = gy.contiguous()
        ga = gskip = gwb = None
        if ctx.needs_input_grad[0]:
            rix = ctx.rix if ctx.rix is not None else ReverseIndex(idx, weight, ctx.m)
            ga = rix.gather(gy)
        if skip is not None:
            if ctx.needs_input_grad[3]:
                gskip = torch.matmul(gy, wb).transpose(1, 2)                  # (B, n, cs) -> (B, cs, n)
            if ctx.needs_input_grad[4]:
                gwb = torch.bmm(skip, gy).sum(0).t()                          # (B, cs, n) x (B, n, C)
        return ga, None, None, gskip, gwb, None, None


def fp_front_cl(a_cl, idx, weight, skip, wb, order=None, rix=None):
    """a_cl (B,m,C) = known_feats^T W_a^T, idx / weight (B,n,3), skip (B,Cs,n) channels-first or None, wb (C,Cs) ->
    (y_cl (B,n,C), partial): fp_front on point-major activations.  order (B,n): the row sequence of the launch (values do
    not depend on it); rix: a ReverseIndex of (idx, weight) built ahead (else the backward builds one)."""
    return _FpFrontClFn.apply(a_cl.contiguous(), idx.contiguous(), weight.contiguous(),
                              None if skip is None else skip.contiguous().float(), wb, order, rix)


class _BnActClFn(Function):
    """_BnActFn on a point-major tensor (B, L, C)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, rstd, scale, shift, relu, count, group, pre_bias=None):
        b, l, c = x.shape
        if scale is None:
            scale = (gamma * rstd).contiguous()
            shift = (beta - mean * scale).contiguous()
        out = torch.empty_like(x)
        call("geot_bn_apply_cl", x.device, b * l, c, int(relu), ptr(x), ptr(scale), ptr(shift), ptr(out))
        ctx.save_for_backward(x, gamma, scale, shift, mean, rstd)
        ctx.cfg = (bool(relu), count, group)
        ctx.has_pre_bias = pre_bias is not None
        return out

    @staticmethod
    def backward(ctx, dz):
        x, gamma, scale, shift, mean, rstd = ctx.saved_tensors
        relu, count, group = ctx.cfg
        b, l, c = x.shape
        dev = x.device
        dz = dz.contiguous()
        partial = torch.empty((int(_lib.load().geot_cl_tiles(1, b * l, c)), 2, c), dtype=torch.float32, device=dev)
        call("geot_bn_bwd_reduce_cl", dev, b * l, c, int(relu), ptr(x), ptr(dz), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
             ptr(partial))
        local = torch.empty((c, 2), dtype=torch.float64, device=dev)
        call("geot_bn_sums_cl", dev, partial.shape[0], c, ptr(partial), ptr(local))
        sums = local
        if group is not None:
            import torch.distributed as dist
            sums = local.clone()
            dist.all_reduce(sums, group=group)
        coef = torch.empty((4, c), dtype=torch.float32, device=dev)
        on_dev = torch.is_tensor(count)
        call("geot_bn_bwd_coef", dev, c, ptr(local), ptr(sums), 0.0 if on_dev else float(count), ptr(count) if on_dev else None,
             ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]))
        dx = torch.empty_like(x)
        call("geot_bn_bwd_apply_cl", dev, b * l, c, int(relu), ptr(x), ptr(dz), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
             ptr(scale), ptr(coef[2]), ptr(coef[3]), ptr(dx))
        g_pre = None
        if ctx.has_pre_bias:
            g_pre = scale * coef[1] if (not on_dev and float(count) == 0.0) else torch.zeros_like(scale)
        return dx, coef[0], coef[1], None, None, None, None, None, None, None, g_pre


def bn_act_cl(bn, x, relu=True, partial=None, pre_bias=None):
    """bn_act for a point-major x (B, L, C) float32 on the GPU with C % 4 == 0 (geot_cl_tiles >= 0); `partial`: the
    statistics buffer fp_front_cl returned.  Same statistics, running buffers and SyncBatchNorm behaviour as bn_act."""
    b, l, c = x.shape
    if not (isinstance(bn, nn.modules.batchnorm._BatchNorm) and x.is_cuda and x.dtype == torch.float32
            and _lib.load().geot_cl_tiles(1, b * l, c) > 0):
        return bn_act(bn, x.transpose(1, 2).contiguous(), relu, None, pre_bias).transpose(1, 2).contiguous()
    x = x.contiguous()
    dev = x.device
    gamma = bn.weight if bn.weight is not None else torch.ones(c, device=dev)
    beta = bn.bias if bn.bias is not None else torch.zeros(c, device=dev)
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    if not use_batch:
        with torch.no_grad():
            mean = bn.running_mean.float() if pre_bias is None else bn.running_mean.float() - pre_bias.detach().float()
            rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        return _BnActClFn.apply(x, gamma, beta, mean, rstd, None, None, relu, 0.0, None, pre_bias)
    stats, count, group = _batch_statistics(bn, x, gamma, beta, partial, pre_bias, cl=True)
    return _BnActClFn.apply(x, gamma, beta, stats[0], stats[1], stats[2], stats[3], relu, count, group, pre_bias)


class _FpStageClFn(Function):
    """[fp_front_cl -> BatchNorm (+ ReLU)] as ONE node: z_cl = act(bn(three_interpolate(A) + skip^T Wb^T)).
    Forward: the two launches of the separate ops.  Backward: two passes over (y, dz) instead of four over (y, dz, gy) --
    geot_bn_bwd_reduce_skip_cl (BatchNorm sums + the sums of the skip-weight gradient) and geot_gather_rows_csr_bn_cl (the
    interpolation gradient with gy = scale (g - c1 - xhat c2) formed on the fly): gy is never written."""

    @staticmethod
    def forward(ctx, a_cl, idx, weight, skip, wb, gamma, beta, bn, relu, order, rix):
        b, m, c = a_cl.shape
        n = idx.shape[1]
        dev = a_cl.device
        cs = 0 if skip is None else skip.shape[1]
        tiles = int(_lib.load().geot_fp_front_cl_tiles(b, c, n, cs))
        y = torch.empty((b, n, c), dtype=torch.float32, device=dev)
        partial = _cl_stat_buffer(tiles, c, dev)
        wbc = wb.contiguous() if cs else None
        call("geot_fp_front_cl", dev, b, c, m, n, cs, ptr(a_cl), ptr(idx), ptr(weight), ptr(skip), ptr(wbc), ptr(order), ptr(y),
             ptr(partial))
        use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
        if use_batch:
            stats, count, group = _batch_statistics(bn, y, gamma, beta, partial, None, cl=True)
            mean, rstd, scale, shift = stats[0], stats[1], stats[2], stats[3]
        else:
            mean = bn.running_mean.float()
            rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
            scale = (gamma.detach() * rstd).contiguous()
            shift = (beta.detach() - mean * scale).contiguous()
            count, group = 0.0, None
        z = torch.empty_like(y)
        call("geot_bn_apply_cl", dev, b * n, c, int(relu), ptr(y), ptr(scale), ptr(shift), ptr(z))
        ctx.save_for_backward(y, idx, weight, skip, wbc, scale, shift, mean, rstd)
        ctx.cfg = (bool(relu), count, group, m, order, rix)
        # S2 of the skip-weight gradient (input data only); not under no_grad / for frozen weights
        ctx.skip_sums = rowsum_f64(skip).sum(0) if (cs and ctx.needs_input_grad[4]) else None     # (B, cs) -> (cs,), fp64
        return z

    @staticmethod
    def backward(ctx, dz):
        y, idx, weight, skip, wb, scale, shift, mean, rstd = ctx.saved_tensors
        relu, count, group, m, order, rix = ctx.cfg
        b, n, c = y.shape
        dev = y.device
        cs = 0 if skip is None else skip.shape[1]
        k = 2 + 2 * cs
        dz = dz.contiguous()
        lib = _lib.load()
        partial = torch.empty((int(lib.geot_cl_tiles(1, b * n, c)), k, c), dtype=torch.float32, device=dev)
        call("geot_bn_bwd_reduce_skip_cl", dev, b, n, c, cs, int(relu), ptr(y), ptr(dz), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
             ptr(skip), ptr(partial))
        sums_k = torch.empty((c, k), dtype=torch.float64, device=dev)
        call("geot_bn_sums_k_cl", dev, partial.shape[0], c, k, ptr(partial), ptr(sums_k))
        local = sums_k[:, :2].contiguous()                                   # sum g, sum g xhat (this rank)
        sums = local
        if group is not None:
            import torch.distributed as dist
            sums = local.clone()
            dist.all_reduce(sums, group=group)
        coef = torch.empty((4, c), dtype=torch.float32, device=dev)          # g_gamma, g_beta, c1, c2
        on_dev = torch.is_tensor(count)
        call("geot_bn_bwd_coef", dev, c, ptr(local), ptr(sums), 0.0 if on_dev else float(count), ptr(count) if on_dev else None,
             ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]))
        ga = gskip = gwb = None
        if ctx.needs_input_grad[0]:
            if rix is None:
                rix = ReverseIndex(idx, weight, m)
            ga = torch.empty((b, m, c), dtype=torch.float32, device=dev)
            call("geot_gather_rows_csr_bn_cl", dev, b, c, n, m, rix.nt, int(relu), ptr(y), ptr(dz), ptr(scale), ptr(shift), ptr(mean),
                 ptr(rstd), ptr(coef[2]), ptr(coef[3]), ptr(rix.ws), ptr(rix.order), ptr(ga))
        if cs and ctx.needs_input_grad[4]:
            # grad_wb[c, j] = sum_e gy[e, c] skip_j[e] = scale_c (sum g skip_j - c1_c sum skip_j - c2_c sum xhat skip_j)
            gwb = torch.empty((c, cs), dtype=torch.float32, device=dev)
            call("geot_fp_skip_wgrad_cl", dev, c, cs, ptr(sums_k), ptr(scale), ptr(coef[2]), ptr(coef[3]), ptr(ctx.skip_sums), ptr(gwb))
        if cs and ctx.needs_input_grad[3]:                                   # (never in the model: the skip tensor is input data)
            gy = torch.empty_like(y)
            call("geot_bn_bwd_apply_cl", dev, b * n, c, int(relu), ptr(y), ptr(dz), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                 ptr(scale), ptr(coef[2]), ptr(coef[3]), ptr(gy))
            gskip = torch.matmul(gy, wb).transpose(1, 2)
        return ga, None, None, gskip, gwb, coef[0], coef[1], None, None, None, None


def fp_stage_cl(bn, a_cl, idx, weight, skip, wb, relu=True, order=None, rix=None):
    """act(bn(fp_front_cl(...))) for a BatchNorm module `bn` (training or eval mode, SyncBatchNorm included) as one
    autograd node whose backward never materialises the gradient of the BatchNorm's input.  -> z_cl (B, n, C)."""
    c = a_cl.shape[2]
    dev = a_cl.device
    gamma = bn.weight if bn.weight is not None else torch.ones(c, device=dev)
    beta = bn.bias if bn.bias is not None else torch.zeros(c, device=dev)
    return _FpStageClFn.apply(a_cl.contiguous(), idx.contiguous(), weight.contiguous(),
                              None if skip is None else skip.contiguous().float(), wb, gamma, beta, bn, relu, order, rix)


def rowsum_f64(x):
    """x (..., n) contiguous float32 on the GPU -> fp64 sums over the last axis, shape x.shape[:-1]: one workgroup per row, fixed
    order (geot_rowsum_f64).  For long rows and few of them, where torch's own reduction zeroes a semaphore buffer with
    hipMemsetAsync (a memset node once captured: graph_step.py)."""
    x = x.contiguous()
    out = torch.empty(x.shape[:-1], dtype=torch.float64, device=x.device)
    call("geot_rowsum_f64", x.device, out.numel(), x.shape[-1], ptr(x), ptr(out))
    return out


class _AddChannelBiasFn(Function):
    """y (B, C, L) or (C, L) + bias (C) as one launch; the bias gradient = sums of the incoming gradient over (B, L) through
    rowsum_f64."""

    @staticmethod
    def forward(ctx, y, bias):
        return y + bias.view(-1, 1)

    @staticmethod
    def backward(ctx, g):
        gb = None
        if ctx.needs_input_grad[1]:
            gb = rowsum_f64(g)                          # fp64 row sums; (B, C): then the B rows, a small reduction
            gb = (gb.sum(0) if g.dim() == 3 else gb).float()
        return g, gb


def add_channel_bias(y, bias):
    """y (B, C, L) + bias.view(1, C, 1), or y (C, L) + bias.view(C, 1), for float32 GPU tensors; anything else: the torch
    expression (whose bias gradient is an aten::sum -- for some shapes with a memset in front: see rowsum_f64)."""
    if not (y.is_cuda and y.dtype == torch.float32 and bias.dtype == torch.float32 and y.dim() in (2, 3) and y.numel() > 0):
        return y + bias.view(-1, 1)
    return _AddChannelBiasFn.apply(y, bias)


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


class _AddLastBroadcastFn(Function):
    @staticmethod
    def forward(ctx, a, p):
        ctx.n = a.shape[-1]
        return a + p.unsqueeze(-1)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gp = torch.empty(g.shape[:-1], dtype=torch.float32, device=g.device)
        call("geot_segment_sum", g.device, gp.numel(), ctx.n, ptr(g), ptr(gp))
        return g, gp


def add_last_broadcast(a, p):
    """a (..., n) + p (...)[..., None] whose gradient for p is one streaming row-sum kernel (torch's reduction over a
    short last dimension runs at 0.95 TB/s: 283 us for the Encoder's (512, 4096, 32) tensor at 8 clouds, 55 us here)."""
    n = a.shape[-1]
    if not (a.is_cuda and a.dtype == torch.float32 and p.dtype == torch.float32 and 4 <= n <= 256 and n % 4 == 0
            and tuple(p.shape) == tuple(a.shape[:-1]) and a.numel() > 0):
        return a + p.unsqueeze(-1)
    return _AddLastBroadcastFn.apply(a, p)


class _ThinMmFn(Function):
    """w (C, J) @ x (J, L) for J <= 8: the forward is a fine library GEMM, its weight gradient (C x J x L with L ~ 1e5) is
    not -- one streaming pass of geot_rowdot_small instead."""

    @staticmethod
    def forward(ctx, w, x):
        ctx.save_for_backward(w, x)
        return torch.mm(w, x)

    @staticmethod
    def backward(ctx, g):
        w, x = ctx.saved_tensors
        gw = gx = None
        if ctx.needs_input_grad[0]:
            g = g.contiguous()
            c, l = g.shape
            j = x.shape[0]
            s = int(_lib.load().geot_rowdot_small_slices(c, l))
            partial = torch.empty((c, s, j), dtype=torch.float32, device=g.device)
            call("geot_rowdot_small", g.device, c, l, j, ptr(g), ptr(x), ptr(partial))
            gw = partial.sum(1)
        if ctx.needs_input_grad[1]:
            gx = torch.mm(w.t(), g)
        return gw, gx


def thin_mm(w, x):
    """torch.mm(w, x) for w (C, J), x (J, L) contiguous float32 on the GPU with J <= 8 and C <= 65535."""
    if not (w.is_cuda and w.dtype == torch.float32 and x.dtype == torch.float32 and w.dim() == 2 and x.dim() == 2
            and 1 <= x.shape[0] <= 8 and w.shape[0] <= 65535 and x.numel() > 0):
        return torch.mm(w, x)
    return _ThinMmFn.apply(w, x.contiguous())


class _LinearFn(Function):
    """F.linear(x, w, b) with the same three GEMMs as torch's own backward; the bias gradient (column sums of the
    output gradient) from geot_colsum: 2 launches of ~4 us instead of a 5 us fill + a 19 us generic reduction."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g2, w).view(x.shape)
        if ctx.needs_input_grad[1]:
            gw = torch.mm(g2.t(), x2)
        if ctx.needs_input_grad[2]:
            g2 = g2.contiguous()
            rows, cols = g2.shape
            gb = torch.empty(cols, dtype=torch.float32, device=g.device)
            ws = torch.empty(int(_lib.load().geot_colsum_ws_floats(rows, cols)), dtype=torch.float32, device=g.device)
            call("geot_colsum", g.device, rows, cols, ptr(g2), ptr(gb), ptr(ws))
        return gx, gw, gb


def linear(mod, x):
    """mod(x) for an nn.Linear with bias on a float32 GPU tensor (anything else: the module itself)."""
    if mod.bias is None or not (x.is_cuda and x.dtype == torch.float32 and mod.weight.dtype == torch.float32 and x.numel() > 0):
        return mod(x)
    return _LinearFn.apply(x, mod.weight, mod.bias)


def _res_ln_forward(x, y, s, extra, gamma, beta, eps, want_t):
    c = x.shape[-1]
    rows = x.numel() // c
    rps = max(rows // x.shape[0], 1)
    t = torch.empty_like(x) if want_t else None
    z = torch.empty_like(x)
    stats = torch.empty((2, rows), dtype=torch.float32, device=x.device)
    call("geot_res_ln", x.device, rows, c, rps, float(eps), ptr(x), ptr(y), ptr(s), ptr(extra), ptr(gamma), ptr(beta),
         ptr(t), ptr(z), ptr(stats[0]), ptr(stats[1]))
    return t, z, stats, (rows, c, rps)


def _res_ln_backward(t, stats, gamma, s, dims, gt, gz, want_gy):
    rows, c, rps = dims
    dev = t.device
    gt = None if gt is None else gt.contiguous()
    gz = None if gz is None else gz.contiguous()
    g = torch.empty_like(t)
    gy = torch.empty_like(t) if want_gy else None
    dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
    ws = torch.empty(int(_lib.load().geot_res_ln_ws_floats(rows, c)), dtype=torch.float32, device=dev)
    call("geot_res_ln_grad", dev, rows, c, rps, ptr(gz), ptr(gt), ptr(t), ptr(stats[0]), ptr(stats[1]), ptr(gamma), ptr(s),
         ptr(g), ptr(gy), ptr(dgb[0]), ptr(dgb[1]), ptr(ws))
    return g, gy, dgb[0], dgb[1]


class _ResLnFn(Function):
    """t = x + s * y + extra, z = LayerNorm(t): csrc/layernorm.hip, one launch forward, one + a finish backward."""

    @staticmethod
    def forward(ctx, x, y, s, extra, gamma, beta, eps):
        t, z, stats, dims = _res_ln_forward(x, y, s, extra, gamma, beta, eps, True)
        ctx.save_for_backward(t, stats, gamma, s)
        ctx.cfg = (dims, y is not None, extra is not None)
        ctx.set_materialize_grads(False)
        return t, z

    @staticmethod
    def backward(ctx, gt, gz):
        t, stats, gamma, s = ctx.saved_tensors
        dims, has_y, has_extra = ctx.cfg
        g, gy, dg, db = _res_ln_backward(t, stats, gamma, s, dims, gt, gz, has_y and s is not None)
        return g, (gy if gy is not None else g) if has_y else None, None, g if has_extra else None, dg, db, None


class _LnFn(Function):
    """z = LayerNorm(x) through the same kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _, z, stats, dims = _res_ln_forward(x, None, None, None, gamma, beta, eps, False)
        ctx.save_for_backward(x, stats, gamma)
        ctx.dims = dims
        return z

    @staticmethod
    def backward(ctx, gz):
        x, stats, gamma = ctx.saved_tensors
        g, _, dg, db = _res_ln_backward(x, stats, gamma, None, ctx.dims, None, gz, False)
        return g, dg, db, None


def res_ln_eligible(x, norm):
    return (isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 3 and len(norm.normalized_shape) == 1
            and norm.normalized_shape[0] == x.shape[-1] and x.numel() > 0
            and bool(_lib.load().geot_res_ln_supported(int(x.shape[-1]))))


def res_ln(x, y, s, extra, norm):
    """(t, norm(t)) with t = x + s * y + extra; y (branch output), s ((B,1,1) drop-path factors) and extra (position
    embedding) may be None.  Falls back to the torch composition where the kernels do not apply."""
    if not res_ln_eligible(x, norm):
        t = x
        if y is not None:
            t = t + y if s is None else torch.addcmul(t, y, s)
        if extra is not None:
            t = t + extra
        return t, norm(t)
    if y is None and extra is None:
        x = x.contiguous()
        return x, _LnFn.apply(x, norm.weight, norm.bias, norm.eps)
    cont = lambda v: None if v is None else v.contiguous()                      # noqa: E731
    return _ResLnFn.apply(x.contiguous(), cont(y), cont(s), cont(extra), norm.weight, norm.bias, norm.eps)


class _QkvSplitFn(Function):
    @staticmethod
    def forward(ctx, qkv, heads, scale):
        b, n, c3 = qkv.shape
        d = c3 // (3 * heads)
        out = torch.empty((3, b * heads, n, d), dtype=torch.float32, device=qkv.device)
        call("geot_qkv_split", qkv.device, b, n, heads, d, float(scale), ptr(qkv), ptr(out))
        ctx.cfg = (b, n, heads, d, float(scale))
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, gq, gk, gv):
        b, n, heads, d, scale = ctx.cfg
        gq, gk, gv = (None if g is None else g.contiguous() for g in (gq, gk, gv))
        dev = next(g for g in (gq, gk, gv) if g is not None).device
        grad = torch.empty((b, n, 3 * heads * d), dtype=torch.float32, device=dev)
        call("geot_qkv_split_grad", dev, b, n, heads, d, scale, ptr(gq), ptr(gk), ptr(gv), ptr(grad))
        return grad, None, None


def qkv_split(qkv, heads, scale):
    """qkv (B, N, 3*H*d) -> (q * scale, k, v), each (B*H, N, d) contiguous (csrc/layernorm.hip); None where it does not apply."""
    if not (qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 3 and qkv.is_contiguous() and qkv.numel() > 0
            and qkv.shape[2] % (3 * heads) == 0 and (qkv.shape[2] // (3 * heads)) % 4 == 0):
        return None
    return _QkvSplitFn.apply(qkv, heads, scale)


class _SoftmaxLastFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.softmax(x, dim=-1)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        g = g.contiguous()
        gi = torch.empty_like(y)
        call("geot_softmax_grad", y.device, y.numel() // y.shape[-1], y.shape[-1], ptr(g), ptr(y), ptr(gi))
        return gi


def softmax_last(x):
    """x.softmax(dim=-1) whose gradient is one pass (csrc/layernorm.hip) for float32 GPU rows of 64 .. 1024 (power of two)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] in (64, 128, 256, 512, 1024) and x.numel() > 0):
        return x.softmax(dim=-1)
    return _SoftmaxLastFn.apply(x)


class _BnPoolFn(Function):
    """max over the last n of relu(bn(y)) for y (B, C, G*n) under batch statistics, the normalised tensor never built."""

    @staticmethod
    def forward(ctx, y, gamma, beta, stats, n, count):
        b, c, l = y.shape
        g = l // n
        out = torch.empty((b, c, g), dtype=torch.float32, device=y.device)
        sel = torch.empty((b, c, g), dtype=torch.float32, device=y.device)
        arg = torch.empty((b, c, g), dtype=torch.uint8, device=y.device)
        call("geot_bn_pool", y.device, b, c, g, n, 1, ptr(y), ptr(stats[2]), ptr(stats[3]), ptr(out), ptr(sel), ptr(arg))
        ctx.save_for_backward(y, stats, sel, arg, out)
        ctx.cfg = (n, count)
        return out

    @staticmethod
    def backward(ctx, gp):
        y, stats, sel, arg, out = ctx.saved_tensors
        n, count = ctx.cfg
        b, c, l = y.shape
        g = l // n
        mean, rstd, scale = stats[0], stats[1], stats[2]
        gm = (gp * (out > 0)).contiguous()                                  # gradient of the normalised tensor at arg
        xh = (sel - mean.view(1, c, 1)) * rstd.view(1, c, 1)
        s1 = gm.sum(dim=(0, 2), dtype=torch.float64)
        s2 = (gm * xh).sum(dim=(0, 2), dtype=torch.float64)
        c1, c2 = (s1 / count).float().contiguous(), (s2 / count).float().contiguous()
        dx = torch.empty_like(y)
        call("geot_bn_pool_grad", y.device, b, c, g, n, ptr(y), ptr(gm), ptr(arg), ptr(mean), ptr(rstd), ptr(scale), ptr(c1), ptr(c2),
             ptr(dx))
        return dx, s2.float(), s1.float(), None, None, None


def bn_relu_max(bn, y, n):
    """relu(bn(y)) max-pooled over the last n of y (B, C, G*n): (B, C, G).  Training mode of a plain BatchNorm goes through
    the monotone form (csrc/bnrelu.hip bn_pool: one pass forward besides the statistics, one backward); everything else
    composes bn_act + max_last."""
    b, c, l = y.shape
    fused = (_covered(bn, y) and bn.training and _sync_group(bn) is None and not isinstance(bn, nn.SyncBatchNorm)
             and bn.weight is not None and bn.bias is not None and 4 <= n <= 256 and n % 4 == 0 and l % n == 0)
    if not fused:
        return max_last(bn_act(bn, y, relu=True).view(b, c, l // n, n))
    y = y.contiguous()
    stats, count, _ = _batch_statistics(bn, y, bn.weight, bn.bias, None, None)
    return _BnPoolFn.apply(y, bn.weight, bn.bias, stats, n, count)
