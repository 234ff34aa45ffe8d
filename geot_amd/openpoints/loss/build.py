"""The two criteria the configured FixMatch+NTM step uses (cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml
criterion_args / criterion_u_args): Poly1FocalLoss (openpoints/loss/build.py:183-258) on the labelled clouds and
Poly1FocalLoss_U_corr (:799-892) on the NTM-corrected strong-view logits.  Element-wise torch on (B, 17, N)
tensors -- callers of the hot path, mirrored only so that the step of BASELINE configs[4] closes; same
constructor arguments and forward signatures, same arithmetic."""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function


class _Poly1FocalFn(Function):
    """csrc/loss.hip: the loss from integer labels in two launches, its gradient in one."""

    @staticmethod
    def forward(ctx, logits, labels, keep, alpha, gamma, epsilon):
        from ... import _lib
        from ...ext._common import call, ptr
        b, c, n = logits.shape
        ws = torch.empty(int(_lib.load().geot_poly1_focal_ws_doubles(b, c, n)), dtype=torch.float64, device=logits.device)
        out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
        call("geot_poly1_focal", logits.device, b, c, n, float(alpha), float(gamma), float(epsilon), ptr(logits), ptr(labels),
             ptr(keep), ptr(ws), ptr(out2))
        ctx.save_for_backward(logits, labels, keep, out2)
        ctx.cfg = (float(alpha), float(gamma), float(epsilon))
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        from ...ext._common import call, ptr
        logits, labels, keep, out2 = ctx.saved_tensors
        b, c, n = logits.shape
        up = g.reshape(1).float().contiguous()
        grad = torch.empty_like(logits)
        call("geot_poly1_focal_grad", logits.device, b, c, n, *ctx.cfg, ptr(logits), ptr(labels), ptr(keep), ptr(out2), ptr(up),
             ptr(grad))
        return grad, None, None, None, None, None


def _check_labels(logits, labels):
    """GEOT_CHECK_LABELS=1: the synchronous range check F.one_hot does in the reference (openpoints/loss/build.py:223-230
    raise on a label outside [0, C)).  Off by default -- it costs a device round trip per call; without it the fused
    kernel returns NaN for such input (csrc/loss.hip) instead of a silently different loss."""
    if os.environ.get("GEOT_CHECK_LABELS", "0") == "1" and labels.numel():
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0:
            raise RuntimeError("Class values must be non-negative.")
        if hi >= logits.shape[1]:
            raise RuntimeError("Class values must be smaller than num_classes.")


def _fused_ok(mod, logits, labels, reduction_ok):
    return (reduction_ok and not mod.label_is_onehot and mod.weight is None and mod.pos_weight is None and logits.is_cuda
            and logits.dtype == torch.float32 and logits.dim() == 3 and labels.dim() == 2 and labels.dtype == torch.int64
            and tuple(labels.shape) == (logits.shape[0], logits.shape[2]) and logits.numel() > 0
            and logits.shape[0] <= 65535 and logits.shape[1] <= 65535)


def _one_hot_like(logits, labels):
    if labels.ndim == 1:
        labels = F.one_hot(labels, num_classes=logits.shape[1])
    else:
        labels = F.one_hot(labels.unsqueeze(1), logits.shape[1]).transpose(1, -1).squeeze(-1)
    return labels.to(device=logits.device, dtype=logits.dtype)


def _poly1(logits, labels, weight, pos_weight, alpha, gamma, epsilon):
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(input=logits, target=labels, reduction="none", weight=weight,
                                            pos_weight=pos_weight)
    pt = labels * p + (1 - labels) * (1 - p)
    fl = ce * ((1 - pt) ** gamma)
    if alpha >= 0:
        fl = (alpha * labels + (1 - alpha) * (1 - labels)) * fl
    return fl + epsilon * torch.pow(1 - pt, gamma + 1)


class Poly1FocalLoss(torch.nn.Module):
    def __init__(self, epsilon=1.0, alpha=0.25, gamma=2.0, reduction="mean", weight=None, pos_weight=None,
                 label_is_onehot=False, **kwargs):
        super().__init__()
        self.epsilon, self.alpha, self.gamma, self.reduction = epsilon, alpha, gamma, reduction
        self.weight, self.pos_weight, self.label_is_onehot = weight, pos_weight, label_is_onehot

    def forward(self, logits, labels):
        if _fused_ok(self, logits, labels, self.reduction == "mean"):
            _check_labels(logits, labels)
            return _Poly1FocalFn.apply(logits.contiguous(), labels.contiguous(), None, self.alpha, self.gamma, self.epsilon)
        if not self.label_is_onehot:
            labels = _one_hot_like(logits, labels)
        poly1 = _poly1(logits, labels.to(logits.dtype), self.weight, self.pos_weight, self.alpha, self.gamma, self.epsilon)
        if self.reduction == "mean":
            return poly1.mean()
        return poly1.sum() if self.reduction == "sum" else poly1


class Poly1FocalLoss_U_corr(Poly1FocalLoss):
    def forward(self, logits, labels, logits_pred, thresh=0.95, mask=None):
        # a soft (float) mask multiplies the loss by its VALUES in the reference (:872-875): only a 0/1 mask is a `keep` flag
        hard_mask = mask is None or mask.dtype in (torch.bool, torch.uint8)
        if hard_mask and _fused_ok(self, logits, labels, True):
            _check_labels(logits, labels)
            keep = (mask if mask is not None else logits_pred.ge(thresh)).to(torch.uint8).contiguous()
            if tuple(keep.shape) == tuple(labels.shape):
                return _Poly1FocalFn.apply(logits.contiguous(), labels.contiguous(), keep, self.alpha, self.gamma, self.epsilon)
        if not self.label_is_onehot:
            labels = _one_hot_like(logits, labels)
        poly1 = _poly1(logits, labels.to(logits.dtype), self.weight, self.pos_weight, self.alpha, self.gamma, self.epsilon)
        keep = mask if mask is not None else logits_pred.ge(thresh)
        keep = keep.unsqueeze(1).to(poly1.dtype)                      # broadcast over the class axis (:872-875)
        return torch.sum(poly1 * keep) / (keep.sum() * poly1.shape[1] + 0.001)
