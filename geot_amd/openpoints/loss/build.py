"""The two criteria the configured FixMatch+NTM step uses (cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml
criterion_args / criterion_u_args): Poly1FocalLoss (openpoints/loss/build.py:183-258) on the labelled clouds and
Poly1FocalLoss_U_corr (:799-892) on the NTM-corrected strong-view logits.  Element-wise torch on (B, 17, N)
tensors -- callers of the hot path, mirrored only so that the step of BASELINE configs[4] closes; same
constructor arguments and forward signatures, same arithmetic."""
import torch
import torch.nn.functional as F


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
        if not self.label_is_onehot:
            labels = _one_hot_like(logits, labels)
        poly1 = _poly1(logits, labels.to(logits.dtype), self.weight, self.pos_weight, self.alpha, self.gamma, self.epsilon)
        if self.reduction == "mean":
            return poly1.mean()
        return poly1.sum() if self.reduction == "sum" else poly1


class Poly1FocalLoss_U_corr(Poly1FocalLoss):
    def forward(self, logits, labels, logits_pred, thresh=0.95, mask=None):
        if not self.label_is_onehot:
            labels = _one_hot_like(logits, labels)
        poly1 = _poly1(logits, labels.to(logits.dtype), self.weight, self.pos_weight, self.alpha, self.gamma, self.epsilon)
        keep = mask if mask is not None else logits_pred.ge(thresh)
        keep = keep.unsqueeze(1).to(poly1.dtype)                      # broadcast over the class axis (:872-875)
        return torch.sum(poly1 * keep) / (keep.sum() * poly1.shape[1] + 0.001)
