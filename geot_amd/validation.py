"""Validation-path upsampling (examples/segmentation/train.py:781-800 ``get_pred_whole``): per scan,
de-normalise the N sampled points, take the 3 nearest sampled points of every full-resolution vertex,
inverse-distance interpolate the class probabilities and arg-max.  Same kernels as the training path
(three_nn / three_interpolate) at m ~ 1e5 unknown vertices; scans keep their own vertex counts, so
the loop over scans stays (each launch already fills the GPU)."""
import torch
import torch.nn.functional as F

from .pointnet2 import pointnet2_utils as pt_utils


@torch.no_grad()
def get_pred_whole(logits, points, points_whole, center, scale):
    """logits (B,C,N); points (B,N,3) normalised; points_whole: list of (M_i,3); center/scale: per-scan
    tensors broadcastable to (1,N,3) -> list of (1, M_i) int64 predicted labels."""
    logits = F.softmax(logits, dim=1)
    dev = logits.device
    preds_whole = []
    for index in range(logits.shape[0]):
        logit = logits[index].unsqueeze(0).contiguous()
        point = points[index].unsqueeze(0).contiguous()
        s = torch.as_tensor(scale[index]).to(dev).unsqueeze(0).contiguous()
        c = torch.as_tensor(center[index]).to(dev).unsqueeze(0).contiguous()
        point_whole = torch.as_tensor(points_whole[index]).to(dev).unsqueeze(0).contiguous()
        point = (point * s + c).contiguous()
        dist, idx = pt_utils.three_nn(point_whole.float(), point.float())
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        logit_whole = pt_utils.three_interpolate(logit, idx, weight)
        preds_whole.append(logit_whole.argmax(dim=1))
    return preds_whole
