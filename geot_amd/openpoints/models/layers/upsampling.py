"""Mirror of openpoints/models/layers/upsampling.py:11-102: ThreeNN, ThreeInterpolate (fp32 even under
autocast, as the reference's custom_fwd(cast_inputs=float32)), three_interpolation."""
import torch
from torch.autograd import Function

from ...cpp import pointnet2_cuda


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,N,3), known (B,M,3) -> (dist (B,N,3) L2, idx (B,N,3) int32)."""
        assert unknown.is_contiguous() and known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty((B, N, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, N, 3), dtype=torch.int32, device=unknown.device)
        pointnet2_cuda.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, idx, weight):
        """features (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n)."""
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        pointnet2_cuda.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros((B, c, m), dtype=torch.float32, device=grad_out.device)
        pointnet2_cuda.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous().float(), idx, weight,
                                                      grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


def three_interpolation(unknown_xyz, known_xyz, know_feat):
    """Inverse-distance 3-NN interpolation of know_feat (B,C,m) onto unknown_xyz -> (B,C,n)."""
    dist, idx = three_nn(unknown_xyz, known_xyz)
    dist_recip = 1.0 / (dist + 1e-8)
    weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
    return three_interpolate(know_feat, idx, weight)
