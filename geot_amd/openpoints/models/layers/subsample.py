"""Mirror of openpoints/models/layers/subsample.py:42-156 over ``pointnet2_cuda`` (HIP):
random_sample, FurthestPointSampling (no origin-skip, tie rule of a <=1024-thread block),
GatherOperation (+ scatter-add backward), fps."""
import torch
from torch.autograd import Function

from ...cpp import pointnet2_cuda


def random_sample(xyz, npoint):
    B, N, _ = xyz.shape
    return torch.randint(0, N, (B, npoint), device=xyz.device)


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> idx (B,npoint) int32."""
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2_cuda.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)."""
        assert features.is_contiguous() and idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
        pointnet2_cuda.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
        pointnet2_cuda.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


def fps(data, number):
    """data (B,N,C>=3) -> the `number` furthest-sampled rows (B,number,C)."""
    idx = furthest_point_sample(data[:, :, :3].contiguous(), number)
    return torch.gather(data, 1, idx.unsqueeze(-1).long().expand(-1, -1, data.shape[-1]))
