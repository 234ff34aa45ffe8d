"""Mirror of openpoints/models/layers/subsample.py:11-156 over ``pointnet2_cuda`` (HIP):
BaseSampler / RandomSample, random_sample, FurthestPointSampling (no origin-skip, tie rule of a <=1024-thread
block), GatherOperation (+ scatter-add backward), fps."""
import math
from abc import ABC, abstractmethod

import torch
from torch.autograd import Function

from ...cpp import pointnet2_cuda


class BaseSampler(ABC):
    """subsample.py:11-50: a sampler is configured by exactly one of num_to_sample (that many points), ratio
    (floor(N * ratio) points) or subsampling_param; num_to_sample excludes the other two."""

    def __init__(self, ratio=None, num_to_sample=None, subsampling_param=None):
        given = [(name, value) for name, value in (("_num_to_sample", num_to_sample), ("_ratio", ratio),
                                                   ("_subsampling_param", subsampling_param)) if value is not None]
        if not given:
            raise Exception('At least ["ratio, num_to_sample, subsampling_param"] should be defined')
        if num_to_sample is not None and len(given) > 1:
            raise ValueError("Can only specify ratio or num_to_sample or subsampling_param, not several !")
        setattr(self, *given[0])        # same precedence as the reference: num_to_sample, ratio, subsampling_param

    def __call__(self, xyz):
        return self.sample(xyz)

    def _get_num_to_sample(self, npoints) -> int:
        fixed = getattr(self, "_num_to_sample", None)
        return fixed if fixed is not None else math.floor(npoints * self._ratio)

    def _get_ratio_to_sample(self, batch_size) -> float:
        ratio = getattr(self, "_ratio", None)
        return ratio if ratio is not None else self._num_to_sample / float(batch_size)

    @abstractmethod
    def sample(self, xyz, feature=None, batch=None):
        ...


class RandomSample(BaseSampler):
    """subsample.py:53-68: uniform indices with replacement, xyz (B,N,3) -> (sampled (B,m,3), idx (B,m))."""

    def sample(self, xyz, **kwargs):
        if len(xyz.shape) != 3:
            raise ValueError(" Expects the xyz tensor to be of dimension 3")
        B, N, _ = xyz.shape
        idx = torch.randint(0, N, (B, self._get_num_to_sample(N)), device=xyz.device)
        return torch.gather(xyz, 1, idx.unsqueeze(-1).expand(-1, -1, 3)), idx


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
