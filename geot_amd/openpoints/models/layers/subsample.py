"""Mirror of openpoints/models/layers/subsample.py:11-156 over ``pointnet2_cuda`` (HIP):
BaseSampler / RandomSample, random_sample, FurthestPointSampling (no origin-skip, tie rule of a <=1024-thread
block), GatherOperation (+ scatter-add backward), fps."""
import math
from abc import ABC, abstractmethod

import torch
from torch.autograd import Function

from ...cpp import pointnet2_cuda


class BaseSampler(ABC):
    """subsample.py:11-50: sample exactly num_to_sample points, or floor(N * ratio)."""

    def __init__(self, ratio=None, num_to_sample=None, subsampling_param=None):
        if num_to_sample is not None:
            if ratio is not None or subsampling_param is not None:
                raise ValueError("Can only specify ratio or num_to_sample or subsampling_param, not several !")
            self._num_to_sample = num_to_sample
        elif ratio is not None:
            self._ratio = ratio
        elif subsampling_param is not None:
            self._subsampling_param = subsampling_param
        else:
            raise Exception('At least ["ratio, num_to_sample, subsampling_param"] should be defined')

    def __call__(self, xyz):
        return self.sample(xyz)

    def _get_num_to_sample(self, npoints) -> int:
        return self._num_to_sample if hasattr(self, "_num_to_sample") else math.floor(npoints * self._ratio)

    def _get_ratio_to_sample(self, batch_size) -> float:
        return self._ratio if hasattr(self, "_ratio") else self._num_to_sample / float(batch_size)

    @abstractmethod
    def sample(self, xyz, feature=None, batch=None):
        pass


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
