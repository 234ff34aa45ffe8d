"""openpoints/cpp/pointnet2_batch/__init__.py:1-2."""
from ....ext import pointnet2_batch_cuda  # noqa: F401
from ....ext import pointnet2_batch_cuda as pointnet2_cuda  # noqa: F401
