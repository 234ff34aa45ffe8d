"""openpoints/cpp/__init__.py:6 re-exports the batch extension as ``pointnet2_cuda``."""
from ...ext import pointnet2_batch_cuda as pointnet2_cuda  # noqa: F401
