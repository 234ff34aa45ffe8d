"""Mirror of the reference's top-level ``pointnet2`` package (wrappers + modules)."""
from ..ext import pointnet2_ext as _ext  # noqa: F401  (exposed as pointnet2._ext by aliases.install)
