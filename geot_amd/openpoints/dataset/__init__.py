"""Dataloader-side ops on the GPU: the reference runs these on the CPU per scan
(openpoints/dataset/grid_sample.py, openpoints/dataset/tooth_semi/tooth_dataset.py:108-147)."""
from .grid_sample import grid_subsampling  # noqa: F401
from .tooth_prep import pc_norm, prepare_sample  # noqa: F401
