"""Mirrors of the reference's top-level ``utils`` package that sit on the hot path: ``insT_loss`` (lives in
geot_amd/ntm.py) and ``pseudo_mask`` (kNN-based pseudo-label refinement)."""
from ..ntm import threeD_space_loss, feature_space_loss, Idenyity_loss  # noqa: F401  (utils/insT_loss.py)
