from .base_seg import WholePartSeg, Ins_T_mean  # noqa: F401
