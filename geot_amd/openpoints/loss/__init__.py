from .build import Poly1FocalLoss, Poly1FocalLoss_U_corr  # noqa: F401
