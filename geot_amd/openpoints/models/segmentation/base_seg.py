"""WholePartSeg (openpoints/models/segmentation/base_seg.py:76-160) and Ins_T_mean (:254-263): the thin wrappers
the training step calls.  WholePartSeg concatenates the labelled, strong-view and weak-view clouds into one batch
and runs the segmentor; registry-free -- ``segmentor_args`` is the reference's cfg dict (``NAME`` picks the class)
or a ready module."""
import torch
import torch.nn as nn

from ..backbone import transformer as _tr
from ....ntm import Ins_T_mean  # noqa: F401

_SEGMENTORS = {"PointTransformer_seg_T": _tr.PointTransformer_seg_T}


def build_segmentor(args):
    if isinstance(args, nn.Module):
        return args
    args = dict(args)
    name = args.pop("NAME")
    args.pop("pretrained_path", None)
    if name not in _SEGMENTORS:
        raise KeyError("segmentor %r is not mirrored (only %s)" % (name, sorted(_SEGMENTORS)))
    return _SEGMENTORS[name](**args)


class WholePartSeg(nn.Module):
    def __init__(self, segmentor_args=None, gm_args=None, **kwargs):
        super().__init__()
        self.segmentor = build_segmentor(segmentor_args)

    def forward(self, p0, f0=None, cls0=None, u0=None, if_teacher=False, fixmatch=False):
        if if_teacher:
            p0, f0, cls0 = p0["pos_w"].detach(), p0["x_w"].detach(), p0["cls_w"].detach()
        elif hasattr(p0, "keys"):
            if u0 is not None:
                views = [(p0["pos"], p0["x"], p0["cls"]), (u0["pos_s"], u0["x_s"], u0["cls_s"])]
                if fixmatch:
                    views.append((u0["pos_w"], u0["x_w"], u0["cls_w"]))
                p0, f0, cls0 = (torch.cat([v[i] for v in views], 0) for i in range(3))
            else:
                p0, f0, cls0 = p0["pos"], p0["x"], p0["cls"]
        elif f0 is None:
            f0 = p0.transpose(1, 2).contiguous()
        T = u0["T"] if (u0 is not None and "T" in u0.keys()) else None
        f, p, s, _ = self.segmentor(p0, f0, cls0, T)
        return f, p, s
