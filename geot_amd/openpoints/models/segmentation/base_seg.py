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

    @staticmethod
    def batch_positions(p0, u0=None, if_teacher=False, fixmatch=False):
        """The (B', N, 3) coordinates forward() hands to the segmentor for these arguments (labelled + strong (+ weak) views
        concatenated, or the teacher's weak view)."""
        if if_teacher:
            return p0["pos_w"].detach()
        if hasattr(p0, "keys"):
            if u0 is None:
                return p0["pos"]
            views = [p0["pos"], u0["pos_s"]] + ([u0["pos_w"]] if fixmatch else [])
            return torch.cat(views, 0)
        return p0

    def prefetch_geometry(self, p0, u0=None, if_teacher=False, fixmatch=False):
        """Queue the coordinate-only work of the batch a LATER forward(p0, ..., geometry=<result>) will see
        (PointTransformer_seg_T.prefetch_geometry); None when the segmentor has no such thing."""
        if not hasattr(self.segmentor, "prefetch_geometry"):
            return None
        return self.segmentor.prefetch_geometry(self.batch_positions(p0, u0, if_teacher, fixmatch))

    def forward(self, p0, f0=None, cls0=None, u0=None, if_teacher=False, fixmatch=False, geometry=None):
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
        if geometry is not None:
            # the geometry was computed on a tensor of the same coordinates (batch_positions of the same arguments): take that
            # tensor for the positions, so that the segmentor recognises it; a mismatch in shape means it is not ours
            g = geometry["pts"]
            if g.shape == p0.shape and g.device == p0.device:
                p0 = g
            else:
                geometry = None
        T = u0["T"] if (u0 is not None and "T" in u0.keys()) else None
        if geometry is not None:
            f, p, s, _ = self.segmentor(p0, f0, cls0, T, geometry=geometry)
        else:
            f, p, s, _ = self.segmentor(p0, f0, cls0, T)
        return f, p, s
