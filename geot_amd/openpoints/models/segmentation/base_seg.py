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
    def _position_views(p0, u0=None, if_teacher=False, fixmatch=False):
        """The caller's coordinate tensors that make up the batch, in order."""
        if if_teacher:
            return [p0["pos_w"]]
        if hasattr(p0, "keys"):
            if u0 is None:
                return [p0["pos"]]
            return [p0["pos"], u0["pos_s"]] + ([u0["pos_w"]] if fixmatch else [])
        return [p0]

    @classmethod
    def batch_positions(cls, p0, u0=None, if_teacher=False, fixmatch=False):
        """The (B', N, 3) coordinates forward() hands to the segmentor for these arguments (labelled + strong (+ weak) views
        concatenated, or the teacher's weak view)."""
        views = cls._position_views(p0, u0, if_teacher, fixmatch)
        if if_teacher:
            return views[0].detach()
        return views[0] if len(views) == 1 else torch.cat(views, 0)

    def prefetch_geometry(self, p0, u0=None, if_teacher=False, fixmatch=False, inline=False):
        """Queue the coordinate-only work of the batch a LATER forward(p0, ..., geometry=<result>) will see
        (PointTransformer_seg_T.prefetch_geometry); None when the segmentor has no such thing.  The result remembers WHICH
        tensors it was computed from (`src`: the caller's coordinate tensors and their version counters): forward() takes it
        only for exactly those tensors, unedited."""
        if not hasattr(self.segmentor, "prefetch_geometry"):
            return None
        if inline:       # on the current stream, for a caller that vouches for it (graph_step.py)
            return self.segmentor.prefetch_geometry(self.batch_positions(p0, u0, if_teacher, fixmatch), inline=True)
        g = self.segmentor.prefetch_geometry(self.batch_positions(p0, u0, if_teacher, fixmatch))
        if g is not None:
            g["src"] = tuple((t, t._version) for t in self._position_views(p0, u0, if_teacher, fixmatch))
        return g

    @classmethod
    def weak_view_geometry(cls, geometry, p0, u0):
        """From the geometry of a fixmatch=True batch (labelled + strong + weak views of p0 / u0) the geometry a frozen,
        eval-mode teacher's forward(u0, if_teacher=True) takes: the weak view's slice (slice_geometry) -- the same sampling,
        grouping and index work, not done twice.  None when there is nothing to slice."""
        from ..backbone.transformer import slice_geometry
        if geometry is None:
            return None
        lo = p0["pos"].shape[0] + u0["pos_s"].shape[0]
        g = slice_geometry(geometry, lo, lo + u0["pos_w"].shape[0])
        if g is not None and not g.get("static"):
            g["src"] = tuple((t, t._version) for t in cls._position_views(u0, if_teacher=True))
        return g

    def forward(self, p0, f0=None, cls0=None, u0=None, if_teacher=False, fixmatch=False, geometry=None):
        if geometry is not None and not geometry.get("static"):
            # a geometry describes the tensors it was computed from and nothing else: the same objects, not edited since
            # (a batch of the same SHAPE is not the same batch); anything else is computed in line
            views = self._position_views(p0, u0, if_teacher, fixmatch)
            src = geometry.get("src")
            if src is None or len(src) != len(views) or not all(t is v and t._version == ver for v, (t, ver) in zip(views, src)):
                geometry = None
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
        if geometry is not None and not geometry.get("static"):
            # same coordinates (checked above): hand the segmentor the tensor the geometry was computed on, so that it
            # recognises it (the concatenation above made a new one)
            p0 = geometry["pts"]
        T = u0["T"] if (u0 is not None and "T" in u0.keys()) else None
        if geometry is not None:
            f, p, s, _ = self.segmentor(p0, f0, cls0, T, geometry=geometry)
        else:
            f, p, s, _ = self.segmentor(p0, f0, cls0, T)
        return f, p, s
