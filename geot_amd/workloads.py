"""Synthetic hot-path workloads at the BASELINE shapes, shared by bench.py and the tests.

``backbone_hotpath_step`` replays, with the reference's shapes and call order, every sampling /
grouping / interpolation op that one forward+backward of the configured backbone
``PointTransformer_seg_T`` issues per batch -- and nothing else:
the dense layers between them (mini-PointNet encoder, 12 transformer blocks, SharedMLP / Conv2d /
GroupNorm stacks) are stock PyTorch in the reference and out of scope here, so they are replaced by
nothing (features are random tensors of the right shape).  ``ntm_step`` does the same for the
unlabelled half of a FixMatch+NTM step (train.py:505-571).
"""
import os

import torch

from .pointnet2 import pointnet2_utils as pu
from .pointops.functions import pointops
from .knn_cuda import KNN
from .openpoints.models.backbone.transformer_ops import Group, get_graph_feature
from . import ntm as ntm_mod

TRANS_DIM, GROUPS, GROUP_SIZE = 384, 512, 32   # cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml:6-15


def _fp(unknown, known, feats):
    """PointnetFPModule front end (pointnet2_modules.py:619-626): three_nn -> weights -> interpolate, as the
    one fused op the FP modules use (GEOT_FP_IMPL=chain: the reference's op-by-op chain, for A/B runs)."""
    if os.environ.get("GEOT_FP_IMPL", "fused") == "fused":
        return pu.fp_interpolate_concat(unknown, known, None, feats)
    dist, idx = pu.three_nn(unknown, known)
    r = 1.0 / (dist + 1e-8)
    return pu.three_interpolate(feats, idx, r / torch.sum(r, dim=2, keepdim=True))


class BackboneHotPath(torch.nn.Module):
    def __init__(self, overlap=True):
        super().__init__()
        self.group = Group(GROUPS, GROUP_SIZE)
        self.knn4 = KNN(k=4, transpose_mode=False)
        self.overlap = overlap
        self._side = {}

    def _side_stream(self, device):
        key = str(device)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def forward(self, pts, tokens):
        """pts (B,N,3); tokens (B,384,512) stands for the transformer output at the 512 group centres.
        Returns the (B,384,N) propagated features (sum of the interpolation outputs feeds backward)."""
        with pointops.fps_prefix_scope():
            return self._forward(pts, tokens)

    def _forward(self, pts, tokens):
        B, N, _ = pts.shape
        # FPS keeps one CU per cloud busy for milliseconds and leaves the other ~250 idle: the patch-embedding
        # front end (FPS 512 + kNN 32 + gather), which does not depend on the 8192-point FPS, runs beside it on
        # a second HIP stream (the C ABI launches on torch's current stream, so `with torch.cuda.stream` is
        # all it takes); the main stream waits for it before the first consumer.
        side = self._side_stream(pts.device) if self.overlap else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(pts.device))
            with torch.cuda.stream(side):
                neighborhood, center, _ = self.group(pts)                          # FPS 512 + kNN 32 + gather
        else:
            neighborhood, center, _ = self.group(pts)
        c8192, c4096 = pointops.fps(pts, 8192), pointops.fps(pts, 4096)            # one FPS run (prefix reuse)
        pointops.fps(pts, 2048)                                                    # computed, unused (reference)
        if side is not None:
            # no record_stream: every side-stream allocation of a forward starts behind `side.wait_stream(main)` of
            # that forward, i.e. behind every earlier main-stream use of whatever block it gets
            torch.cuda.current_stream(pts.device).wait_stream(side)
        f_l2 = _fp(c4096, center, tokens)                                          # propogation_2: 512 -> 4096
        f_l1 = _fp(c8192, center, tokens)                                          # propogation_1: 512 -> 8192
        ct, c4t, c8t = (center.transpose(1, 2).contiguous(), c4096.transpose(1, 2).contiguous(),
                        c8192.transpose(1, 2).contiguous())
        g2a = get_graph_feature(self.knn4, c4t, f_l2, ct, tokens)                  # dgcnn_pro_2: 512 -> 4096
        g2b = get_graph_feature(self.knn4, c4t, f_l2, c4t, f_l2)                   #              4096 -> 4096
        g1a = get_graph_feature(self.knn4, c8t, f_l1, c4t, f_l2)                   # dgcnn_pro_1: 4096 -> 8192
        g1b = get_graph_feature(self.knn4, c8t, f_l1, c8t, f_l1)                   #              8192 -> 8192
        f_l1 = f_l1 + g1a[:, :TRANS_DIM].amax(-1) + g1b[:, :TRANS_DIM].amax(-1)    # stand-in for conv+max
        f_l2 = f_l2 + g2a[:, :TRANS_DIM].amax(-1) + g2b[:, :TRANS_DIM].amax(-1)
        f_l0 = _fp(pts, c8192, f_l1)                                               # propogation_0: 8192 -> N
        return f_l0, f_l2, neighborhood


def backbone_hotpath_step(model, pts, tokens):
    """forward + backward of the hot-path ops; returns the scalar that was differentiated."""
    tokens = tokens.detach().requires_grad_(True)
    f_l0, f_l2, neighborhood = model(pts, tokens)
    loss = f_l0.square().mean() + f_l2.mean() + neighborhood.mean()
    loss.backward()
    return loss.detach()


class NtmHotPath(torch.nn.Module):
    """Unlabelled half of a FixMatch+NTM step at threed_k=32, sigma=1, lambda=0.9 (yaml:78-96)."""

    def __init__(self, num_classes=17, k=32):
        super().__init__()
        self.predictor = ntm_mod.Ins_T_mean(nclasses=num_classes)
        self.loss3d = ntm_mod.threeD_space_loss(k=k, sigma=1.0, num_classes=num_classes)
        self.sigma = torch.nn.Parameter(torch.ones(num_classes))
        self.register_buffer("ema_t", torch.eye(num_classes) * 0.9 + 0.1 / num_classes)
        self.register_buffer("cm", torch.eye(num_classes) * 0.9 + 0.1 / num_classes)
        self.overlap = os.environ.get("GEOT_NTM_OVERLAP", "1") != "0"
        self.overlap_min_points = int(os.environ.get("GEOT_NTM_OVERLAP_MIN", "60000"))
        self.overlap_loss = os.environ.get("GEOT_NTM_OVERLAP_LOSS", "1") != "0"
        self._side = None

    def forward(self, raw_pos, pred_weak, pred_strong):
        # The kNN graph and the processing order depend on the coordinates only: they are built on a second HIP
        # stream beside the soft-max / class-transition / per-point-matrix / correction chain (the kNN search is
        # latency-bound, that chain HBM-bound: together they fill the chip better than one after the other).
        nbr = order = None
        big = raw_pos.shape[0] * raw_pos.shape[1] >= self.overlap_min_points   # launch-bound below: the stream hand-offs cost more than they hide
        if self.overlap and big:
            dev = raw_pos.device
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                nbr = self.loss3d.neighbours(raw_pos)
                order = ntm_mod.spatial_order(raw_pos)
        eta = torch.softmax(pred_weak.detach(), dim=1)
        _, label_u = torch.max(eta, dim=1)
        ema_corr, ema_next, _, _ = ntm_mod.class_transition(eta, self.sigma, self.ema_t)
        ins_t = self.predictor(torch.softmax(pred_strong, dim=1).detach(), self.cm)
        if nbr is not None and self.overlap_loss:
            # ... and the graph loss itself stays on that stream: its forward runs beside the logit correction, and --
            # autograd replays every node on its forward stream -- its L2-bound gradient gather beside the HBM-bound
            # correction backward.  The two meet again in the per-point-matrix backward (sum of both ins_T gradients).
            # ins_t / label_u (allocated on main) are read on the side stream: both outlive the join below (autograd
            # saves them), so no block of theirs can be recycled under the side stream's kernels -- no record_stream
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                loss3d = self.loss3d(raw_pos, label_u, ins_t, nbr=nbr, order=order) * 0.1
            corr = ntm_mod.correct_logits(pred_strong, ins_t, ema_corr, 0.9)
            main.wait_stream(self._side)
            self.ema_t.copy_(ema_next.detach())
            return corr, loss3d
        corr = ntm_mod.correct_logits(pred_strong, ins_t, ema_corr, 0.9)
        if nbr is not None:
            main.wait_stream(self._side)
        loss3d = self.loss3d(raw_pos, label_u, ins_t, nbr=nbr, order=order) * 0.1
        self.ema_t.copy_(ema_next.detach())
        return corr, loss3d


def ntm_step(model, raw_pos, pred_weak, pred_strong):
    pred_strong = pred_strong.detach().requires_grad_(True)
    corr, loss3d = model(raw_pos, pred_weak, pred_strong)
    loss = corr.square().mean() + loss3d
    model.zero_grad(set_to_none=True)
    loss.backward()
    return loss.detach()
