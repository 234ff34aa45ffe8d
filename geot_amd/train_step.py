"""The training steps the BASELINE configs time, composed from the mirrored modules -- the part of
examples/segmentation/train.py that drives the hot path, and nothing else of the driver (no loaders,
meters, logging, checkpoints):

* ``SupervisedStep``  -- configs[2]/[3]: forward of the segmentor on a batch of clouds, Poly1FocalLoss,
  backward, (DDP gradient all-reduce when wrapped,) clip + AdamW step (train.py:436-452, 646-657).
* ``FixMatchNTMStep`` -- configs[4]: one semi-supervised iteration (train.py:455-602, 646-660): frozen
  teacher on the weak view -> pseudo labels; student on labelled + strong + weak (WholePartSeg, 6 clouds
  at B_l = B_u = 2); class-level transition (anchors, Gaussian prior, EMA); per-point transition matrices
  (T_predictor); corrected strong logits; 3-D smoothness loss; Poly1Focal losses; backward; both optimisers.

Both are plain callables over device tensors; `ddp()` wraps student and T_predictor the way train.py:159-166
does (SyncBatchNorm + DistributedDataParallel) -- and also wraps T_predictor, which the reference leaves
unsynchronised (last row: an intentional divergence).
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ntm as ntm_mod
from .openpoints.loss import Poly1FocalLoss, Poly1FocalLoss_U_corr
from .openpoints.models.segmentation import WholePartSeg

NTM_CFG = dict(threshold=0.0, unsupervised_loss_weight=1.0, ema_t_decay=0.999, lambma=0.9, geo_lambma=0.999,
               threed_loss_weight=0.1, threed_k=32, threed_sigma=1.0, filter_outlier=False, lr=1e-3,
               weight_decay=1e-4, grad_norm_clip=None, batch_size_l=2, batch_size_u=2, num_classes=17)
# cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml:45-96 (+ default.yaml)


# parameters the configured step never differentiates: T_revision is not used by forward() at all and the
# `correction` output of T_linear is dropped by the caller (train.py:490 `pred_all, delta_T, sigma = ...`, delta_T
# unused); `sigma` only receives a gradient through the class-transition prior of the FixMatch step.  DDP's reducer
# must be told, or it waits for their gradients (the reference never ran its DDP path: train.py:7 pins one GPU).
UNUSED_SUPERVISED = ("T_revision.weight", "T_linear.weight", "sigma")
UNUSED_FIXMATCH = ("T_revision.weight", "T_linear.weight")


def ddp(module, device, sync_bn=True, unused=(), min_world=2, **kw):
    """train.py:159-166: SyncBatchNorm conversion + DistributedDataParallel when a process group of at least `min_world`
    ranks is up (min_world=1: wrap even a single rank -- the one way to put DDP's reducer and RCCL under a step on a
    one-GPU box, tests/test_dist_gpu.py)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() >= min_world):
        return module
    if sync_bn:
        module = nn.SyncBatchNorm.convert_sync_batchnorm(module)
    skip = [n for n, _ in module.named_parameters() if any(n.endswith(u) for u in unused)]
    skip += ["." + n for n in skip if "." not in n]     # DDP spells a root-level parameter "<module_name>.<param>" = ".sigma"
    if skip:
        nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(module, skip)
    ids = [device.index] if device.type == "cuda" else None
    kw.setdefault("gradient_as_bucket_view", True)      # gradients live in the all-reduce buckets: no 108 MB copy per step
    return nn.parallel.DistributedDataParallel(module, device_ids=ids, **kw)


def sync_only(module, sync_bn=True, min_world=2):
    """The data-parallel form of a module for a step that is REPLAYED FROM hipGraphs (geot_amd/graph_step.py): SyncBatchNorm
    conversion as train.py:159-166 does it, but NO DistributedDataParallel wrapper -- the step exchanges its gradients itself
    (GradSync).  Why not wrap and bypass: DDP's reducer keeps every parameter's AccumulateGrad node alive from its
    construction on, on the stream that was current then; a backward captured on the capture stream hops to that stream and
    back for each of them -- forks inside a capture, which this runtime answers with a crash in hipStreamEndCapture
    (profiles/r04_graph_capture_notes.txt; reproduced with DDP in round 5)."""
    import torch.distributed as dist
    if sync_bn and dist.is_available() and dist.is_initialized() and dist.get_world_size() >= min_world:
        module = nn.SyncBatchNorm.convert_sync_batchnorm(module)
    return module


class GradSync:
    """The data-parallel gradient exchange as ONE flat all-reduce, for a step that is replayed from a hipGraph
    (geot_amd/graph_step.py): DistributedDataParallel's reducer is host logic (bucket hooks fired from the autograd engine)
    and cannot be captured, an RCCL collective on the current stream can -- it is a kernel node.  The step holds the bare
    (SyncBatchNorm-converted: sync_only) modules and calls this between the backward and the optimizer: every gradient that
    exists is packed into one static fp32 buffer, divided by the world size (before the sum, as the reducer does), summed
    over the ranks, and copied back.  108 MB at the segmentor's size: one collective instead of five 25-MB buckets; nothing
    overlaps the backward -- the price of a host-free step (train.py:159-166 wraps the model; :646-669 is the loop this
    replaces).  Construction broadcasts rank 0's parameters and buffers, as DDP's does."""

    def __init__(self, modules, group=None, broadcast=True):
        import torch.distributed as dist
        modules = [m.module if hasattr(m, "module") else m for m in modules]
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group)
        self.flat = None
        self.views = None
        self.collectives = 0
        if broadcast and self.world > 1:
            with torch.no_grad():
                for m in modules:
                    for t in list(m.parameters()) + list(m.buffers()):
                        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)

    def __call__(self):
        import torch.distributed as dist
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        sizes = [g.numel() for g in grads]
        if self.flat is None or self.flat.numel() != sum(sizes):
            self.flat = torch.empty(sum(sizes), dtype=grads[0].dtype, device=grads[0].device)
            self.views = list(torch.split(self.flat, sizes))
        flat_grads = [g.reshape(-1) for g in grads]          # (gradients are contiguous: views, not copies)
        torch._foreach_copy_(self.views, flat_grads)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        dist.all_reduce(self.flat, group=self.group)
        self.collectives += 1
        torch._foreach_copy_(flat_grads, self.views)


def _inner(module, bypass):
    """The module a step calls: DDP's wrapped module when the step exchanges its gradients itself (GradSync)."""
    return module.module if (bypass and hasattr(module, "module")) else module


def parameter_groups(model, weight_decay=1e-4, skip_list=()):
    """The two AdamW parameter groups of the reference's optimizer factory (openpoints/optim/optim_factory.py:66-119
    get_parameter_groups, reached through build_optimizer_from_cfg :190-198 with its default filter_bias_and_bn=True,
    train.py:169 / :225): every 1-D parameter (BatchNorm / LayerNorm / GroupNorm weights, `sigma`, ...), every
    `*.bias` and everything named in the model's no_weight_decay() goes WITHOUT weight decay, the rest with it.
    Group order and the `lr_scale` key are the factory's (first appearance; 1.0 without layer decay)."""
    module = model.module if hasattr(model, "module") else model          # DDP wrapper: the factory looks inside (:190-193)
    if not skip_list and hasattr(module, "no_weight_decay"):
        skip_list = module.no_weight_decay()
    groups = {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        no_decay = param.ndim == 1 or name.endswith(".bias") or any(key in name for key in skip_list)
        key = "no_decay" if no_decay else "decay"
        if key not in groups:
            groups[key] = {"weight_decay": 0.0 if no_decay else weight_decay, "params": [], "lr_scale": 1.0}
        groups[key]["params"].append(param)
    return list(groups.values())


def _join(dev, side, *tensors):
    """The one place a side stream's results are handed to the current stream: the current stream waits for `side`,
    and -- outside graph capture, where record_stream is not allowed and the capture's own dependency edges keep the
    memory alive -- every tensor allocated on `side` is recorded as used by the current stream, so the caching allocator
    cannot hand its block to a later allocation on `side` while a reader queued here is still pending (whatever the
    order of frees / allocations a future change introduces)."""
    cur = torch.cuda.current_stream(dev)
    cur.wait_stream(side)
    if not torch.cuda.is_current_stream_capturing():
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(cur)


def make_optimizer(model, lr=1e-3, weight_decay=1e-4):
    """AdamW as train.py:169 / :225 build it (cfgs/tooth_semi/default.yaml:66-68: adamw, weight_decay 1e-4)."""
    groups = parameter_groups(model, weight_decay) if weight_decay else \
        [{"params": [p for p in model.parameters() if p.requires_grad]}]
    params = [p for g in groups for p in g["params"]]
    fused = bool(params) and all(p.is_cuda for p in params)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=0.0 if weight_decay else weight_decay, fused=fused)


_MODE_SENSITIVE = (nn.modules.batchnorm._BatchNorm, nn.modules.dropout._DropoutNd, nn.GroupNorm, nn.LayerNorm)


def _mode(module, training):
    """module.train(training) unless the tree is in that mode already: train.py sets the modes once per epoch
    (train_one_epoch: model.train(), model_t.eval()); walking ~180 modules three times per iteration cost 2 ms of host time.
    "Already" = the root's flag AND the flag of every submodule whose behaviour depends on it (BatchNorm, Dropout, DropPath --
    a short list cached on the module): a child toggled on its own (a BatchNorm put in eval() for a partial validation, a
    helper that flips the teacher's children) is put back instead of silently running in the wrong mode."""
    watch = module.__dict__.get("_geot_mode_watch")
    if watch is None:
        watch = [m for m in module.modules()
                 if isinstance(m, _MODE_SENSITIVE) or type(m).__name__ == "DropPath"]
        module.__dict__["_geot_mode_watch"] = watch
    if module.training != training or any(m.training != training for m in watch):
        module.train(training)


def _lookahead_at_blocks(segmentor, default):
    """Where a step queues its look-ahead: inside the backward, when it reaches the transformer blocks ("blocks"; the model
    must offer the hook), or right behind the forward ("forward").  GEOT_LOOKAHEAD_AT overrides the step's default --
    measured (profiles/r03_ab_lookahead_at*.txt): the supervised step 33.27 ms at "blocks" against 33.61 at "forward" (behind
    the forward the 8192-sample FPS shares the chip with the decoder's widest GEMMs and takes 5.5 instead of 4.8 ms); the
    FixMatch iteration 32.1 ms at "forward" against 32.7 at "blocks" (behind its student forward come the NTM block and the
    losses: a stretch of small kernels that leaves the FPS launches room)."""
    return os.environ.get("GEOT_LOOKAHEAD_AT", default) == "blocks" and hasattr(segmentor, "at_blocks_backward")


class SupervisedStep:
    def __init__(self, model, lr=1e-3, weight_decay=1e-4, grad_norm_clip=None, grad_sync=None):
        self.model = model
        self.criterion = Poly1FocalLoss()
        self.optimizer = make_optimizer(model, lr, weight_decay)
        self.clip = grad_norm_clip
        self._geometry = None          # coordinate-only work of the next batch, queued by the previous call
        self.grad_sync = grad_sync     # a GradSync: the step exchanges its gradients itself (bare modules: sync_only)

    def optimizers(self):
        return [self.optimizer]

    def sync_modules(self):
        return [self.model]

    def __call__(self, pos, cls, target, next_pos=None):
        """pos (B,N,3) f32, cls (B,1) int64 jaw id, target (B,N) int64 -> detached loss.
        next_pos: the coordinates of the NEXT batch, when the loop already holds them (a data loader with one batch of
        look-ahead): their sampling / grouping / index work is queued between this batch's forward and backward
        (PointTransformer_seg_T.prefetch_geometry) and picked up by the next call -- same results, 0.6 ms less per step.
        The queued work is used only if the next call passes that very tensor, unedited (the model checks the tensor, its
        version counter and, for WholePartSeg, the tensors it was assembled from); anything else is computed in line."""
        geometry, self._geometry = self._geometry, None
        loss, self._geometry = self.iteration(pos, cls, target, geometry, next_pos)
        return loss

    def lookahead_work(self, pos):
        """Everything of an iteration that depends on its batch alone (the geometry: Group, the 8192-sample FPS, the index
        plan), on the CURRENT stream -> what iteration(geometry=...) takes.  graph_step.py replays this as a graph of its
        own beside the previous iteration's training graph."""
        _mode(self.model, True)
        inner = self.model.module if hasattr(self.model, "module") else self.model
        return inner.prefetch_geometry(pos, inline=True) if hasattr(inner, "prefetch_geometry") else None

    def forward_loss(self, pos, cls, target, geometry=None):
        """The first half of an iteration: the forward and the loss (with its autograd graph)."""
        _mode(self.model, True)
        logits = _inner(self.model, self.grad_sync is not None)(pos, pos.transpose(1, 2).contiguous(), cls, geometry=geometry)[0]
        return self.criterion(logits, target)

    def backward_update(self, loss):
        """The second half: backward, clipping, the optimizer -> the detached loss."""
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        return loss.detach()

    def forward_backward_head(self, pos, cls, target, geometry=None):
        """The iteration up to the point where the backward reaches the transformer blocks -- forward, loss, the backward of
        the head and the decoder -- for a model that can cut its autograd graph there (cut_at_blocks / take_cut: the
        segmentor); otherwise the whole backward.  -> (detached loss, what backward_rest_update needs)."""
        inner = self.model.module if hasattr(self.model, "module") else self.model
        can_cut = hasattr(inner, "take_cut")
        if can_cut:
            inner.cut_at_blocks = True
        try:
            loss = self.forward_loss(pos, cls, target, geometry)
        finally:
            if can_cut:
                inner.cut_at_blocks = False
        cut = inner.take_cut() if can_cut else None
        loss.backward()
        rest = None if cut is None else (cut[0], [d.grad for d in cut[1]])
        return loss.detach(), rest

    def backward_rest_update(self, rest):
        """The rest of the backward (the blocks, the patch encoder), clipping, the optimizer."""
        if rest is not None:
            torch.autograd.backward(rest[0], rest[1])
        if self.grad_sync is not None:
            self.grad_sync()
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)

    def iteration(self, pos, cls, target, geometry=None, next_pos=None):
        """One iteration -> (detached loss, the geometry queued for next_pos or None)."""
        inner = self.model.module if hasattr(self.model, "module") else self.model
        queued = [None]
        queue = None
        if next_pos is not None and hasattr(inner, "prefetch_geometry"):
            def queue():
                queued[0] = inner.prefetch_geometry(next_pos)
        at_blocks = _lookahead_at_blocks(inner, "blocks")
        if queue is not None and at_blocks:
            inner.at_blocks_backward = queue          # runs inside the backward, when it reaches the transformer blocks
        loss = self.forward_loss(pos, cls, target, geometry)
        if queue is not None and not at_blocks:
            queue()                                   # (GEOT_LOOKAHEAD_AT=forward: right behind the forward)
        loss = self.backward_update(loss)
        if at_blocks:
            inner.at_blocks_backward = None
        return loss, queued[0]


class FixMatchNTMStep:
    """State of train_one_epoch that survives an iteration: ema_t (C,C), cm (C,C), both optimisers."""

    def __init__(self, student, teacher, t_predictor, cm=None, cfg=None, group=None):
        self.cfg = dict(NTM_CFG, **(cfg or {}))
        c = self.cfg["num_classes"]
        self.model, self.model_t, self.T_predictor = student, teacher, t_predictor
        for p in self.model_t.parameters():
            p.requires_grad = False                                                    # train.py:221-222
        dev = next(student.parameters()).device
        self.criterion, self.criterion_u = Poly1FocalLoss(), Poly1FocalLoss_U_corr()
        self.threed_loss = ntm_mod.threeD_space_loss(self.cfg["threed_k"], self.cfg["threed_sigma"], c)
        self.optimizer = make_optimizer(student, self.cfg["lr"], self.cfg["weight_decay"])
        self.T_optimizer = make_optimizer(t_predictor, self.cfg["lr"], self.cfg["weight_decay"])
        self.ema_t = torch.eye(c, device=dev)                                          # train.py:274
        self.cm = cm if cm is not None else torch.full((c, c), 1.0 / c, device=dev)    # cal_mean_feature's output
        self.group = group
        self._side = None
        self._geometry = (None, None)      # coordinate-only work of the next student / teacher batch (look-ahead)
        self._geometry_src = None
        self._teacher_stream = None
        # the frozen teacher's forward on its own stream beside the student's (same results; GEOT_TEACHER_STREAM=0: in line)
        self.overlap_teacher = os.environ.get("GEOT_TEACHER_STREAM", "1") != "0"
        self.share_weak_geometry = os.environ.get("GEOT_SHARE_WEAK_GEOMETRY", "1") != "0"
        self.grad_sync = None          # a GradSync: see SupervisedStep

    def optimizers(self):
        return [self.optimizer, self.T_optimizer]

    def sync_modules(self):
        return [self.model, self.T_predictor]

    def __call__(self, data, data_u, next_batches=None):
        """data: labelled batch {pos (B_l,N,3), x (B_l,3,N), cls (B_l,1), y (B_l,N)}; data_u: unlabelled batch
        {pos_w, x_w, cls_w, pos_s, x_s, cls_s, raw_pos (B_u,N,3)} -> dict of detached losses.
        next_batches = (data', data_u') of the NEXT iteration when the loop already holds them: the coordinate-only work of
        the next student and teacher batches is queued behind this iteration's student forward (SupervisedStep's look-ahead;
        the caller must then pass exactly those dicts next time -- the positions are matched by identity and version
        counter, not trusted)."""
        geoms = self._geometry
        self._geometry = (None, None)
        if geoms[0] is not None and not _same_positions_impl(self._geometry_src, data, data_u):
            geoms = (None, None)         # not the batches the look-ahead was given: do the work in line
        losses, self._geometry = self.iteration(data, data_u, geoms, next_batches)
        if next_batches is not None:
            src = (next_batches[0]["pos"], next_batches[1]["pos_s"], next_batches[1]["pos_w"])
            self._geometry_src = src + (tuple(t._version for t in src),)
        return losses

    def _pseudo_labels(self, data_u, geom_t):
        """train.py:462-475: the frozen teacher on the weak view -> (softmax, its maximum, its arg-max)."""
        with torch.no_grad():
            _mode(self.model_t, False)
            pred_u = F.softmax(self.model_t(data_u, if_teacher=True, geometry=geom_t)[0], dim=1)
            logits_u_aug, label_u_aug = torch.max(pred_u, dim=1)
        return pred_u, logits_u_aug, label_u_aug

    def _teacher_geometry(self, inner, geom_s, data, data_u, inline=False):
        """The teacher's geometry: the weak view is the last third of the student's batch, so its sampling / grouping / index
        work is a slice of the student's (WholePartSeg.weak_view_geometry; the teacher's own 8192-sample FPS was 4.7 ms on two
        CUs beside the student's); computed on its own only when there is nothing to slice (GEOT_SHARE_WEAK_GEOMETRY=0: always)."""
        g = None
        if self.share_weak_geometry and hasattr(inner, "weak_view_geometry"):
            g = inner.weak_view_geometry(geom_s, data, data_u)
        return g if g is not None else self.model_t.prefetch_geometry(data_u, if_teacher=True, inline=inline)

    def lookahead_work(self, data, data_u):
        """Everything of an iteration that depends on its batches and on FROZEN state alone, on the CURRENT stream: the
        student's and the teacher's geometry, the teacher's forward -> pseudo labels (the teacher is never updated:
        train.py:218-222 loads and freezes it; it is in eval mode and draws no random numbers), the kNN graph and Morton
        order of the 3-D loss.  -> the `pre` dict student_iteration() takes.  graph_step.py replays this as a graph of its own
        on a second stream beside the PREVIOUS iteration's training graph; the eager iteration() below spreads the same work
        over side streams inside the iteration instead."""
        _mode(self.model, True)
        inner = self.model.module if hasattr(self.model, "module") else self.model
        with torch.no_grad():
            geom_s = inner.prefetch_geometry(data, data_u, fixmatch=True, inline=True)
            _mode(self.model_t, False)
            geom_t = self._teacher_geometry(inner, geom_s, data, data_u, inline=True)
            raw = data_u["raw_pos"].contiguous()
            nbr = self.threed_loss.neighbours(raw)
            order = ntm_mod.spatial_order(raw)
        pred_u, logits_u_aug, label_u_aug = self._pseudo_labels(data_u, geom_t)
        return {"geom_s": geom_s, "pseudo": (pred_u, logits_u_aug, label_u_aug), "knn": (nbr, order)}

    def iteration(self, data, data_u, geoms=(None, None), next_batches=None):
        """One iteration, eagerly, its batch-only work spread over side streams -> (dict of detached losses, (student,
        teacher) geometries queued for next_batches)."""
        geom_s, geom_t = geoms
        # the kNN graph of the 3-D loss needs raw_pos only: build it beside the teacher / student forwards
        dev = data["pos"].device
        nbr = order = None
        if dev.type == "cuda":
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                raw = data_u["raw_pos"].contiguous()
                nbr = self.threed_loss.neighbours(raw)
                order = ntm_mod.spatial_order(raw)
        # 1. pseudo labels from the frozen teacher on the weak view (train.py:462-475).  The teacher's 2-cloud forward
        #    is short and mostly waits for its own 8192-sample FPS (4.7 ms on 2 CUs); nothing in the student's forward
        #    depends on it, so it is queued on a stream of its own and the student's kernels fill the gap (the eval-mode
        #    teacher draws no random numbers: identical results).  Its outputs are first used at step 3, after the join;
        #    next iteration's entry wait keeps the stream's allocations ordered behind this iteration's readers.
        t_stream = None
        if dev.type == "cuda" and self.overlap_teacher:
            if self._teacher_stream is None:
                self._teacher_stream = torch.cuda.Stream(device=dev)
            t_stream = self._teacher_stream
            t_stream.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(t_stream) if t_stream is not None else contextlib.nullcontext()):
            pseudo = self._pseudo_labels(data_u, geom_t)
        inner = self.model.module if hasattr(self.model, "module") else self.model
        queued = [(None, None)]
        queue = None
        if next_batches is not None:
            nd, nu = next_batches

            def queue():
                _mode(self.model_t, False)
                g_s = inner.prefetch_geometry(nd, nu, fixmatch=True)
                queued[0] = (g_s, self._teacher_geometry(inner, g_s, nd, nu))
        at_blocks = _lookahead_at_blocks(inner.segmentor, "forward")
        if queue is not None and at_blocks:
            inner.segmentor.at_blocks_backward = queue      # runs when the student's backward reaches the transformer blocks

        def after_forward():
            # the teacher joins BEFORE the look-ahead is queued: the look-ahead's side streams (the student's and the
            # teacher's segmentor streams) start behind the current stream only, and a teacher forward without a prefetched
            # geometry allocates its in-line index plan on the teacher segmentor's side stream and frees it when the no_grad
            # forward returns -- behind this join those blocks cannot be handed to the look-ahead's kernels while the
            # teacher's decoder still reads them (the teacher is long done by the end of the student's forward: free)
            if t_stream is not None:
                _join(dev, t_stream, *pseudo)
            if queue is not None and not at_blocks:
                queue()

        def knn_graph():
            if nbr is not None:
                _join(dev, self._side, nbr, order)
            return nbr, order
        losses = self.student_iteration(data, data_u, geom_s, pseudo, knn_graph, after_forward)
        if at_blocks:
            inner.segmentor.at_blocks_backward = None
        return losses, queued[0]

    def student_iteration(self, data, data_u, geom_s, pseudo, knn_graph, after_forward=None, ema_in_place=False,
                          defer_rest=False):
        """Steps 2-5 of the iteration (train.py:478-602, 646-660): the student on labelled + strong + weak views, the class
        transition, the per-point matrices, the corrected logits, the losses, backward, both optimisers.
        pseudo = (pred_u, logits_u_aug, label_u_aug) of the teacher; knn_graph = (nbr, order) of raw_pos or a callable that
        returns them (called where they are first needed); after_forward(): called behind the student's forward (the eager
        iteration joins its teacher stream and queues its look-ahead there); ema_in_place: update ema_t IN its buffer
        (a captured graph holds the buffer, not the attribute); defer_rest: stop where the backward reaches the student's
        transformer blocks (the segmentor cuts its autograd graph there) and return (losses, rest) -- rest() runs the backward
        of the blocks and the patch encoder, the EMA update and both optimisers (graph_step's split capture)."""
        cfg = self.cfg
        seg = getattr(self.model.module if hasattr(self.model, "module") else self.model, "segmentor", None)
        can_cut = defer_rest and hasattr(seg, "take_cut")
        bl, bu = data["pos"].shape[0], data_u["pos_w"].shape[0]
        n = data["pos"].shape[1]
        # 2. student on labelled + strong + weak (train.py:478-492)
        _mode(self.model, True)
        _mode(self.T_predictor, True)
        data_u = dict(data_u, T=self.ema_t)
        if can_cut:
            seg.cut_at_blocks = True
        try:
            pred_all, _, sigma = _inner(self.model, self.grad_sync is not None)(data, u0=data_u, fixmatch=True, geometry=geom_s)
        finally:
            if can_cut:
                seg.cut_at_blocks = False
        cut = seg.take_cut() if can_cut else None
        # (split, not two slices: its backward is one concatenation kernel; a slice's backward copies the gradient into a zero
        # tensor with a device-to-device memcpy -- a memcpy node when the iteration is captured)
        pred_l, pred_u_strong = torch.split(pred_all, [bl, bu, pred_all.shape[0] - bl - bu])[:2]
        if after_forward is not None:
            after_forward()
        pred_u, logits_u_aug, label_u_aug = pseudo
        # 3. class-level transition matrix, prior, EMA (train.py:502-545, 556-557)
        ema_t_corr, ema_next, _, _ = ntm_mod.class_transition(
            pred_u, sigma, self.ema_t, cfg["geo_lambma"], cfg["ema_t_decay"], group=self.group,
            filter_outlier=cfg["filter_outlier"])
        # 4. per-point matrices + corrected strong logits (train.py:547-552)
        ins_t = _inner(self.T_predictor, self.grad_sync is not None)(F.softmax(pred_u_strong, dim=1).detach(), self.cm)
        pred_u_strong_corr = ntm_mod.correct_logits(pred_u_strong, ins_t, ema_t_corr, cfg["lambma"])
        if not ema_in_place:
            with torch.no_grad():      # in its buffer, never rebound: a captured replay (graph_step) holds this very tensor
                self.ema_t.copy_(ema_next)
        # 5. losses (train.py:570-602)
        nbr, order = knn_graph() if callable(knn_graph) else knn_graph
        loss_3d = self.threed_loss(data_u["raw_pos"], label_u_aug, ins_t, nbr=nbr, order=order) * cfg["threed_loss_weight"]
        sup_loss = self.criterion(pred_l, data["y"])
        unsup_loss = self.criterion_u(pred_u_strong_corr, label_u_aug.detach(), logits_u_aug.detach(),
                                      thresh=cfg["threshold"])
        thresh_mask = logits_u_aug.ge(cfg["threshold"])
        unsup_loss = unsup_loss * (cfg["unsupervised_loss_weight"] * (bu * n) / thresh_mask.sum())
        loss = sup_loss + unsup_loss + loss_3d
        loss.backward()
        rest_in = None if cut is None else (cut[0], [d.grad for d in cut[1]])

        def rest():
            if rest_in is not None:
                torch.autograd.backward(rest_in[0], rest_in[1])
            if ema_in_place:
                with torch.no_grad():
                    torch.mul(ema_next, 1.0, out=self.ema_t)   # behind everything that read the old one (a kernel, not a memcpy node)
            if self.grad_sync is not None:
                self.grad_sync()
            if cfg["grad_norm_clip"] is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), cfg["grad_norm_clip"])
            self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)
            self.T_optimizer.step()
            self.T_optimizer.zero_grad(set_to_none=True)
        losses = {"loss": loss.detach(), "sup": sup_loss.detach(), "unsup": unsup_loss.detach(), "threed": loss_3d.detach()}
        if defer_rest:
            return losses, rest
        rest()
        return losses


def _same_positions_impl(src, data, data_u):
    return src is not None and src[0] is data["pos"] and src[1] is data_u["pos_s"] and src[2] is data_u["pos_w"] \
        and all(t._version == v for t, v in zip(src, src[3]))


def build_fixmatch(device, seg_cfg=None, cfg=None, use_ddp=True, group=None, graph_sync=False, min_world=2):
    """Student, frozen teacher and T_predictor as train.py:154-226 builds them (random init: the pretrained
    checkpoints are the authors' local files).  graph_sync: the data-parallel form for a replay from hipGraphs -- bare
    SyncBatchNorm-converted modules + one flat gradient all-reduce (sync_only / GradSync) instead of DDP wrappers."""
    from .openpoints.models.backbone.transformer import TOOTH_SEG_CFG
    seg = dict(NAME="PointTransformer_seg_T", **(seg_cfg or TOOTH_SEG_CFG))
    student = WholePartSeg(segmentor_args=seg).to(device)
    teacher = WholePartSeg(segmentor_args=seg).to(device)
    teacher.load_state_dict(student.state_dict())
    t_pred = ntm_mod.Ins_T_mean(nclasses=(cfg or NTM_CFG).get("num_classes", 17)).to(device)
    if graph_sync:
        import torch.distributed as dist
        student = sync_only(student, min_world=min_world)
        step = FixMatchNTMStep(student, teacher, t_pred, cfg=cfg, group=group)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() >= min_world:
            step.grad_sync = GradSync([student, t_pred], group)
        return step
    if use_ddp:
        student = ddp(student, device, unused=UNUSED_FIXMATCH, min_world=min_world)
        t_pred = ddp(t_pred, device, sync_bn=False, min_world=min_world)
    return FixMatchNTMStep(student, teacher, t_pred, cfg=cfg, group=group)
