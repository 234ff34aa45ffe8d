"""Per-point instance-dependent transition-matrix (NTM) estimation on MI355X.

Mirrors the reference pieces that make up the FixMatch+NTM step:

  sig_t_mean, Ins_T_mean     openpoints/models/backbone/transformer.py:1099-1131,
                             openpoints/models/segmentation/base_seg.py:254-263
  class_transition()         examples/segmentation/train.py:505-545, 556-557 (+ gaussian :835-836)
  correct_logits()           examples/segmentation/train.py:549-552
  threeD_space_loss          utils/insT_loss.py:61-110

Same constructor arguments / call signatures / returned tensors as the reference.  The
per-point (B*N, C, C) work runs in fused HIP kernels (geot_amd/csrc/ntm.hip); the C x C
bookkeeping of class_transition() is vectorised torch (no Python loop over classes, no
host round trips -- the reference does 289 scalar device ops per step there).
"""
import math
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from .ext._common import f32, i32, same_device, need, call, ptr
from .knn_cuda import knn_sorted

LABEL_PROJ = [0, 8, 7, 6, 5, 4, 3, 2, 1, 9, 10, 11, 12, 13, 14, 15, 16]  # train.py:48
NTM_CLASSES = len(LABEL_PROJ)   # = GEOT_NTM_C in include/geot_hip.h: the class count the specialised kernels are built for
NTM_MAX_CLASSES = 32            # any other count up to here runs the run-time-C kernels (csrc/ntm_generic.hip)


def _need_ntm_classes(c, what):
    need(1 <= c <= NTM_MAX_CLASSES, "%s: the per-point NTM kernels take 1..%d classes (a half-wave per matrix row; "
                                    "include/geot_hip.h GEOT_NTM_MAX_C), got %d" % (what, NTM_MAX_CLASSES, c))


# ---------------------------------------------------------------------------------------------
class _SigTMeanFn(Function):
    @staticmethod
    def forward(ctx, p, cm, W):
        p = f32(p.contiguous(), "x", 3)
        cm = f32(cm.contiguous(), "cm", 2)
        W = f32(W.contiguous(), "weight", 3)
        dev = same_device(p, cm, W)
        b, c, n = p.shape
        need(tuple(cm.shape) == (c, c) and tuple(W.shape) == (c, c, 2 * c), "sig_t_mean shape mismatch")
        _need_ntm_classes(c, "sig_t_mean")
        out = torch.empty((b * n, c, c), dtype=torch.float32, device=dev)
        call("geot_ntm_sig_t_mean", dev, b, n, c, ptr(p), ptr(W), ptr(cm), ptr(out))
        ctx.save_for_backward(p, cm, W)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        p, cm, W = ctx.saved_tensors
        b, c, n = p.shape
        g = grad_out.contiguous()
        if c != NTM_CLASSES:
            # run-time class count: d raw from the generic kernel, then ONE library GEMM  G = d raw^T [p | 1]
            # (columns C..2C-1 of a head see the constant cm row: G[:, C] * cm[kk])
            raw = torch.empty_like(g)
            call("geot_ntm_sig_t_mean_grad_raw", p.device, b, n, c, ptr(p), ptr(W), ptr(cm), ptr(g), ptr(raw))
            aug = torch.cat([p.permute(0, 2, 1).reshape(b * n, c), p.new_ones((b * n, 1))], dim=1)
            G = torch.mm(raw.view(b * n, c * c).t(), aug)                   # (C*C, C + 1)
            return None, None, torch.cat([G[:, :c].reshape(c, c, c), G[:, c].reshape(c, c, 1) * cm.unsqueeze(1)], dim=2)
        # the Linear heads' weight gradient, fused: d raw is formed on chip and contracted with [p_i | 1]
        # by a second MFMA GEMM (train.py never needs d/dp: the predictor's input is detached)
        lib = _lib.load()
        ws = torch.empty(int(lib.geot_ntm_sig_t_mean_ws_floats(b, n)), dtype=torch.float32, device=p.device)
        gw = torch.zeros_like(W)
        call("geot_ntm_sig_t_mean_grad_w", p.device, b, n, c, ptr(p), ptr(W), ptr(cm), ptr(g), ptr(gw), ptr(ws))
        return None, None, gw


def sig_t_mean_grad_raw(p, cm, W, grad_out):
    """d loss / d (pre-clamp rows), (B*N, C, C) -- the unfused building block (kept for tests / callers that
    want d raw itself)."""
    b, c, n = p.shape
    g = grad_out.contiguous()
    raw = torch.empty_like(g)
    call("geot_ntm_sig_t_mean_grad_raw", p.device, b, n, c, ptr(p), ptr(W), ptr(cm), ptr(g), ptr(raw))
    return raw


class sig_t_mean(nn.Module):
    """`nclasses` bias-free Linear(2C -> C) heads, one per transition-matrix row
    (transformer.py:1101-1131).  forward(x (B,C,N) softmax, cm (C,C)) -> ins_T (B*N, C, C)."""

    def __init__(self, nclasses):
        super().__init__()
        self.nclasses = nclasses
        self.fc = nn.ModuleList([nn.Linear(nclasses * 2, nclasses, bias=False) for _ in range(nclasses)])

    def stacked_weight(self):
        return torch.stack([l.weight for l in self.fc], dim=0)          # (C, C, 2C), differentiable

    def forward(self, x, cm):
        return _SigTMeanFn.apply(x, cm, self.stacked_weight())


class Ins_T_mean(nn.Module):
    """base_seg.py:254-263 (registry-free: pass the predictor or the class count)."""

    def __init__(self, T_args=None, nclasses=17, **kwargs):
        super().__init__()
        if isinstance(T_args, nn.Module):
            self.T_predictor = T_args
        else:
            if isinstance(T_args, dict):
                nclasses = T_args.get("nclasses", nclasses)
            self.T_predictor = sig_t_mean(nclasses)

    def forward(self, clean, cm):
        return self.T_predictor(clean, cm)


# ---------------------------------------------------------------------------------------------
def gaussian(x, mu, s):
    """train.py:835-836."""
    return (1 / (s * math.sqrt(2 * math.pi))) * torch.exp(-((x - mu) ** 2) / (2 * s ** 2))


class _ClassTransitionFn(Function):
    """The 17 x 17 arithmetic of the class-transition block in one launch (and one for d/d sigma) instead of
    ~40 tiny torch kernels each way; the op-by-op torch version below stays as the reference (GEOT_NTM_CT=torch)."""

    @staticmethod
    def forward(ctx, class_T, sigma, ema_t, proj, geo_lambda, ema_decay):
        class_T, ema_t = f32(class_T.contiguous(), "class_T", 2), f32(ema_t.contiguous(), "ema_t", 2)
        sigma_c, proj = f32(sigma.detach().contiguous(), "sigma", 1), f32(proj.contiguous(), "proj", 1)
        dev = same_device(class_T, sigma_c, ema_t, proj)
        c = class_T.shape[0]
        need(tuple(class_T.shape) == (c, c) and tuple(ema_t.shape) == (c, c) and proj.numel() == c
             and sigma_c.numel() == c, "class transition: class_T / ema_t must be (C, C), sigma / proj (C,)")
        corr, nxt, prior, keep = torch.empty((4, c, c), dtype=torch.float32, device=dev).unbind(0)
        call("geot_ntm_class_transition", dev, c, geo_lambda, ema_decay, ptr(class_T), ptr(sigma_c), ptr(ema_t), ptr(proj),
             ptr(corr), ptr(nxt), ptr(prior), ptr(keep))
        # `keep` = ema_t as it was: a loop holding ema_t in one buffer overwrites it (train.py:556-557) before backward
        ctx.save_for_backward(class_T, sigma_c, keep, proj)
        ctx.consts = (geo_lambda, ema_decay)
        ctx.mark_non_differentiable(nxt)
        return corr, nxt, prior

    @staticmethod
    def backward(ctx, g_corr, g_next, g_prior):
        class_T, sigma_c, ema_t, proj = ctx.saved_tensors
        geo_lambda, ema_decay = ctx.consts
        gs = torch.empty_like(sigma_c)      # written in full by the kernel
        gc = g_corr.contiguous() if g_corr is not None else None
        gp = g_prior.contiguous() if g_prior is not None else None
        call("geot_ntm_class_transition_grad", class_T.device, class_T.shape[0], geo_lambda, ema_decay, ptr(class_T),
             ptr(sigma_c), ptr(ema_t), ptr(proj), ptr(gc), ptr(gp), ptr(gs))
        return None, gs, None, None, None, None


_TRANSITION_CONSTANTS = {}


def _transition_constants(C, dtype, device):
    """(tooth-adjacency projection of the labels (train.py:48), e_0, 1 - e_0) on `device`, built once."""
    key = (C, dtype, str(device))
    if key not in _TRANSITION_CONSTANTS:
        proj = torch.tensor(LABEL_PROJ[:C], dtype=dtype, device=device)
        row0 = torch.zeros(C, dtype=dtype, device=device)
        row0[0] = 1
        _TRANSITION_CONSTANTS[key] = (proj, row0, 1 - row0)
    return _TRANSITION_CONSTANTS[key]


def exchange_anchor_rows(v_star, class_T, group):
    """The one real exchange step of the NTM block when the unlabelled batch is sharded over ranks
    (SURVEY.md section 8e): each rank holds, per class, its most confident point's probability v_star (C,)
    and that point's softmax row class_T (C, C); the global anchor of a class is the row of the rank with the
    largest v_star, the LOWEST rank among equals -- clouds are sharded contiguously by rank, so that is the
    first maximum over the flattened (b, n) order of the whole batch, what a single process computes
    (train.py:519-526).  One all-gather of C * (C + 1) floats (1.2 KB at 17 classes) over RCCL / gloo."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    pack = torch.cat([v_star.unsqueeze(1), class_T], dim=1).contiguous()
    if world == 1:
        parts = pack.unsqueeze(0)
    else:       # ONE collective into one tensor (the list form copies W pieces out afterwards: memcpy nodes under a hipGraph capture)
        flat = torch.empty((world * pack.shape[0], pack.shape[1]), dtype=pack.dtype, device=pack.device)
        dist.all_gather_into_tensor(flat, pack, group=group)                # rank-major concatenation (the form gloo takes too)
        parts = flat.view(world, pack.shape[0], pack.shape[1])              # (W, C, 1 + C)
    r_star = torch.argmax(parts[:, :, 0], dim=0)                          # first (= lowest) rank with the maximum
    cols = torch.arange(class_T.shape[0], device=class_T.device)
    return parts[r_star, cols, 1:]


def _filtered_anchor_rows(eta, q):
    """train.py:510-517 (cfg.filter_outlier): before the arg-max of class cc every probability of that class at or
    above its q-quantile over the unlabelled batch is set to 0 -- IN PLACE in ``eta_corr`` (``robust_eta`` is a view,
    :511-513), so the anchor row read afterwards (:526) sees columns 0..cc already filtered and the later ones not.
    Returns (class_T (C,C), v_star (C,)) with exactly that sequential semantics, vectorised."""
    B, C, N = eta.shape
    flat = eta.transpose(0, 1).reshape(C, B * N)
    thresh = torch.quantile(flat, q, dim=1)                                # (C,), linear interpolation like :511
    filt = torch.where(eta >= thresh.view(1, C, 1), torch.zeros_like(eta), eta)
    best = torch.argmax(filt.transpose(0, 1).reshape(C, B * N), dim=1)     # first maximum over the flattened (b, n)
    b_star, n_star = best // N, best % N
    rows_f = filt[b_star, :, n_star]                                       # (C, C): row cc = filt[b*, :, n*]
    rows_u = eta[b_star, :, n_star]
    done = torch.ones((C, C), dtype=torch.bool, device=eta.device).tril()  # column c' <= cc already filtered
    cols = torch.arange(C, device=eta.device)
    return torch.where(done, rows_f, rows_u), rows_f[cols, cols]


def class_transition(eta, sigma, ema_t, geo_lambda=0.999, ema_decay=0.999, group=None, filter_outlier=False,
                     outlier_q=0.97):
    """The class-level transition estimate of train.py:505-545 + EMA update :556-557.

    `group`: a torch.distributed process group (e.g. dist.group.WORLD) over which the unlabelled batch is
    sharded -- the anchor rows are then the whole batch's (exchange_anchor_rows) and every rank gets the same
    ema_t; None (default) keeps them per rank, which is what the reference does under DDP.

    eta (B_u, C, N): softmax of the weak view (detached), sigma (C,): learnable widths returned by
    the segmentor, ema_t (C, C).  Returns (ema_t_corr, ema_t_next, class_T, prior_T).
    `X / X.sum(1)` is kept exactly as written in the reference (it broadcasts the row sums along
    the last axis, i.e. divides column j by row-sum j)."""
    B, C, N = eta.shape
    eta = eta.detach()
    fused = eta.is_cuda and C <= 32 and eta.dtype == torch.float32 and os.environ.get("GEOT_NTM_CT", "fused") == "fused"
    need(sigma.numel() == C and tuple(ema_t.shape) == (C, C), "class_transition: sigma must be (C,), ema_t (C, C)")
    fused = fused and C <= len(LABEL_PROJ)            # the tooth-adjacency projection has 17 entries (train.py:48)
    empty = B == 0 or N == 0      # a rank of an uneven split that owns no unlabelled cloud: it only takes part in the exchange
    if empty:
        need(group is not None, "class_transition: an empty batch has no anchors (only a rank of a sharded batch may pass one)")
        class_T_e = torch.zeros((C, C), dtype=eta.dtype, device=eta.device)
        v_star_e = torch.full((C,), float("-inf"), dtype=eta.dtype, device=eta.device)
    if fused:      # anchors in one launch (+ the transition block in another) instead of ~6 + ~40 torch launches
        if empty:
            class_T, v_star = class_T_e, v_star_e
        elif filter_outlier:
            class_T, v_star = _filtered_anchor_rows(eta, outlier_q)
        else:
            eta_c = eta.contiguous()
            class_T = torch.empty((C, C), dtype=torch.float32, device=eta.device)
            v_star = torch.empty(C, dtype=torch.float32, device=eta.device)
            call("geot_ntm_class_anchors", eta.device, B, N, C, ptr(eta_c), ptr(class_T), ptr(v_star))
        if group is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
                class_T = exchange_anchor_rows(v_star, class_T, group)
        proj, _, _ = _transition_constants(C, eta.dtype, eta.device)
        ema_t_corr, ema_next, prior_T = _ClassTransitionFn.apply(class_T.contiguous(), sigma, ema_t, proj,
                                                                 float(geo_lambda), float(ema_decay))
        return ema_t_corr, ema_next, class_T, prior_T
    # first maximum of class cc over the flattened (b, n) order, without materialising the two (C, B*N) /
    # (B*N, C) transposes the reference builds: arg-max over n per (b, cc), then the first b that attains it
    if empty:
        class_T, v_star = class_T_e, v_star_e
    elif filter_outlier:
        class_T, v_star = _filtered_anchor_rows(eta, outlier_q)
    else:
        n_best = torch.argmax(eta, dim=2)                                     # (B, C), first maximum along n
        v_best = torch.gather(eta, 2, n_best.unsqueeze(2)).squeeze(2)         # (B, C)
        b_star = torch.argmax(v_best, dim=0)                                  # (C,), first b with the maximum
        n_star = torch.gather(n_best, 0, b_star.unsqueeze(0)).squeeze(0)      # (C,)
        cols = torch.arange(C, device=eta.device)
        class_T = eta[b_star.unsqueeze(1), cols.unsqueeze(0), n_star.unsqueeze(1)]   # (C, C): row cc = eta[b*, :, n*]
        v_star = torch.gather(v_best, 0, b_star.unsqueeze(0)).squeeze(0)
    if group is not None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            class_T = exchange_anchor_rows(v_star, class_T, group)
    proj, row0, keep = _transition_constants(C, eta.dtype, eta.device)   # cached: no host->device copy per step
    prior_T = gaussian(proj.unsqueeze(0), proj.unsqueeze(1), sigma.unsqueeze(1))   # [cc][k]
    prior_T = torch.cat([row0.unsqueeze(0), prior_T[1:] * keep.unsqueeze(0)], dim=0)  # [:,0]=0; [0,0]=1
    prior_T = prior_T / torch.sum(prior_T, 1)
    new_T = geo_lambda * class_T + (1 - geo_lambda) * prior_T
    new_T = torch.cat([class_T[:1], new_T[1:]], dim=0)
    new_T = new_T / torch.sum(new_T, 1)
    ema_t_corr = ema_t * ema_decay + new_T * (1 - ema_decay)
    ema_t_corr = ema_t_corr / torch.sum(ema_t_corr, 1)
    ema_next = ema_t * ema_decay + class_T * (1 - ema_decay)
    ema_next = ema_next / torch.sum(ema_next, 1)
    return ema_t_corr, ema_next, class_T, prior_T


# ---------------------------------------------------------------------------------------------
class _CorrectFn(Function):
    @staticmethod
    def forward(ctx, logits, ins_T, ema_t, lam):
        logits = f32(logits.contiguous(), "logits", 3)
        ins_T = f32(ins_T.contiguous(), "ins_T", 3)
        ema_t = f32(ema_t.contiguous(), "ema_t", 2)
        dev = same_device(logits, ins_T, ema_t)
        b, c, n = logits.shape
        need(tuple(ins_T.shape) == (b * n, c, c) and tuple(ema_t.shape) == (c, c), "correct_logits shape mismatch")
        _need_ntm_classes(c, "correct_logits")
        out = torch.empty_like(logits)
        call("geot_ntm_correct", dev, b, n, c, float(lam), ptr(logits), ptr(ins_T), ptr(ema_t), ptr(out))
        ctx.save_for_backward(logits, ins_T, ema_t)
        ctx.lam = float(lam)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        logits, ins_T, ema_t = ctx.saved_tensors
        b, c, n = logits.shape
        g = grad_out.contiguous()
        gl = torch.empty_like(logits)
        gi = torch.empty_like(ins_T)
        ge = torch.zeros_like(ema_t)
        # (the workspace carries per-workgroup partial sums of grad_ema_t for the 17-class kernel; other counts ignore it)
        ws = torch.empty(int(_lib.load().geot_ntm_correct_ws_floats(b, n)), dtype=torch.float32, device=logits.device)
        call("geot_ntm_correct_grad_ws", logits.device, b, n, c, ctx.lam, ptr(logits), ptr(ins_T), ptr(ema_t),
             ptr(g), ptr(gl), ptr(gi), ptr(ge), ptr(ws))
        return gl, gi, ge, None


def correct_logits(pred_u_strong, ins_T, ema_t_corr, lam):
    """train.py:549-552 fused: returns pred_u_strong_corr (B_u, C, N) =
    bmm(logits_i (1,C), normalize(lam*ema_t_corr + (1-lam)*ins_T_i, p=1, dim=-1)); newT is never
    materialised.  Differentiable w.r.t. the logits, ins_T and ema_t_corr."""
    return _CorrectFn.apply(pred_u_strong, ins_T, ema_t_corr, lam)


# ---------------------------------------------------------------------------------------------
class _ThreeDLossFn(Function):
    @staticmethod
    def forward(ctx, positions, labels, ins_T, nbr, sigma, order=None):
        positions = f32(positions.contiguous(), "positions", 3)
        ins_T = f32(ins_T.contiguous(), "ins_T", 3)
        labels = i32(labels.contiguous(), "labels", 2)
        nbr = i32(nbr.contiguous(), "nbr", 3)
        dev = same_device(positions, labels, ins_T, nbr)
        b, n, _ = positions.shape
        c = ins_T.shape[1]
        k = nbr.shape[2]
        need(tuple(ins_T.shape) == (b * n, c, c) and tuple(labels.shape) == (b, n) and tuple(nbr.shape) == (b, n, k),
             "threeD_space_loss shape mismatch")
        _need_ntm_classes(c, "threeD_space_loss")
        per_point = torch.empty(b * n, dtype=torch.float32, device=dev)
        if order is None:
            order = spatial_order(positions)      # processing order only: neighbour rows then hit L2
        mode = os.environ.get("GEOT_NTM_GRAD", "graph")   # graph | gather | atomic (A/B tests)
        if c != NTM_CLASSES:
            mode = "atomic"      # the reverse-adjacency kernels are built for the 17-class rows; other counts scatter
        graph = None
        if ctx.needs_input_grad[2] and mode == "graph":
            # the forward pass leaves the reverse adjacency behind: the backward is then the gather alone
            gbytes = int(_lib.load().geot_ntm_threed_graph_bytes(b, n, k))
            graph = torch.empty(gbytes, dtype=torch.uint8, device=dev)
            call("geot_ntm_threed_loss_fwd_graph", dev, b, n, c, k, float(sigma), ptr(positions), ptr(labels),
                 ptr(ins_T), ptr(nbr), ptr(order), ptr(per_point), ptr(graph), gbytes)
        else:
            call("geot_ntm_threed_loss_ord", dev, b, n, c, k, float(sigma), ptr(positions), ptr(labels), ptr(ins_T),
                 ptr(nbr), ptr(order), ptr(per_point))
        ctx.save_for_backward(positions, labels, ins_T, nbr, order, graph)
        ctx.sigma = float(sigma)
        ctx.mode = mode
        return per_point.mean()

    @staticmethod
    def backward(ctx, grad_out):
        positions, labels, ins_T, nbr, order, graph = ctx.saved_tensors
        b, n, _ = positions.shape
        c, k = ins_T.shape[1], nbr.shape[2]
        if graph is not None:   # writes g in full; the upstream gradient stays on the device (no host sync)
            g = torch.empty_like(ins_T)
            up = grad_out.reshape(1).float().contiguous()
            call("geot_ntm_threed_loss_grad_graph", positions.device, b, n, c, k, 1.0 / (b * n), ptr(up), ptr(ins_T),
                 ptr(nbr), ptr(order), ptr(graph), graph.numel(), ptr(g))
            return None, None, g, None, None, None
        g = torch.zeros_like(ins_T)
        scale = 1.0 / (b * n)        # the upstream gradient is applied on the device below: no host synchronisation
        if ctx.mode == "atomic":     # the scatter form
            call("geot_ntm_threed_loss_grad", positions.device, b, n, c, k, ctx.sigma, scale, ptr(positions),
                 ptr(labels), ptr(ins_T), ptr(nbr), ptr(g))
        else:                        # graph rebuilt here (callers that did not keep the forward's)
            nbytes = int(_lib.load().geot_ntm_threed_loss_ws_bytes(b, n, k))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=positions.device)
            call("geot_ntm_threed_loss_grad_ws", positions.device, b, n, c, k, ctx.sigma, scale, ptr(positions),
                 ptr(labels), ptr(ins_T), ptr(nbr), ptr(order), ptr(g), ptr(ws), nbytes)
        return None, None, g * grad_out.reshape(()), None, None, None


@torch.no_grad()
def spatial_order(positions):
    """(B,N,3) -> int32 (B*N,) global point ids sorted by (cloud, Morton cell): the order in which the graph
    kernels walk the points (GEOT_NTM_ORDER=off: None, memory order)."""
    if os.environ.get("GEOT_NTM_ORDER", "on") == "off":
        return None
    b, n, _ = positions.shape
    lib = _lib.load()
    nbytes = int(lib.geot_knn_grid_ws_bytes(b, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=positions.device)
    order = torch.empty(b * n, dtype=torch.int32, device=positions.device)
    call("geot_spatial_order", positions.device, b, n, ptr(positions), ptr(order), ptr(ws), nbytes)
    return order


class threeD_space_loss(nn.Module):
    """utils/insT_loss.py:61-110.  forward(positions (B,N,3), labels (B,N), ins_T (B*N,C,C)) -> scalar.
    Neighbours = the k nearest other points (reference: knn_point(k+1)[..., 1:], i.e. the nearest hit
    is dropped as "self"); the N x N distance matrix and the (BN, k, C*C) gathers are never built."""

    def __init__(self, k=7, sigma=1.0, num_classes=17):
        super().__init__()
        self.k = k
        self.sigma = sigma
        self.num_classes = num_classes

    @torch.no_grad()
    def neighbours(self, positions):
        _, idx = knn_sorted(positions.contiguous().float(), positions.contiguous().float(), self.k + 1)
        return idx[:, :, 1:].contiguous()

    def forward(self, positions, labels, ins_T, nbr=None, order=None):
        """nbr / order: the kNN graph (self.neighbours) and the processing order (spatial_order) of `positions`
        when the caller has already computed them -- e.g. on a side stream, beside the segmentor's own work: they
        depend on the coordinates only."""
        if nbr is None:
            nbr = self.neighbours(positions)
        return _ThreeDLossFn.apply(positions, labels.to(torch.int32), ins_T, nbr, self.sigma, order)


# ---------------------------------------------------------------------------------------------
class _FeatureLossFn(Function):
    @staticmethod
    def forward(ctx, feats, labels, ins_T, nbr, sigma):
        feats = f32(feats.contiguous(), "logits", 3)
        ins_T = f32(ins_T.contiguous(), "ins_T", 3)
        labels = i32(labels.contiguous(), "labels", 2)
        nbr = i32(nbr.contiguous(), "nbr", 3)
        dev = same_device(feats, labels, ins_T, nbr)
        b, n, d = feats.shape
        c, k = ins_T.shape[1], nbr.shape[2]
        need(tuple(ins_T.shape) == (b * n, c, c) and tuple(labels.shape) == (b, n) and tuple(nbr.shape) == (b, n, k),
             "feature_space_loss shape mismatch")
        _need_ntm_classes(c, "feature_space_loss")
        per_point = torch.empty(b * n, dtype=torch.float32, device=dev)
        call("geot_ntm_feature_loss", dev, b, n, c, k, d, float(sigma), ptr(feats), ptr(labels), ptr(ins_T),
             ptr(nbr), ptr(per_point))
        ctx.save_for_backward(feats, labels, ins_T, nbr)
        ctx.sigma = float(sigma)
        return per_point.sum() / (b * n * k)

    @staticmethod
    def backward(ctx, grad_out):
        feats, labels, ins_T, nbr = ctx.saved_tensors
        b, n, d = feats.shape
        c, k = ins_T.shape[1], nbr.shape[2]
        g = torch.zeros_like(ins_T)
        call("geot_ntm_feature_loss_grad", feats.device, b, n, c, k, d, ctx.sigma, 1.0 / (b * n * k),
             ptr(feats), ptr(labels), ptr(ins_T), ptr(nbr), ptr(g))
        return None, None, g * grad_out.reshape(()), None, None      # upstream gradient stays on the device


class feature_space_loss(nn.Module):
    """utils/insT_loss.py:9-58 (use_feat_loss; off in the shipped cfg).  forward(logits (B,C,N), labels
    (B,N), ins_T (B*N,C,C)) -> mean_{i,j} s_ij exp(-|l_i-l_j|^2/(2 sigma^2)) |T_i-T_j|^2 over the k nearest
    other points in LOGIT space, s_ij = +1 for equal labels else -1.  Weights are detached, as there."""

    def __init__(self, k=7, sigma=1.0, num_classes=17):
        super().__init__()
        self.k, self.sigma, self.num_classes = k, sigma, num_classes

    def forward(self, logits, labels, ins_T, nbr=None):
        feats = logits.detach().permute(0, 2, 1).contiguous()
        if nbr is None:
            from .openpoints.models.layers.knn import knn_point
            nbr = knn_point(self.k + 1, feats, feats)[1][:, :, 1:]
        return _FeatureLossFn.apply(feats, labels.to(torch.int32), ins_T, nbr.to(torch.int32), self.sigma)


class Idenyity_loss(nn.Module):
    """utils/insT_loss.py:117-132 (sic): mean_i sum((T_i - I)^2 * I) / sum(I), without the (BN,C,C) repeat."""

    def forward(self, insT, Identity):
        ident = Identity.reshape(1, -1)
        diff = (insT.reshape(insT.size(0), -1) - ident).pow(2)
        return (torch.sum(diff * ident, dim=1) / torch.sum(ident)).mean()


@torch.no_grad()
def cal_mean_feature(batches, num_classes=17):
    """train.py:868-897 with the model call factored out: ``batches`` yields (logits (B,C,N) raw model
    output, target (B,N) int64).  Reproduces the reference arithmetic literally -- including
    ``cur_feats = logits[target]`` (rows of the flattened softmax selected BY LABEL VALUE, not by a
    class mask), so every visited class receives the same running mean."""
    cm = c_num = None
    for logits, target in batches:
        c = num_classes
        if cm is None:
            cm = torch.zeros((c, c), device=logits.device)
            c_num = torch.zeros((c,), device=logits.device)
        b, _, n = logits.shape
        sm = torch.softmax(logits, dim=1).permute(0, 2, 1).contiguous().view(b * n, c)
        target = target.view(-1)
        mean_feats = sm[target].mean(0)
        counts = torch.bincount(target, minlength=c)[:c].to(cm.dtype)
        seen = counts > 0
        upd = (cm * c_num[:, None] + mean_feats[None, :] * counts[:, None]) / (c_num + counts).clamp_min(1)[:, None]
        cm = torch.where(seen[:, None], upd, cm)
        c_num = c_num + counts
    return cm.to(torch.float32)
