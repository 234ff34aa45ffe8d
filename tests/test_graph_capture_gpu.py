"""include/geot_hip.h promises that every entry point is safe to capture into a hipGraph (no allocation, no host
synchronisation, nothing re-configured per call).  One capture + replay per op family, replayed on NEW input data
written into the captured input buffers, compared bit for bit with the eager result on the same data -- then the two
composite steps (SetAbstraction forward, NTM half-step with and without its side stream).

Each family is its own test so that a failure names the family.  Round 1 recorded a fault on replay of a captured
NTM step; what was changed since is listed in profiles/DESIGN_r01_r03.md section 7 (LDS opt-in raised once per kernel instead of per
call, scratch cleared by a kernel instead of hipMemsetAsync, no record_stream on pool tensors)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cloud(b, n, start):
    from geot_amd.synth import make_batch
    return torch.from_numpy(make_batch(b, n, start_index=start)[0]).to(DEV)


def _capture(fn, warmup=2):
    """torch's documented recipe: warm up on a side stream, then capture `fn` (static inputs are closed over)."""
    s = torch.cuda.Stream(device=DEV)
    s.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(s):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream(DEV).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def _check(fn, inputs, fresh, n_replays=2):
    """`fn()` reads the tensors in `inputs`; capture it, overwrite the inputs with `fresh` data, replay, and compare
    every output with an eager call on the same data."""
    g, out = _capture(fn)
    for _ in range(n_replays):
        for dst, src in zip(inputs, fresh):
            dst.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in (out if isinstance(out, (tuple, list)) else [out])]
        want = fn()
        torch.cuda.synchronize()
        want = want if isinstance(want, (tuple, list)) else [want]
        for a, b in zip(got, want):
            if a.dtype.is_floating_point:      # outputs summed with float atomics differ in the last bits from run to run
                tol = 2e-5 * float(b.abs().max()) + 1e-12
                assert torch.allclose(a, b, rtol=2e-5, atol=tol), float((a - b).abs().max())
            else:
                assert torch.equal(a, b)
        fresh = [f.flip(0) if f.shape[0] > 1 else f for f in fresh]   # different data for the next replay


def test_capture_fps():
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.pointops.functions import pointops
    x = _cloud(2, 24000, 0)
    _check(lambda: (pu.furthest_point_sample(x, 512), pointops.fps(x, 2048)), [x], [_cloud(2, 24000, 7)])


def test_capture_neighbour_searches():
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.knn_cuda import knn_sorted
    x = _cloud(2, 24000, 1)
    q = x[:, :6000].contiguous()

    def fn():
        d2, i = knn_sorted(x, x, 9)                       # grid path (large problem)
        d3, i3 = pu.three_nn(x, q)
        return d2, i, d3, i3, pu.ball_query(0.1, 32, x, q)
    y = _cloud(2, 24000, 9)
    _check(fn, [x, q], [y, y[:, :6000].contiguous()])


def test_capture_gathers_and_their_gradients():
    from geot_amd.pointnet2 import pointnet2_utils as pu
    x = _cloud(2, 8192, 2)
    known = x[:, :2048].contiguous()
    feats = torch.randn(2, 64, 2048, device=DEV, requires_grad=True)
    idx = torch.randint(0, 2048, (2, 1024, 16), device=DEV, dtype=torch.int32)
    grad = torch.zeros(2, 64, 2048, device=DEV)

    def fn():
        dist, i3 = pu.three_nn(x, known)
        r = 1.0 / (dist + 1e-8)
        up = pu.three_interpolate(feats, i3, r / r.sum(2, keepdim=True))
        grouped = pu.grouping_operation(feats, idx)
        loss = up.square().sum() + grouped.sum()
        g, = torch.autograd.grad(loss, feats)
        grad.copy_(g)
        return up.detach(), grouped.detach(), grad
    y = _cloud(2, 8192, 11)
    _check(fn, [x, known], [y, y[:, :2048].contiguous()])


def test_capture_sa_forward():
    from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(0)
    sa = PointnetSAModuleVotes(mlp=[3, 64, 64, 128], npoint=1024, radius=0.1, nsample=32, use_xyz=True).to(DEV).eval()
    x = _cloud(2, 8192, 3)
    f = torch.randn(2, 3, 8192, device=DEV)

    def fn():
        with torch.no_grad():
            new_xyz, feats, inds = sa(x, f)
        return new_xyz, feats, inds
    _check(fn, [x, f], [_cloud(2, 8192, 13), torch.randn(2, 3, 8192, device=DEV)])


def test_capture_edgeconv_tail():
    from geot_amd.openpoints.models.backbone.transformer_ops import edgeconv_tail
    p = torch.randn(2, 128, 4096, device=DEV, requires_grad=True)
    q = torch.randn(2, 128, 4096, device=DEV, requires_grad=True)
    idx = torch.randint(0, 4096, (2, 4096, 4), device=DEV, dtype=torch.int32)
    norm = torch.nn.GroupNorm(4, 128).to(DEV)
    gp, gq = torch.zeros_like(p), torch.zeros_like(q)

    def fn():
        out = edgeconv_tail(p, q, idx, norm, 0.2)
        a, b = torch.autograd.grad(out.square().sum(), (p, q))
        gp.copy_(a)
        gq.copy_(b)
        return out.detach(), gp, gq
    _check(fn, [p.data, q.data], [torch.randn(2, 128, 4096, device=DEV), torch.randn(2, 128, 4096, device=DEV)])


def _ntm_inputs(b, n, start):
    from geot_amd.synth import make_batch, make_logits
    xyz = make_batch(b, n, start_index=start)[0]
    return (torch.from_numpy(xyz).to(DEV), torch.from_numpy(make_logits(xyz, index=start)).to(DEV),
            torch.from_numpy(make_logits(xyz, index=start + 1, sharp=3.0)).to(DEV))


def test_capture_ntm_kernels():
    from geot_amd import ntm
    xyz, pw, ps = _ntm_inputs(2, 6000, 4)
    pred = ntm.Ins_T_mean(nclasses=17).to(DEV)
    crit = ntm.threeD_space_loss(k=8, sigma=1.0)
    cm = torch.eye(17, device=DEV) * 0.9 + 0.1 / 17
    ema = torch.eye(17, device=DEV)
    gw = [torch.zeros_like(l.weight) for l in pred.T_predictor.fc]
    gl = torch.zeros_like(ps)

    def fn():
        logits = ps.detach().requires_grad_(True)
        ins_t = pred(torch.softmax(logits, dim=1).detach(), cm)
        corr = ntm.correct_logits(logits, ins_t, ema, 0.9)
        labels = pw.argmax(1)
        loss = crit(xyz, labels, ins_t) + corr.square().mean()
        grads = torch.autograd.grad(loss, [logits] + [l.weight for l in pred.T_predictor.fc])
        gl.copy_(grads[0])
        for dst, g in zip(gw, grads[1:]):
            dst.copy_(g)
        return (corr.detach(), loss.detach(), gl) + tuple(gw)
    _check(fn, [xyz, pw, ps], list(_ntm_inputs(2, 6000, 14)))


@pytest.mark.parametrize("overlap", [False, True])
def test_capture_ntm_half_step(overlap):
    """The step round 1 could not replay: NtmHotPath (kNN graph + order on a side stream when `overlap`) forward +
    backward as ONE graph; replay == eager on new data."""
    from geot_amd import workloads as wl
    torch.manual_seed(0)
    nt = wl.NtmHotPath().to(DEV)
    nt.overlap = overlap
    nt.overlap_min_points = 0
    xyz, pw, ps = _ntm_inputs(2, 6000, 5)
    names = [n for n, _ in nt.named_parameters()]
    grads = [torch.zeros_like(p) for p in nt.parameters()]
    ema0 = nt.ema_t.clone()

    def fn():
        nt.ema_t.copy_(ema0)                      # the step updates the EMA buffer in place: same start every time
        strong = ps.detach().requires_grad_(True)
        corr, loss3d = nt(xyz, pw, strong)
        loss = corr.square().mean() + loss3d
        gs = torch.autograd.grad(loss, list(nt.parameters()), allow_unused=True)
        for dst, g in zip(grads, gs):
            if g is not None:
                dst.copy_(g)
        return (corr.detach(), loss.detach(), nt.ema_t) + tuple(grads)
    _check(fn, [xyz, pw, ps], list(_ntm_inputs(2, 6000, 15)))
    assert len(names) == len(grads)


def test_capture_lean_transformer_block_and_losses():
    """The small kernels of DESIGN 4.10 (residual + LayerNorm, head split, soft-max gradient, Linear bias gradients,
    fused Poly-1 focal loss): two encoder blocks forward + loss + backward captured as one graph, replay == eager."""
    from geot_amd.openpoints.models.backbone.transformer import TransformerEncoder_h
    from geot_amd.openpoints.loss import Poly1FocalLoss
    torch.manual_seed(0)
    enc = TransformerEncoder_h(embed_dim=384, depth=2, num_heads=4, drop_path_rate=0.0, extract_layers=[2]).to(DEV)
    for blk in enc.blocks:
        blk.attn.lean = blk.mlp.lean = True
    head = torch.nn.Linear(384, 17).to(DEV)
    crit = Poly1FocalLoss()
    x, pos = torch.randn(2, 512, 384, device=DEV), torch.randn(2, 512, 384, device=DEV)
    labels = torch.randint(0, 17, (2, 512), device=DEV)
    params = list(enc.parameters()) + list(head.parameters())
    grads = [torch.zeros_like(p) for p in params]

    def fn():
        feats = enc(x, pos)[0]
        loss = crit(head(feats).transpose(1, 2).contiguous(), labels)
        gs = torch.autograd.grad(loss, params)
        for dst, g in zip(grads, gs):
            dst.copy_(g)
        return (loss.detach(), feats.detach()) + tuple(grads)
    _check(fn, [x, pos], [torch.randn(2, 512, 384, device=DEV), torch.randn(2, 512, 384, device=DEV)])


def test_capture_point_major_fp_stage():
    """The point-major FP stage (csrc/channels_last.hip): Morton order + reverse index + fp_stage_cl forward and backward
    (the BatchNorm backward folded into the row gather) in one capture, replayed on new coordinates and features."""
    from geot_amd import fused_norm as fn
    from geot_amd.pointnet2 import pointnet2_utils as pu
    b, n, m, c, cs = 2, 3000, 700, 256, 5
    pos, known = _cloud(b, n, 3), _cloud(b, n, 3)[:, :m].contiguous()
    torch.manual_seed(0)
    a = torch.randn(b, m, c, device=DEV)
    skip = torch.randn(b, cs, n, device=DEV)
    wb = torch.randn(c, cs, device=DEV)
    up = torch.randn(b, n, c, device=DEV)
    bn = torch.nn.BatchNorm1d(c).to(DEV).train()

    def step():
        d2, idx = pu._ext.three_nn(pos, known)
        weight = pu._ext.fp_weights(d2)
        rix = fn.ReverseIndex(idx, weight, m, fn.local_spatial_order(known))
        a_req = a.detach().requires_grad_(True)
        wb_req = wb.detach().requires_grad_(True)
        z = fn.fp_stage_cl(bn, a_req, idx, weight, skip, wb_req, True, fn.local_spatial_order(pos), rix)
        ga, gwb, ggamma = torch.autograd.grad((z * up).sum(), (a_req, wb_req, bn.weight))
        return z.detach(), ga, gwb, ggamma

    pos2 = _cloud(b, n, 40)
    _check(step, [pos, known, a, skip], [pos2, pos2[:, :m].contiguous(), a.flip(1) * 0.5, skip.flip(2) + 0.1])
