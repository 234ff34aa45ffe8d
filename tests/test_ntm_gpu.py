"""GPU suite: the fused NTM kernels against the fp64 restatement (oracle/np_ntm.py), fp32
tolerance 1e-5 relative (sums of <= 289 products; atomically accumulated gradients 1e-4)."""
import numpy as np
import pytest
import torch

from geot_amd.synth import make_batch
from oracle import np_ntm

pytestmark = pytest.mark.gpu
C = 17
DEV = "cuda:0"


def _softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


@pytest.mark.parametrize("B,N", [(1, 64), (2, 500), (3, 1001), (1, 7), (2, 24000)])
def test_sig_t_mean_forward_backward(B, N):
    from geot_amd.ntm import Ins_T_mean
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    p = _softmax(rng.standard_normal((B, C, N)) * 2, 1).astype(np.float32)
    cm = _softmax(rng.standard_normal((C, C)), 1).astype(np.float32)
    mod = Ins_T_mean(nclasses=C).to(DEV)
    W = torch.stack([l.weight for l in mod.T_predictor.fc]).detach().cpu().numpy()
    out = mod(T(p), T(cm))
    want = np_ntm.sig_t_mean(p, cm, W)
    assert out.shape == (B * N, C, C)
    # rows are L1-normalised (entries <= 1): 1e-5 relative to the row scale.  Entries sitting on the
    # 1e-5 clamp come from a 34-term fp32 dot product of O(1) magnitude (abs error ~1e-7).
    # A row whose entries are nearly all clamped has a tiny L1 norm, which amplifies that error;
    # the fp32 reference has the same conditioning, so the bound is absolute on the normalised scale
    # (every row sums to 1): 2e-5.
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    g = rng.standard_normal(want.shape).astype(np.float32)
    (out * T(g)).sum().backward()
    got = torch.stack([l.weight.grad for l in mod.T_predictor.fc]).cpu().numpy()
    # fused weight gradient == unfused building block (d raw kernel) + the Linear layers' own GEMM
    from geot_amd.ntm import sig_t_mean_grad_raw
    raw = sig_t_mean_grad_raw(T(p), T(cm), T(W), T(g))
    aug = torch.cat([T(p).permute(0, 2, 1).reshape(B * N, C), torch.ones((B * N, 1), device=DEV)], 1).double()
    G = raw.view(B * N, C * C).double().t() @ aug
    unfused = torch.cat([G[:, :C].reshape(C, C, C), G[:, C].reshape(C, C, 1) * T(cm).double().unsqueeze(1)], 2)
    np.testing.assert_allclose(got, unfused.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(unfused.abs().max()))
    if N > 5000:
        return
    ref = np_ntm.sig_t_mean_grad_W(p, cm, W, g)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


def test_class_transition_matches_oracle():
    from geot_amd.ntm import class_transition
    rng = np.random.default_rng(1)
    B, N = 2, 3000
    eta = _softmax(rng.standard_normal((B, C, N)) * 3, 1)
    sigma = 0.5 + rng.random(C)
    ema = _softmax(rng.standard_normal((C, C)), 1)
    r = np_ntm.class_transition(eta, sigma, ema)
    sg = T(sigma, torch.float64).requires_grad_(True)
    corr, nxt, cT, pT = class_transition(T(eta, torch.float64), sg, T(ema, torch.float64))
    np.testing.assert_allclose(cT.cpu().numpy(), r["class_T"], rtol=0, atol=0)
    np.testing.assert_allclose(pT.detach().cpu().numpy(), r["prior_T"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(corr.detach().cpu().numpy(), r["ema_t_corr"], rtol=1e-12)
    np.testing.assert_allclose(nxt.cpu().numpy(), r["ema_t_next"], rtol=1e-12)
    corr.sum().backward()          # sigma is learnable through the prior
    assert torch.isfinite(sg.grad).all() and sg.grad.abs().sum() > 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_class_transition_fused_kernel(seed, monkeypatch):
    """fp32 single-launch kernel (+ its d/d sigma) against the oracle and against fp64 torch autograd of the
    op-by-op version; both upstream gradients (ema_t_corr and prior_T) random."""
    from geot_amd.ntm import class_transition
    rng = np.random.default_rng(seed)
    eta = _softmax(rng.standard_normal((2, C, 2000)) * 3, 1)
    sigma = 0.3 + 1.5 * rng.random(C)
    ema = _softmax(rng.standard_normal((C, C)), 1)
    gc, gp = rng.standard_normal((C, C)), rng.standard_normal((C, C))
    r = np_ntm.class_transition(eta.astype(np.float32).astype(np.float64), sigma, ema)
    s32 = T(sigma).requires_grad_(True)
    corr, nxt, cT, pT = class_transition(T(eta), s32, T(ema))
    assert corr.dtype == torch.float32 and corr.grad_fn is not None and "ClassTransition" in type(corr.grad_fn).__name__
    np.testing.assert_allclose(pT.detach().cpu().numpy(), r["prior_T"], rtol=2e-5, atol=1e-30)
    np.testing.assert_allclose(corr.detach().cpu().numpy(), r["ema_t_corr"], rtol=2e-5)
    np.testing.assert_allclose(nxt.cpu().numpy(), r["ema_t_next"], rtol=2e-5)
    assert not nxt.requires_grad
    ((corr * T(gc)).sum() + (pT * T(gp)).sum()).backward()
    monkeypatch.setenv("GEOT_NTM_CT", "torch")
    s64 = T(sigma, torch.float64).requires_grad_(True)
    corr64, _, _, pT64 = class_transition(T(eta).double(), s64, T(ema, torch.float64))
    ((corr64 * T(gc, torch.float64)).sum() + (pT64 * T(gp, torch.float64)).sum()).backward()
    want = s64.grad.cpu().numpy()
    np.testing.assert_allclose(s32.grad.cpu().numpy(), want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize("B,N", [(1, 70), (2, 1000)])
def test_correct_logits_forward_backward(B, N):
    from geot_amd.ntm import correct_logits
    rng = np.random.default_rng(2)
    logits = (rng.standard_normal((B, C, N)) * 2).astype(np.float32)
    insT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2).astype(np.float32)
    E = np_ntm.l1_normalize(rng.random((C, C)) + 0.01, 1).astype(np.float32)
    tl, ti, tE = T(logits).requires_grad_(True), T(insT).requires_grad_(True), T(E).requires_grad_(True)
    out = correct_logits(tl, ti, tE, 0.9)
    _, want = np_ntm.correct_logits(logits, insT, E, 0.9)
    # A row whose entries are nearly all clamped has a tiny L1 norm, which amplifies that error;
    # the fp32 reference has the same conditioning, so the bound is absolute on the normalised scale
    # (every row sums to 1): 2e-5.
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    g = rng.standard_normal(want.shape).astype(np.float32)
    (out * T(g)).sum().backward()
    gl, gi, gE = np_ntm.correct_logits_grads(logits, insT, E, 0.9, g)
    np.testing.assert_allclose(tl.grad.cpu().numpy(), gl, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ti.grad.cpu().numpy(), gi, rtol=1e-4, atol=1e-4 * np.abs(gi).max())
    np.testing.assert_allclose(tE.grad.cpu().numpy(), gE, rtol=1e-3, atol=1e-3 * np.abs(gE).max())


@pytest.mark.parametrize("B,N,k,nlab", [(2, 300, 7, 3), (1, 2000, 32, 17), (2, 1500, 32, 1)])
def test_threed_space_loss_forward_backward(B, N, k, nlab, oracle):
    from geot_amd.ntm import threeD_space_loss
    rng = np.random.default_rng(3)
    xyz, _ = make_batch(B, N, start_index=40, origin_pts=0)
    labels = rng.integers(0, nlab, (B, N))
    insT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2).astype(np.float32)
    crit = threeD_space_loss(k=k, sigma=1.0, num_classes=C)
    pos, tT = T(xyz), T(insT).requires_grad_(True)
    nbr = crit.neighbours(pos)
    widx, _ = oracle.knn_sorted(xyz, xyz, k + 1)
    assert np.array_equal(nbr.cpu().numpy(), widx[:, :, 1:])          # the kNN graph is bit-exact
    loss = crit(pos, T(labels, torch.int64), tT)
    want, wgrad, _ = np_ntm.threed_space_loss(xyz, labels, insT, widx[:, :, 1:], 1.0)
    assert abs(loss.item() - want) <= 2e-5 * abs(want) + 1e-9
    loss.backward()
    np.testing.assert_allclose(tT.grad.cpu().numpy(), wgrad, rtol=1e-3, atol=2e-4 * np.abs(wgrad).max())
    # the processing order is a permutation of all points (and changes nothing but the speed)
    from geot_amd.ntm import spatial_order
    od = spatial_order(pos).cpu().numpy()
    assert np.array_equal(np.sort(od), np.arange(B * N)) and np.all(od // N == np.repeat(np.arange(B), N))
    # the other two backward forms (graph rebuilt in the backward; scatter with atomics) give the same gradient
    import os
    for mode in ("gather", "atomic"):
        os.environ["GEOT_NTM_GRAD"] = mode
        try:
            t2 = T(insT).requires_grad_(True)
            l2 = crit(pos, T(labels, torch.int64), t2)
            l2.backward()
        finally:
            del os.environ["GEOT_NTM_GRAD"]
        assert abs(l2.item() - want) <= 2e-5 * abs(want) + 1e-9
        np.testing.assert_allclose(t2.grad.cpu().numpy(), wgrad, rtol=1e-3, atol=2e-4 * np.abs(wgrad).max())


def test_threed_space_loss_hubs_overflow_the_fixed_lists(oracle):
    """Hundreds of points all listing the same few neighbours (exact duplicates): in-degrees far above the 64
    slots a point has in the forward-built graph, so the overflow list carries most edges."""
    from geot_amd.ntm import threeD_space_loss
    rng = np.random.default_rng(9)
    B, N, k = 1, 900, 8
    xyz = rng.standard_normal((B, N, 3)).astype(np.float32) * 0.3
    xyz[:, 100:700] = xyz[:, 5:6]                       # 600 copies of point 5
    labels = np.zeros((B, N), np.int64)
    insT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2).astype(np.float32)
    crit = threeD_space_loss(k=k, sigma=1.0, num_classes=C)
    pos, tT = T(xyz), T(insT).requires_grad_(True)
    nbr = crit.neighbours(pos)
    widx, _ = oracle.knn_sorted(xyz, xyz, k + 1)
    assert np.array_equal(nbr.cpu().numpy(), widx[:, :, 1:])
    indeg = np.bincount(widx[0, :, 1:].ravel(), minlength=N)
    assert indeg.max() > 64
    loss = crit(pos, T(labels, torch.int64), tT)
    want, wgrad, _ = np_ntm.threed_space_loss(xyz, labels, insT, widx[:, :, 1:], 1.0)
    assert abs(loss.item() - want) <= 2e-5 * abs(want) + 1e-9
    loss.backward()
    np.testing.assert_allclose(tT.grad.cpu().numpy(), wgrad, rtol=1e-3, atol=2e-4 * np.abs(wgrad).max())


@pytest.mark.parametrize("B,N,k,nlab", [(2, 300, 7, 3), (1, 1500, 16, 17)])
def test_feature_space_loss_forward_backward(B, N, k, nlab):
    from geot_amd.ntm import feature_space_loss, Idenyity_loss
    rng = np.random.default_rng(4)
    logits = _softmax(rng.normal(size=(B, C, N)) * 2, 1).astype(np.float32)
    labels = rng.integers(0, nlab, (B, N))
    insT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2).astype(np.float32)
    crit = feature_space_loss(k=k, sigma=1.0, num_classes=C)
    tl, tT = T(logits), T(insT).requires_grad_(True)
    from geot_amd.openpoints.models.layers.knn import knn_point
    feats = tl.permute(0, 2, 1).contiguous()
    nbr = knn_point(k + 1, feats, feats)[1][:, :, 1:]
    # neighbour ids: every returned neighbour is at least as close as the true (k+1)-th (fp64 brute force)
    f64 = logits.transpose(0, 2, 1).astype(np.float64)
    d = np.sqrt(((f64[:, :, None, :] - f64[:, None, :, :]) ** 2).sum(-1)) if N <= 400 else None
    if d is not None:
        kth = np.sort(d, axis=2)[:, :, k]
        got = np.take_along_axis(d, nbr.cpu().numpy().astype(np.int64), 2)
        assert np.all(got <= kth[..., None] + 1e-5)
    loss = crit(tl, T(labels, torch.int64), tT, nbr=nbr)
    want, wgrad, _ = np_ntm.feature_space_loss(logits, labels, insT, nbr.cpu().numpy(), 1.0)
    assert abs(loss.item() - want) <= 2e-5 * abs(want) + 1e-7
    loss.backward()
    np.testing.assert_allclose(tT.grad.cpu().numpy(), wgrad, rtol=1e-3, atol=2e-4 * np.abs(wgrad).max())
    ident = torch.eye(C, device=DEV)
    il = Idenyity_loss()(T(insT), ident).item()
    assert abs(il - np_ntm.identity_loss(insT, np.eye(C))) <= 1e-5 * abs(il)


def test_cal_mean_feature_matches_oracle():
    from geot_amd.ntm import cal_mean_feature
    rng = np.random.default_rng(6)
    batches = [(rng.normal(size=(2, C, 500)).astype(np.float32), rng.integers(0, 9, (2, 500))) for _ in range(3)]
    got = cal_mean_feature(((T(l), T(t, torch.int64)) for l, t in batches), C).cpu().numpy()
    np.testing.assert_allclose(got, np_ntm.cal_mean_feature(batches, C), rtol=1e-5, atol=1e-7)


def test_ntm_step_composition():
    """The unlabelled half of one FixMatch+NTM step (train.py:505-571) end to end: shapes, finiteness,
    gradients reach the T-predictor, sigma and the student logits."""
    from geot_amd.ntm import Ins_T_mean, class_transition, correct_logits, threeD_space_loss
    torch.manual_seed(0)
    B, N = 2, 4000
    xyz, _ = make_batch(B, N, start_index=3)
    pos = T(xyz)
    pred_weak = torch.randn(B, C, N, device=DEV)
    pred_strong = torch.randn(B, C, N, device=DEV, requires_grad=True)
    sigma = (0.5 + torch.rand(C, device=DEV)).requires_grad_(True)
    ema_t = torch.softmax(torch.randn(C, C, device=DEV), 1)
    cm = torch.softmax(torch.randn(C, C, device=DEV), 1)
    predictor = Ins_T_mean(nclasses=C).to(DEV)
    eta = torch.softmax(pred_weak, 1)
    _, label_u = torch.max(eta, 1)
    ema_corr, ema_next, _, _ = class_transition(eta, sigma, ema_t)
    insT = predictor(torch.softmax(pred_strong, 1).detach(), cm)
    corr = correct_logits(pred_strong, insT, ema_corr, 0.9)
    loss3d = threeD_space_loss(k=32, sigma=1.0)(pos, label_u, insT) * 0.1
    (corr.square().mean() + loss3d).backward()
    assert corr.shape == (B, C, N) and torch.isfinite(corr).all() and torch.isfinite(loss3d)
    assert all(l.weight.grad is not None and torch.isfinite(l.weight.grad).all() for l in predictor.T_predictor.fc)
    assert sigma.grad.abs().sum() > 0 and pred_strong.grad.abs().sum() > 0
    assert ema_next.shape == (C, C)


def test_ntm_workload_two_stream_schedule_gives_the_same_step(monkeypatch):
    """workloads.NtmHotPath builds the kNN graph / processing order on a second HIP stream and keeps the graph
    loss (forward and, through autograd's stream replay, backward) there: loss and every gradient must equal the
    single-stream schedule's (atomics reorder a few float adds: 1e-5)."""
    from geot_amd import workloads as wl
    from geot_amd.synth import make_logits
    xyz_np = make_batch(3, 6000, start_index=11)[0]
    xyz = T(xyz_np)
    pw, ps = T(make_logits(xyz_np, 0)), T(make_logits(xyz_np, 1, sharp=3.0))
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GEOT_NTM_OVERLAP", mode)
        monkeypatch.setenv("GEOT_NTM_OVERLAP_MIN", "1")
        torch.manual_seed(0)
        nt = wl.NtmHotPath().to(DEV)
        assert nt.overlap == (mode == "1")
        strong = ps.clone().requires_grad_(True)
        for _ in range(2):          # second step: ema_t has moved, streams are warm
            corr, loss3d = nt(xyz, pw, strong)
            loss = corr.square().mean() + loss3d
            nt.zero_grad(set_to_none=True)
            strong.grad = None
            loss.backward()
        torch.cuda.synchronize()
        res[mode] = (loss.item(), strong.grad.clone(), nt.sigma.grad.clone(),
                     torch.stack([l.weight.grad for l in nt.predictor.T_predictor.fc]).clone(), nt.ema_t.clone())
    a, b = res["0"], res["1"]
    assert abs(a[0] - b[0]) <= 1e-6 * abs(a[0])
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=1e-4, atol=1e-6 * float(x.abs().max()) + 1e-12)


def test_class_anchor_kernel_first_maximum_and_rows():
    """geot_ntm_class_anchors: per class the FIRST maximum over the flattened (b, n) order (exact ties planted within
    a cloud and across clouds), its value, and the whole soft-max row of that point."""
    from geot_amd.ext._common import call, ptr
    rng = np.random.default_rng(9)
    B, N = 3, 5000
    eta = _softmax(rng.standard_normal((B, C, N)), 1).astype(np.float32)
    for cls, spots in ((2, [(0, 4000), (0, 4999), (1, 3)]), (7, [(1, 77), (2, 0)]), (16, [(2, 4999)])):
        for bb, nn in spots:
            row = np.full(C, 0.0005, dtype=np.float32)
            row[cls] = 0.99
            row[(cls + 1 + nn) % C if (cls + 1 + nn) % C != cls else (cls + 2) % C] = 0.0021   # rows differ
            eta[bb, :, nn] = row
    te = T(eta)
    cT = torch.empty((C, C), device=DEV)
    vs = torch.empty(C, device=DEV)
    call("geot_ntm_class_anchors", te.device, B, N, C, ptr(te), ptr(cT), ptr(vs))
    flat = eta.transpose(1, 0, 2).reshape(C, B * N)
    first = flat.argmax(1)                                     # numpy: first maximum
    want = np.stack([eta[f // N, :, f % N] for f in first])
    assert np.array_equal(cT.cpu().numpy(), want) and np.array_equal(vs.cpu().numpy(), flat.max(1))
    assert first[2] == 0 * N + 4000 and first[7] == 1 * N + 77
    # and through the public function, against torch's own op chain
    from geot_amd.ntm import class_transition
    import os
    sig, ema = T(0.5 + rng.random(C)), T(_softmax(rng.standard_normal((C, C)), 1))
    a = class_transition(te, sig, ema)[2]
    os.environ["GEOT_NTM_CT"] = "torch"
    try:
        b_ = class_transition(te, sig, ema)[2]
    finally:
        del os.environ["GEOT_NTM_CT"]
    assert torch.equal(a, b_)


@pytest.mark.parametrize("Cn", [2, 5, 7, 8, 12, 13, 16, 20, 31, 32])   # 6 up: the MFMA form, every row-tile width, whole and partial pieces
def test_any_class_count_matches_the_oracle(Cn):
    """transformer.py:1104-1110 builds `nclasses` heads for ANY nclasses and insT_loss.py takes num_classes: every
    per-point kernel at class counts other than the specialised 17 (run-time-C kernels, csrc/ntm_generic.hip)."""
    from geot_amd import ntm
    rng = np.random.default_rng(Cn)
    B, N, k = 2, 700, 9
    p = _softmax(rng.standard_normal((B, Cn, N)) * 2, 1).astype(np.float32)
    cm = _softmax(rng.standard_normal((Cn, Cn)), 1).astype(np.float32)
    mod = ntm.Ins_T_mean(nclasses=Cn).to(DEV)
    with torch.no_grad():
        for kk, l in enumerate(mod.T_predictor.fc):
            l.weight[kk, kk] += 0.6
            l.weight += 0.03
    W = torch.stack([l.weight for l in mod.T_predictor.fc]).detach().cpu().numpy()
    insT = mod(T(p), T(cm))
    want = np_ntm.sig_t_mean(p, cm, W)
    np.testing.assert_allclose(insT.detach().cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    g = rng.standard_normal(want.shape).astype(np.float32)
    (insT * T(g)).sum().backward()
    got = torch.stack([l.weight.grad for l in mod.T_predictor.fc]).cpu().numpy()
    ref = np_ntm.sig_t_mean_grad_W(p, cm, W, g)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())
    # logit correction, forward + the three gradients
    logits = (rng.standard_normal((B, Cn, N)) * 2).astype(np.float32)
    E = np_ntm.l1_normalize(rng.random((Cn, Cn)) + 0.01, 1).astype(np.float32)
    tl, ti, tE = T(logits).requires_grad_(True), T(want).requires_grad_(True), T(E).requires_grad_(True)
    out = ntm.correct_logits(tl, ti, tE, 0.9)
    _, corr = np_ntm.correct_logits(logits, want.astype(np.float32), E, 0.9)
    np.testing.assert_allclose(out.detach().cpu().numpy(), corr, rtol=1e-5, atol=1e-5 * np.abs(corr).max())
    go = rng.standard_normal(corr.shape).astype(np.float32)
    (out * T(go)).sum().backward()
    gl, gi, gE = np_ntm.correct_logits_grads(logits, want.astype(np.float32), E, 0.9, go)
    np.testing.assert_allclose(tl.grad.cpu().numpy(), gl, rtol=1e-5, atol=1e-5 * np.abs(gl).max())
    np.testing.assert_allclose(ti.grad.cpu().numpy(), gi, rtol=1e-5, atol=1e-5 * np.abs(gi).max())
    np.testing.assert_allclose(tE.grad.cpu().numpy(), gE, rtol=1e-4, atol=1e-4 * np.abs(gE).max())
    # graph losses (neighbours from the exact kNN)
    xyz, _ = make_batch(B, N, start_index=3, origin_pts=0)
    labels = rng.integers(0, min(Cn, 3), (B, N))
    loss_mod = ntm.threeD_space_loss(k=k, sigma=1.0, num_classes=Cn)
    nbr = loss_mod.neighbours(T(xyz))
    for env in ("graph", "atomic"):       # "graph" falls to the scatter form at class counts without the graph kernels
        ti = T(want).requires_grad_(True)
        import os
        os.environ["GEOT_NTM_GRAD"] = env
        try:
            loss = loss_mod(T(xyz), T(labels, torch.int64), ti, nbr=nbr)
            loss.backward()
        finally:
            del os.environ["GEOT_NTM_GRAD"]
        wl, wg, _ = np_ntm.threed_space_loss(xyz, labels, want, nbr.cpu().numpy(), sigma=1.0)
        assert abs(loss.item() - wl) <= 1e-5 * abs(wl)
        np.testing.assert_allclose(ti.grad.cpu().numpy(), wg, rtol=1e-4, atol=1e-4 * np.abs(wg).max())
    ti = T(want).requires_grad_(True)
    fl = ntm.feature_space_loss(k=k, sigma=1.0, num_classes=Cn)(T(p), T(labels, torch.int64), ti)
    fl.backward()
    assert torch.isfinite(fl) and torch.isfinite(ti.grad).all()
    from geot_amd.openpoints.models.layers.knn import knn_point
    feats = T(p).permute(0, 2, 1).contiguous()
    fnbr = knn_point(k + 1, feats, feats)[1][:, :, 1:].cpu().numpy()
    wl, wg, _ = np_ntm.feature_space_loss(p, labels, want, fnbr, sigma=1.0)
    assert abs(fl.item() - wl) <= 1e-5 * max(abs(wl), 1e-3)
    np.testing.assert_allclose(ti.grad.cpu().numpy(), wg, rtol=1e-4, atol=1e-4 * np.abs(wg).max())


@pytest.mark.parametrize("Cn", [8, 13, 20, 32])
@pytest.mark.parametrize("impl", ["mfma", "rows"])
def test_any_class_count_at_full_size_against_fp64(Cn, impl, monkeypatch):
    """140 002 points: more point tiles than resident waves (every wave walks several tiles, the next tile's inputs in flight),
    a last tile of 2 points, C = 32 with its weights split over two workgroup ranges -- forward and d raw of the run-time-C
    sig_t_mean against the same arithmetic in fp64 (transformer.py:1111-1131: heads, clamp, L1 norm over o)."""
    from geot_amd import ntm
    monkeypatch.setenv("GEOT_NTM_GENERIC", impl)
    torch.manual_seed(Cn)
    B, N = 2, 70001
    p = torch.softmax(torch.randn(B, Cn, N, device=DEV) * 2, 1)
    cm = torch.softmax(torch.randn(Cn, Cn, device=DEV), 1)
    mod = ntm.Ins_T_mean(nclasses=Cn).to(DEV)
    with torch.no_grad():
        for kk, l in enumerate(mod.T_predictor.fc):
            l.weight[kk, kk] += 0.6
            l.weight += 0.03
    W = torch.stack([l.weight for l in mod.T_predictor.fc]).detach().contiguous()
    g = torch.randn(B * N, Cn, Cn, device=DEV)
    with torch.no_grad():
        got = mod(p, cm)
        draw = ntm.sig_t_mean_grad_raw(p, cm, W, g)
    W64, p64 = W.double(), p.double()
    raw = torch.einsum("bjn,koj->bnko", p64, W64[:, :, :Cn]).reshape(B * N, Cn, Cn) + torch.einsum("kj,koj->ko", cm.double(), W64[:, :, Cn:])
    raw.requires_grad_(True)
    cl = raw.clamp(1e-5, 1 - 1e-5)
    want = cl / cl.abs().sum(-1, keepdim=True).clamp_min(1e-12)
    assert float((got.double() - want.detach()).abs().max()) <= 2e-6
    (want * g.double()).sum().backward()
    # an element whose fp32 raw value rounds across the clamp's edge flips its gradient on or off: compare where fp64 is clear of it
    clear = ((raw.detach() - 1e-5).abs() > 1e-6) & ((raw.detach() - (1 - 1e-5)).abs() > 1e-6)
    err = ((draw.double() - raw.grad).abs() * clear).max()
    assert float(err) <= 1e-5 * float(raw.grad.abs().max())


def test_class_count_out_of_range_is_an_error():
    from geot_amd import ntm
    p = torch.softmax(torch.randn(1, 33, 50, device=DEV), 1)
    with pytest.raises(RuntimeError, match="1..32 classes"):
        ntm.sig_t_mean(33).to(DEV)(p, torch.eye(33, device=DEV))
