"""CPU suite, part 3: the fp64 restatement of the NTM block (oracle/np_ntm.py) checked against
torch autograd on a straightforward fp64 torch transcription of the same formulas and against
finite differences -- the reference code itself hard-codes .cuda() and cannot run here."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import np_ntm

C = 17


def _softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def test_sig_t_mean_matches_torch_transcription_and_autograd():
    rng = np.random.default_rng(0)
    B, N = 2, 50
    p = _softmax(rng.standard_normal((B, C, N)) * 2, 1)
    cm = _softmax(rng.standard_normal((C, C)), 1)
    W = rng.standard_normal((C, C, 2 * C)) * 0.3
    out = np_ntm.sig_t_mean(p, cm, W)
    assert out.shape == (B * N, C, C)
    assert np.allclose(out.sum(2), 1.0)
    tp, tcm = torch.tensor(p), torch.tensor(cm)
    tW = torch.tensor(W, requires_grad=True)
    o = tp.permute(0, 2, 1).reshape(-1, C)
    rows = [torch.cat((o, tcm[kk].unsqueeze(0).repeat(o.shape[0], 1)), 1) @ tW[kk].T for kk in range(C)]
    T = F.normalize(torch.clamp(torch.stack(rows, 1), min=1e-5, max=1 - 1e-5), p=1, dim=2)
    assert np.allclose(T.detach().numpy(), out, atol=1e-12)
    g = rng.standard_normal(out.shape)
    (T * torch.tensor(g)).sum().backward()
    assert np.allclose(tW.grad.numpy(), np_ntm.sig_t_mean_grad_W(p, cm, W, g), atol=1e-9)


def test_class_transition_quirks():
    rng = np.random.default_rng(1)
    B, N = 2, 300
    eta = _softmax(rng.standard_normal((B, C, N)) * 3, 1)
    sigma = 0.5 + rng.random(C)
    ema = _softmax(rng.standard_normal((C, C)), 1)
    r = np_ntm.class_transition(eta, sigma, ema)
    # anchor rows are copies of the most confident point's full distribution
    for cc in range(C):
        flat = eta[:, cc, :].reshape(-1)
        i = int(flat.argmax())
        assert np.array_equal(r["class_T"][cc], eta[i // N, :, i % N])
    assert np.all(r["prior_T"][1:, 0] == 0)
    # the reference's X / X.sum(1) divides COLUMN j by ROW-sum j (broadcast along the last axis)
    raw = r["ema_t_next"] * (ema * 0.999 + r["class_T"] * 0.001).sum(1)[None, :]
    assert np.allclose(raw, ema * 0.999 + r["class_T"] * 0.001)
    # torch transcription of train.py:505-545 agrees
    te, ts, tm = torch.tensor(eta), torch.tensor(sigma), torch.tensor(ema)
    class_T = torch.empty(C, C, dtype=torch.float64)
    prior = torch.zeros(C, C, dtype=torch.float64)
    for cc in range(C):
        flat = te[:, cc, :].contiguous().view(B * N)
        ib = torch.argmax(flat)
        class_T[cc] = te[ib // N, :, ib % N]
        if cc == 0:
            continue
        for k in range(C):
            x, mu = np_ntm.LABEL_PROJ[k], np_ntm.LABEL_PROJ[cc]
            prior[cc, k] = (1 / (ts[cc] * torch.sqrt(torch.tensor(2 * torch.pi)))) * torch.exp(-((x - mu) ** 2) / (2 * ts[cc] ** 2))
    prior[:, 0] = 0
    prior[0, 0] = 1
    prior = prior / torch.sum(prior, 1)
    new_T = 0.999 * class_T + 0.001 * prior
    new_T[0] = class_T[0]
    new_T = new_T / torch.sum(new_T, 1)
    corr = tm * 0.999 + new_T * 0.001
    corr = corr / torch.sum(corr, 1)
    assert np.allclose(corr.numpy(), r["ema_t_corr"], atol=1e-7)


def test_correct_logits_and_grads():
    rng = np.random.default_rng(2)
    B, N = 2, 40
    logits = rng.standard_normal((B, C, N)) * 2
    insT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2)
    E = np_ntm.l1_normalize(rng.random((C, C)) + 0.01, 1)
    newT, corr = np_ntm.correct_logits(logits, insT, E, 0.9)
    tl = torch.tensor(logits, requires_grad=True)
    ti = torch.tensor(insT, requires_grad=True)
    tE = torch.tensor(E, requires_grad=True)
    nT = F.normalize(0.9 * tE + 0.1 * ti, p=1, dim=2)
    pc = torch.bmm(tl.permute(0, 2, 1).contiguous().view(-1, C).unsqueeze(1), nT).squeeze(1)
    pc = pc.view(B, N, C).permute(0, 2, 1).contiguous()
    assert np.allclose(pc.detach().numpy(), corr, atol=1e-12) and np.allclose(nT.detach().numpy(), newT)
    g = rng.standard_normal(corr.shape)
    (pc * torch.tensor(g)).sum().backward()
    gl, gi, gE = np_ntm.correct_logits_grads(logits, insT, E, 0.9, g)
    assert np.allclose(tl.grad.numpy(), gl, atol=1e-10)
    assert np.allclose(ti.grad.numpy(), gi, atol=1e-10)
    assert np.allclose(tE.grad.numpy(), gE, atol=1e-9)


def test_threed_space_loss_matches_torch_transcription():
    from oracle import capi
    from geot_amd.synth import make_batch
    rng = np.random.default_rng(3)
    B, N, k = 2, 120, 7
    xyz, _ = make_batch(B, N, start_index=40, origin_pts=0)
    labels = rng.integers(0, 3, (B, N))
    insT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2)
    idx, _ = capi.knn_sorted(xyz, xyz, k + 1)
    nbr = idx[:, :, 1:]
    loss, grad, _ = np_ntm.threed_space_loss(xyz, labels, insT, nbr, sigma=1.0)
    # torch transcription of utils/insT_loss.py:68-110 (index_select/cat loop, detached weights)
    pos = torch.tensor(xyz, dtype=torch.float64)
    tl = torch.tensor(labels)
    tT = torch.tensor(insT, requires_grad=True)
    top = torch.tensor(nbr.astype(np.int64))
    factor = torch.arange(B).unsqueeze(-1).repeat(1, N)
    P = pos.view(B * N, -1)
    L = tl.view(-1)
    nP, nL, nT = [], [], []
    for i in range(k):
        cur = (top[:, :, i] + factor * N).view(-1)
        nP.append(torch.index_select(P, 0, cur).unsqueeze(1))
        nL.append(torch.index_select(L, 0, cur).unsqueeze(1))
        nT.append(torch.index_select(tT, 0, cur).unsqueeze(1))
    nP, nL, nT = torch.cat(nP, 1), torch.cat(nL, 1), torch.cat(nT, 1)
    dm = torch.zeros(B * N, k, dtype=torch.float64)
    dm[L.unsqueeze(1).repeat(1, k) == nL] = 1
    eij = torch.exp(-torch.sum((P.unsqueeze(1).repeat(1, k, 1) - nP) ** 2, dim=2) / 2.0)
    dm = dm * eij
    vT = tT.unsqueeze(1).repeat(1, k, 1, 1).view(B * N, k, -1)
    td = torch.sum((vT - nT.view(B * N, k, -1)) ** 2, dim=2)
    ml = (torch.sum(dm.detach() * td, dim=1) / (torch.sum(dm.detach(), dim=1) + 0.001)).mean()
    assert np.isclose(ml.item(), loss, rtol=1e-12)
    ml.backward()
    assert np.allclose(tT.grad.numpy(), grad, atol=1e-12)


def test_feature_space_and_identity_loss_match_torch_transcription():
    rng = np.random.default_rng(5)
    B, N, k = 2, 90, 7
    logits = _softmax(rng.normal(size=(B, C, N)), 1)
    labels = rng.integers(0, 3, (B, N))
    insT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2)
    # torch transcription of utils/insT_loss.py:16-58 (knn_point = cdist + topk, index_select/cat loop)
    lg = torch.tensor(logits).permute(0, 2, 1).contiguous()
    top = torch.cdist(lg, lg).topk(k=k + 1, dim=-1, largest=False, sorted=True).indices[:, :, 1:]
    loss, grad, _ = np_ntm.feature_space_loss(logits, labels, insT, top.numpy(), sigma=1.0)
    tT = torch.tensor(insT, requires_grad=True)
    factor = torch.arange(B).unsqueeze(-1).repeat(1, N)
    P, L = lg.view(B * N, -1), torch.tensor(labels).view(-1)
    nP, nL, nT = [], [], []
    for i in range(k):
        cur = (top[:, :, i] + factor * N).view(-1)
        nP.append(torch.index_select(P, 0, cur).unsqueeze(1))
        nL.append(torch.index_select(L, 0, cur).unsqueeze(1))
        nT.append(torch.index_select(tT, 0, cur).unsqueeze(1))
    nP, nL, nT = torch.cat(nP, 1), torch.cat(nL, 1), torch.cat(nT, 1)
    dm = -torch.ones(B * N, k, dtype=torch.float64)
    dm[L.unsqueeze(1).repeat(1, k) == nL] = 1
    dm = dm * torch.exp(-torch.sum((P.unsqueeze(1).repeat(1, k, 1) - nP) ** 2, dim=2) / 2.0)
    td = torch.sum((tT.unsqueeze(1).repeat(1, k, 1, 1).view(B * N, k, -1) - nT.view(B * N, k, -1)) ** 2, dim=2)
    ml = torch.mean(dm.detach() * td)
    assert np.isclose(ml.item(), loss, rtol=1e-12)
    ml.backward()
    assert np.allclose(tT.grad.numpy(), grad, atol=1e-12)
    # Idenyity_loss, insT_loss.py:122-132
    ident = torch.eye(C, dtype=torch.float64)
    t2 = torch.tensor(insT)
    num = t2.size(0)
    I = ident.repeat(num, 1, 1).view(num, -1)
    want = (torch.sum((t2.view(num, -1) - I).pow(2) * I, dim=1) / torch.sum(I, dim=1)).mean().item()
    assert np.isclose(np_ntm.identity_loss(insT, np.eye(C)), want, rtol=1e-12)


def test_cal_mean_feature_quirk():
    rng = np.random.default_rng(6)
    batches = [(rng.normal(size=(2, C, 50)), rng.integers(0, 5, (2, 50))) for _ in range(3)]
    cm = np_ntm.cal_mean_feature(batches, C)
    assert cm.shape == (C, C) and np.all(cm[5:] == 0)          # unseen classes stay zero
    # within one batch every visited class receives the SAME mean (logits[target] indexes rows by label
    # value, not by a class mask); across batches the rows differ only through the count weighting
    one = np_ntm.cal_mean_feature(batches[:1], C)
    assert np.allclose(one[0], one[1]) and np.allclose(one[0], one[4]) and one[0].sum() > 0
    assert np.allclose(cm[:5].sum(1), 1.0, atol=1e-6)
