"""fp64 numpy restatement of the NTM (noise-transition-matrix) block -- TEST INFRASTRUCTURE ONLY.

Follows, line by line:
  * sig_t_mean.forward            openpoints/models/backbone/transformer.py:1120-1131
  * class-T / prior / EMA block   examples/segmentation/train.py:505-545, 556-557, 835-836 (+ LABEL_PROJ :48)
  * logit correction              examples/segmentation/train.py:547-552
  * threeD_space_loss.forward     utils/insT_loss.py:68-110
Pinned (round 3): tests/golden/make_ntm_golden.py executes the reference's own class bodies / statement ranges
in place on the CPU (Tensor.cuda made the identity) in fp32 and fp64 and stores inputs, outputs and autograd
gradients; tests/test_ref_fixtures_cpu.py holds every function below to the fp64 runs at 1e-13.
tests/test_ntm_cpu.py additionally checks them against autograd / finite differences of literal transcriptions.
"""
import numpy as np

LABEL_PROJ = [0, 8, 7, 6, 5, 4, 3, 2, 1, 9, 10, 11, 12, 13, 14, 15, 16]  # train.py:48


def l1_normalize(x, axis, eps=1e-12):
    """torch.nn.functional.normalize(p=1): x / max(sum|x|, eps)."""
    s = np.abs(x).sum(axis=axis, keepdims=True)
    return x / np.maximum(s, eps)


def sig_t_mean(p, cm, W):
    """p (B,C,N) softmax, cm (C,C), W (C, C, 2C) [kk][out][in] -> ins_T (B*N, C, C)."""
    p = np.asarray(p, dtype=np.float64)
    cm = np.asarray(cm, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    B, C, N = p.shape
    out = p.transpose(0, 2, 1).reshape(-1, C)                       # (BN, C)
    T = np.empty((B * N, C, C))
    for kk in range(C):
        new_in = np.concatenate([out, np.repeat(cm[kk][None], B * N, 0)], 1)   # (BN, 2C)
        T[:, kk, :] = new_in @ W[kk].T
    T = np.clip(T, 1e-5, 1 - 1e-5)
    return l1_normalize(T, 2)


def sig_t_mean_grad_W(p, cm, W, grad_out):
    """d<ins_T, grad_out>/dW (inputs are detached in the reference: only W gets a gradient)."""
    p = np.asarray(p, dtype=np.float64)
    cm = np.asarray(cm, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    g = np.asarray(grad_out, dtype=np.float64)
    B, C, N = p.shape
    out = p.transpose(0, 2, 1).reshape(-1, C)
    gW = np.zeros_like(W)
    for kk in range(C):
        new_in = np.concatenate([out, np.repeat(cm[kk][None], B * N, 0)], 1)
        raw = new_in @ W[kk].T                                       # (BN, C)
        tc = np.clip(raw, 1e-5, 1 - 1e-5)
        s = np.abs(tc).sum(1, keepdims=True)
        den = np.maximum(s, 1e-12)
        tn = tc / den
        gk = g[:, kk, :]
        dtc = np.where(s > 1e-12, (gk - np.sign(tc) * (gk * tn).sum(1, keepdims=True)) / den, gk / den)
        draw = dtc * ((raw >= 1e-5) & (raw <= 1 - 1e-5))
        gW[kk] = draw.T @ new_in
    return gW


def gaussian(x, mu, s):
    return (1.0 / (s * np.sqrt(2 * np.pi))) * np.exp(-((x - mu) ** 2) / (2 * s ** 2))


def class_transition(eta, sigma, ema_t, geo_lambda=0.999, ema_decay=0.999, filter_outlier=False, outlier_q=0.97):
    """eta (B_u, C, N) softmax of the weak view, sigma (C,), ema_t (C,C).
    filter_outlier (train.py:510-517): class cc's probabilities at or above their q-quantile are zeroed IN PLACE in
    the working copy before its arg-max, so the anchor rows read afterwards see classes 0..cc already filtered.
    Returns dict(class_T, prior_T, new_T, ema_t_corr, ema_t_next).  `X / X.sum(1)` broadcasts the
    row sums along the LAST axis (column j divided by row-sum j): reference quirk, reproduced."""
    eta = np.array(eta, dtype=np.float64)        # a copy: the filter writes into it, as the reference's clone() (:507)
    sigma = np.asarray(sigma, dtype=np.float64)
    ema_t = np.asarray(ema_t, dtype=np.float64)
    B, C, N = eta.shape
    class_T = np.empty((C, C))
    prior_T = np.zeros((C, C))
    for cc in range(C):
        if filter_outlier:
            view = eta[:, cc, :]
            view[view >= np.quantile(view, outlier_q)] = 0.0          # torch.quantile: linear interpolation, as numpy
        flat = eta[:, cc, :].reshape(B * N)
        best = int(np.argmax(flat))                                   # first maximum
        class_T[cc] = eta[best // N, :, best % N]
        if cc == 0:
            continue
        for k in range(C):
            prior_T[cc, k] = gaussian(LABEL_PROJ[k], LABEL_PROJ[cc], sigma[cc])
    prior_T[:, 0] = 0
    prior_T[0, 0] = 1
    prior_T = prior_T / prior_T.sum(1)
    new_T = geo_lambda * class_T + (1 - geo_lambda) * prior_T
    new_T[0] = class_T[0]
    new_T = new_T / new_T.sum(1)
    ema_t_corr = ema_t * ema_decay + new_T * (1 - ema_decay)
    ema_t_corr = ema_t_corr / ema_t_corr.sum(1)
    ema_next = ema_t * ema_decay + class_T * (1 - ema_decay)
    ema_next = ema_next / ema_next.sum(1)
    return dict(class_T=class_T, prior_T=prior_T, new_T=new_T, ema_t_corr=ema_t_corr, ema_t_next=ema_next)


def correct_logits(logits, ins_T, ema_t_corr, lam):
    """logits (B,C,N) raw strong-view logits, ins_T (BN,C,C) -> (newT (BN,C,C), pred_corr (B,C,N))."""
    logits = np.asarray(logits, dtype=np.float64)
    ins_T = np.asarray(ins_T, dtype=np.float64)
    B, C, N = logits.shape
    newT = l1_normalize(lam * np.asarray(ema_t_corr, dtype=np.float64)[None] + (1 - lam) * ins_T, 2)
    rows = logits.transpose(0, 2, 1).reshape(-1, 1, C)               # (BN,1,C)
    corr = np.matmul(rows, newT)[:, 0, :]                             # (BN,C)
    return newT, corr.reshape(B, N, C).transpose(0, 2, 1)


def correct_logits_grads(logits, ins_T, ema_t_corr, lam, grad_out):
    """Gradients of <pred_corr, grad_out> w.r.t. logits, ins_T and ema_t_corr."""
    logits = np.asarray(logits, dtype=np.float64)
    ins_T = np.asarray(ins_T, dtype=np.float64)
    E = np.asarray(ema_t_corr, dtype=np.float64)
    g = np.asarray(grad_out, dtype=np.float64)
    B, C, N = logits.shape
    v = lam * E[None] + (1 - lam) * ins_T                             # (BN,C,C)
    s = np.abs(v).sum(2, keepdims=True)
    den = np.maximum(s, 1e-12)
    tn = v / den
    l = logits.transpose(0, 2, 1).reshape(-1, C)                      # (BN,C) rows r
    go = g.transpose(0, 2, 1).reshape(-1, C)                          # (BN,C) cols c
    g_logits = np.einsum("irc,ic->ir", tn, go)
    dtn = l[:, :, None] * go[:, None, :]                              # (BN,C,C)
    dv = np.where(s > 1e-12, (dtn - np.sign(v) * (dtn * tn).sum(2, keepdims=True)) / den, dtn / den)
    return g_logits.reshape(B, N, C).transpose(0, 2, 1), (1 - lam) * dv, lam * dv.sum(0)


def threed_space_loss(positions, labels, ins_T, nbr, sigma=1.0):
    """positions (B,N,3), labels (B,N), ins_T (BN,C,C), nbr (B,N,k) local neighbour ids (self excluded)
    -> (loss, grad wrt ins_T).  w_ij is detached in the reference; the gradient flows to T_i and T_j."""
    pos = np.asarray(positions, dtype=np.float64)
    lab = np.asarray(labels)
    B, N, _ = pos.shape
    k = nbr.shape[2]
    T = np.asarray(ins_T, dtype=np.float64).reshape(B * N, -1)
    gidx = (np.asarray(nbr, dtype=np.int64) + (np.arange(B)[:, None, None] * N)).reshape(B * N, k)
    P = pos.reshape(B * N, 3)
    Lb = lab.reshape(B * N)
    same = (Lb[:, None] == Lb[gidx]).astype(np.float64)
    d2 = ((P[:, None, :] - P[gidx]) ** 2).sum(2)
    w = same * np.exp(-d2 / (2 * sigma ** 2))                         # (BN,k)
    diff = T[:, None, :] - T[gidx]                                    # (BN,k,C2)
    tdist = (diff ** 2).sum(2)
    S = w.sum(1) + 0.001
    per_point = (w * tdist).sum(1) / S
    loss = per_point.mean()
    coef = (2.0 / (B * N)) * (w / S[:, None])                         # (BN,k)
    contrib = coef[:, :, None] * diff                                 # d loss / d T_i (and minus for T_j)
    grad = contrib.sum(1)
    np.subtract.at(grad, gidx.reshape(-1), contrib.reshape(-1, T.shape[1]))
    return loss, grad.reshape(np.asarray(ins_T).shape), per_point


def feature_space_loss(logits, labels, ins_T, nbr, sigma=1.0):
    """utils/insT_loss.py:16-58: logits (B,C,N), labels (B,N), ins_T (BN,C,C), nbr (B,N,k) neighbour ids in
    logit space (self excluded) -> (loss, grad wrt ins_T, per_point sums).  distance_map = -1 / +1 by label
    equality (:44-46), times exp(-|l_i-l_j|^2/(2 sigma^2)), detached; loss = mean over (BN, k)."""
    F = np.asarray(logits, dtype=np.float64).transpose(0, 2, 1)
    B, N, D = F.shape
    k = nbr.shape[2]
    T = np.asarray(ins_T, dtype=np.float64).reshape(B * N, -1)
    gidx = (np.asarray(nbr, dtype=np.int64) + (np.arange(B)[:, None, None] * N)).reshape(B * N, k)
    P = F.reshape(B * N, D)
    Lb = np.asarray(labels).reshape(B * N)
    sign = np.where(Lb[:, None] == Lb[gidx], 1.0, -1.0)
    w = sign * np.exp(-((P[:, None, :] - P[gidx]) ** 2).sum(2) / (2 * sigma ** 2))
    diff = T[:, None, :] - T[gidx]
    per_point = (w * (diff ** 2).sum(2)).sum(1)
    loss = per_point.sum() / (B * N * k)
    contrib = (2.0 / (B * N * k)) * w[:, :, None] * diff
    grad = contrib.sum(1)
    np.subtract.at(grad, gidx.reshape(-1), contrib.reshape(-1, T.shape[1]))
    return loss, grad.reshape(np.asarray(ins_T).shape), per_point


def identity_loss(ins_T, identity):
    """utils/insT_loss.py:122-132 (class Idenyity_loss)."""
    T = np.asarray(ins_T, dtype=np.float64)
    num = T.shape[0]
    I = np.broadcast_to(np.asarray(identity, dtype=np.float64), T.shape).reshape(num, -1)
    diff = (T.reshape(num, -1) - I) ** 2
    return ((diff * I).sum(1) / I.sum(1)).mean()


def cal_mean_feature(batches, c=17):
    """train.py:868-897, literal (model call factored out: batches of (raw logits (B,C,N), target (B,N)))."""
    cm = np.zeros((c, c))
    c_num = np.zeros(c)
    for logits, target in batches:
        z = np.asarray(logits, dtype=np.float64)
        e = np.exp(z - z.max(1, keepdims=True))
        sm = (e / e.sum(1, keepdims=True)).transpose(0, 2, 1).reshape(-1, c)
        t = np.asarray(target).reshape(-1)
        for kk in range(c):
            cur_num = (t == kk).sum()
            if cur_num == 0:
                continue
            mean_feats = sm[t].mean(0)                     # :891  logits[target] -- indexed by label VALUE
            cm[kk] = (cm[kk] * c_num[kk] + mean_feats * cur_num) / (c_num[kk] + cur_num)
            c_num[kk] += cur_num
    return cm.astype(np.float32)
