"""Randomised check of the NTM kernels against the fp64 restatement (oracle/np_ntm.py): random batch / point
counts (not multiples of any tile), k from 1 to 48, label patterns from "all equal" to "all different", clouds
with duplicate points; the three backward forms of the graph loss.

    python tools/ntm_fuzz.py [--cases 30] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.ntm import Ins_T_mean, correct_logits, threeD_space_loss  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from oracle import capi, np_ntm  # noqa: E402  (checker)

DEV, C = "cuda:0", 17


def T(a, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(DEV)


def softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def check(name, got, want, rtol, atol, info):
    if not np.allclose(got, want, rtol=rtol, atol=atol):
        err = np.abs(got - want).max()
        print("MISMATCH %s max abs err %.3g (atol %.3g) %s" % (name, err, atol, info))
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    for case in range(a.cases):
        rng = np.random.default_rng(104729 * a.seed + case)
        torch.manual_seed(case)
        B = int(rng.integers(1, 4))
        N = int(rng.choice([49, 50, 127, 300, 1001, 2048, 2500]))
        k = int(min(N - 1, rng.choice([1, 2, 7, 31, 32, 33, 48])))
        nlab = int(rng.choice([1, 2, 5, 17]))
        info = "case %d seed %d: B %d N %d k %d nlab %d" % (case, a.seed, B, N, k, nlab)
        xyz, _ = make_batch(B, N, start_index=1000 * a.seed + case, origin_pts=0, dup_frac=float(rng.choice([0.0, 0.05])))
        labels = rng.integers(0, nlab, (B, N))
        p = softmax(rng.standard_normal((B, C, N)) * 2, 1).astype(np.float32)
        cm = softmax(rng.standard_normal((C, C)), 1).astype(np.float32)
        # ---- per-point matrices
        mod = Ins_T_mean(nclasses=C).to(DEV)
        W = torch.stack([l.weight for l in mod.T_predictor.fc]).detach().cpu().numpy()
        insT = mod(T(p), T(cm))
        check("sig_t_mean", insT.detach().cpu().numpy(), np_ntm.sig_t_mean(p, cm, W), 1e-5, 2e-5, info)
        g = rng.standard_normal((B * N, C, C)).astype(np.float32)
        (insT * T(g)).sum().backward()
        got = torch.stack([l.weight.grad for l in mod.T_predictor.fc]).cpu().numpy()
        ref = np_ntm.sig_t_mean_grad_W(p, cm, W, g)
        check("sig_t_mean grad_W", got, ref, 2e-4, 2e-4 * np.abs(ref).max(), info)
        # ---- correction
        logits = (rng.standard_normal((B, C, N)) * 2).astype(np.float32)
        iT = np_ntm.l1_normalize(rng.random((B * N, C, C)) + 0.01, 2).astype(np.float32)
        E = np_ntm.l1_normalize(rng.random((C, C)) + 0.01, 1).astype(np.float32)
        tl, ti, tE = T(logits).requires_grad_(True), T(iT).requires_grad_(True), T(E).requires_grad_(True)
        out = correct_logits(tl, ti, tE, 0.9)
        check("correct", out.detach().cpu().numpy(), np_ntm.correct_logits(logits, iT, E, 0.9)[1], 1e-5, 2e-5, info)
        go = rng.standard_normal(out.shape).astype(np.float32)
        (out * T(go)).sum().backward()
        gl, gi, gE = np_ntm.correct_logits_grads(logits, iT, E, 0.9, go)
        check("correct d logits", tl.grad.cpu().numpy(), gl, 1e-4, 1e-5, info)
        check("correct d ins_T", ti.grad.cpu().numpy(), gi, 1e-4, 1e-4 * np.abs(gi).max(), info)
        check("correct d ema_t", tE.grad.cpu().numpy(), gE, 1e-3, 1e-3 * np.abs(gE).max(), info)
        # ---- graph loss, three backward forms
        crit = threeD_space_loss(k=k, sigma=float(rng.choice([0.5, 1.0])), num_classes=C)
        pos = T(xyz)
        nbr = crit.neighbours(pos)
        widx, _ = capi.knn_sorted(xyz, xyz, k + 1)
        if not np.array_equal(nbr.cpu().numpy(), widx[:, :, 1:]):
            print("MISMATCH kNN graph", info)
            sys.exit(1)
        want, wgrad, _ = np_ntm.threed_space_loss(xyz, labels, iT, widx[:, :, 1:], crit.sigma)
        for mode in ("graph", "gather", "atomic"):
            os.environ["GEOT_NTM_GRAD"] = mode
            t2 = T(iT).requires_grad_(True)
            l2 = crit(pos, T(labels, torch.int64), t2)
            l2.backward()
            if abs(l2.item() - want) > 2e-5 * abs(want) + 1e-9:
                print("MISMATCH loss3d/%s %g vs %g %s" % (mode, l2.item(), want, info))
                sys.exit(1)
            check("loss3d grad/" + mode, t2.grad.cpu().numpy(), wgrad, 1e-3, 2e-4 * np.abs(wgrad).max() + 1e-12, info)
        os.environ.pop("GEOT_NTM_GRAD", None)
        if case % 10 == 9:
            print("case %d ok" % case, flush=True)
    print("ntm_fuzz: %d cases within tolerance of the fp64 restatement" % a.cases)


if __name__ == "__main__":
    main()
