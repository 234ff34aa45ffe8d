"""Accuracy of grad_ema_t (the (C,C) reduction over points in the logit correction's backward) against an fp64 torch composition."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from geot_amd import ntm
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "ntm_ref_transition.npz"))
dev = "cuda:0"
for tag in ("c17_plain_",):
    T = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    strong, insT, G = T(g[tag + "strong"]), T(g[tag + "insT_f32"]), T(g[tag + "G"])
    E = T(g[tag + "ema_t_corr_f32"]).requires_grad_(True)
    out = ntm.correct_logits(strong, insT, E, 0.9)
    (out * G).sum().backward()
    ge = E.grad.double()
    for dt in (torch.float32, torch.float64):
        s, t, e, gg = strong.to(dt), insT.to(dt), E.detach().to(dt).requires_grad_(True), G.to(dt)
        b, c, n = s.shape
        newT = torch.nn.functional.normalize(0.9 * e[None] + 0.1 * t, p=1, dim=-1)
        pred = torch.bmm(s.permute(0, 2, 1).reshape(-1, 1, c), newT)[:, 0, :].reshape(b, n, c).permute(0, 2, 1)
        (pred * gg).sum().backward()
        ref = e.grad.double()
        if dt == torch.float64:
            ref64 = ref
        else:
            ref32 = ref
    sc = ref64.abs().max(1, keepdim=True)[0]
    print("grad_ema_t: ours vs fp64 %.3g (row-relative), torch fp32 composition vs fp64 %.3g; |ge| max %.3g" %
          (((ge - ref64).abs() / sc).max(), ((ref32 - ref64).abs() / sc).max(), ref64.abs().max()))
    # d sigma from the class-transition backward, given (a) our grad_ema_t, (b) the fp64 composition's (rounded to fp32)
    sigma0 = T(g[tag + "sigma"])
    for name, gcorr in (("ours", E.grad.float()), ("fp64->fp32", ref64.float())):
        sig = sigma0.clone().requires_grad_(True)
        corr, nxt, cT, prior = ntm.class_transition(T(g[tag + "eta"]), sig, T(g[tag + "ema_t"]), 0.999, 0.999)
        corr.backward(gcorr)
        want = g[tag + "g_sigma_f64"]
        got = sig.grad.double().cpu().numpy()
        print("d sigma with g_corr = %-10s: max |err| / max|want| = %.3g   (want max %.3g)" % (name, np.abs(got - want).max() / np.abs(want).max(), np.abs(want).max()))
    # the same chain in torch fp64 from OUR fp32 corr: is ema_t_corr itself (fp32 forward) the source?
    print("ema_t_corr fp32 fixture vs fp64 fixture: %.3g" % (np.abs(g[tag + "ema_t_corr_f32"].astype(np.float64) - g[tag + "ema_t_corr_f64"]).max()))
