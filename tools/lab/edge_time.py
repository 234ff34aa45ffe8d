"""Lab: the fused EdgeConv tail (csrc/edgeconv.hip) at the four shapes of the model's two DGCNN_Propagation modules, 8 clouds:
HIP-event times of the forward call and of the gradient call (reduce + coefficients + dQ + dP), algorithmic bytes, and a
checksum of dP / dQ so that two builds can be compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import _lib  # noqa: E402
from geot_amd.ext._common import call, ptr  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402
from geot_amd.openpoints.models.backbone.transformer_ops import edgeconv_reverse_index  # noqa: E402

B = int(os.environ.get("B", "8"))
DEV = torch.device("cuda:0")
lib = _lib.load()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


xyz = torch.from_numpy(make_batch(B, 24000)[0]).to(DEV)
groups, k, slope, eps = 4, 4, 0.2, 1e-5
for name, c, nq, nk in (("pro_2.layer1", 512, 8192, 4096), ("pro_2.layer2", 384, 8192, 8192),
                        ("pro_1.layer1", 512, 4096, 512), ("pro_1.layer2", 384, 4096, 4096)):
    torch.manual_seed(c + nq + nk)
    q_pos, k_pos = xyz[:, :nq].contiguous(), xyz[:, :nk].contiguous()
    _, idx = knn_sorted(q_pos, k_pos, k)
    idx = idx.contiguous()
    rix = edgeconv_reverse_index(idx, nk)
    p, q = torch.randn(B, c, nk, device=DEV), torch.randn(B, c, nq, device=DEV)
    gamma, beta = torch.randn(c, device=DEV), torch.randn(c, device=DEV)
    g = torch.randn(B, c, nq, device=DEV)
    nbytes = int(lib.geot_edgeconv_ws_bytes(B, c, nq, nk, k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    out, ysel, ysum = torch.empty((3, B, c, nq), device=DEV).unbind(0)
    jsel = torch.empty((B, c, nq), dtype=torch.uint8, device=DEV)
    stats = torch.empty((B, groups, 2), device=DEV)
    gp, gq = torch.empty_like(p), torch.empty_like(q)
    gg, gb = torch.empty_like(gamma), torch.empty_like(beta)

    def fwd():
        call("geot_edgeconv_gn_max", DEV, B, c, nq, nk, k, groups, eps, slope, ptr(p), ptr(q), ptr(idx), ptr(gamma), ptr(beta),
             ptr(out), ptr(ysel), ptr(ysum), ptr(jsel), ptr(stats), ptr(ws), nbytes)

    def bwd():
        call("geot_edgeconv_gn_max_grad_rix", DEV, B, c, nq, nk, k, groups, slope, ptr(p), ptr(q), ptr(rix), ptr(gamma), ptr(beta),
             ptr(ysel), ptr(ysum), ptr(jsel), ptr(stats), ptr(g), ptr(gp), ptr(gq), ptr(gg), ptr(gb), ptr(ws), nbytes)

    t_f = timed(fwd)
    t_b = timed(bwd)
    # algorithmic bytes: forward P, Q read, out / ysel / ysum / jsel written, indices; gradient: reduce (ysel, g), dQ (ysel, ysum, g -> gq),
    # dP (ysel, g, Q, jsel, P -> gp, pair ids)
    fb = 4.0 * B * c * (nk + nq) + 4.0 * B * nq * k + 13.0 * B * c * nq
    bb = 8.0 * B * c * nq + 16.0 * B * c * nq + (13.0 * B * c * nq + 8.0 * B * c * nk + 4.0 * B * nq * k)
    ck = (float(gp.double().sum()), float(gp.double().abs().sum()), float(gq.double().abs().sum()))
    print("%-13s c=%3d nq=%4d nk=%4d  fwd %6.1f us (%4.2f TB/s)   grad %6.1f us (%4.2f TB/s)   checksum %.9e %.9e %.9e" %
          (name, c, nq, nk, t_f, fb / t_f / 1e6, t_b, bb / t_b / 1e6, *ck), flush=True)
