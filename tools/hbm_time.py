"""HIP-event timing of the HBM-bound kernels of the path at B clouds (the set tools/hbm_lab.py runs under rocprofv3):
average launch time over ITER launches and algorithmic bytes / time.  Lab tool."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch, make_logits  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402
from geot_amd import ntm  # noqa: E402
from geot_amd.openpoints.models.backbone.transformer_ops import graph_feature  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402

B, N, DEV = int(os.environ.get("B", "8")), 24000, "cuda"
ITER = int(os.environ.get("ITER", "20"))
CI = int(os.environ.get("CI", "384"))          # interpolation channels (384 = reference order, 1536 = factored FP)
xyz_np = make_batch(B, N)[0]
xyz = torch.from_numpy(xyz_np).to(DEV)
known = xyz[:, :8192].contiguous()
_, i3 = p2.three_nn(xyz, known)
w = torch.rand(B, N, 3, device=DEV); w = w / w.sum(2, keepdim=True)
fi = torch.randn(B, CI, 8192, device=DEV)
gi = torch.randn(B, CI, N, device=DEV)
feats = torch.randn(B, 64, N, device=DEV)
c6000 = p2.furthest_point_sampling(xyz, 6000)
new_xyz = p2.gather_points(xyz.transpose(1, 2).contiguous(), c6000).transpose(1, 2).contiguous()
bq = p2.ball_query(new_xyz, xyz, 0.1, 32)
go = torch.randn(B, 64, 6000, 32, device=DEV)
xq = torch.randn(B, 384, 8192, device=DEV)
_, kidx = knn_sorted(known, known, 4)
C = 17
logits = torch.from_numpy(make_logits(xyz_np, 1)).to(DEV)
prob = torch.softmax(logits, 1)
cm = torch.softmax(torch.randn(C, C, device=DEV), 1)
pred = ntm.Ins_T_mean(nclasses=C).to(DEV)
with torch.no_grad():
    insT = pred(prob, cm)
gT = torch.randn_like(insT)
MB = 1e6
from geot_amd import _lib  # noqa: E402
from geot_amd.ext._common import call, ptr  # noqa: E402
Wh = torch.stack([l.weight for l in pred.T_predictor.fc]).detach().contiguous()
gW = torch.zeros_like(Wh)
ws_sig = torch.empty(int(_lib.load().geot_ntm_sig_t_mean_ws_floats(B, N)), device=DEV)
ws_cor = torch.empty(int(_lib.load().geot_ntm_correct_ws_floats(B, N)), device=DEV)
gl, gI, gE, gout = torch.empty_like(logits), torch.empty_like(insT), torch.zeros(C, C, device=DEV), torch.randn_like(logits)
probc = prob.contiguous()
cases = {
    "three_interpolate fwd C=%d" % CI: (lambda: p2.three_interpolate(fi, i3, w), 4 * B * (CI * N + CI * 8192) + 24 * B * N),
    "three_interpolate bwd C=%d" % CI: (lambda: p2.three_interpolate_grad(gi, i3, w, 8192), 4 * B * (CI * N + CI * 8192) + 24 * B * N),
    "group_points fwd C=64 6000x32": (lambda: p2.group_points(feats, bq), 4 * B * (64 * 192000 + 192000 + 64 * N)),
    "group_points bwd C=64 6000x32": (lambda: p2.group_points_grad(go, bq, N), 4 * B * (64 * 192000 + 192000 + 64 * N)),
    "graph_feature fwd C=384 8192x4": (lambda: graph_feature(xq, xq, kidx), 4 * B * (2 * 384 * 8192 * 4 + 2 * 384 * 8192 + 8192 * 4)),
    "sig_t_mean fwd": (lambda: pred(prob, cm), 4 * B * N * (17 + 289)),
    "correct_logits fwd": (lambda: ntm.correct_logits(logits, insT, cm, 0.9), 4 * B * N * (17 + 289 + 17)),
    "sig_t_mean bwd (grad W)": (lambda: call("geot_ntm_sig_t_mean_grad_w", xyz.device, B, N, C, ptr(probc), ptr(Wh), ptr(cm), ptr(gT),
                                             ptr(gW), ptr(ws_sig)), 4 * B * N * (17 + 289)),
    "correct_logits bwd": (lambda: call("geot_ntm_correct_grad_ws", xyz.device, B, N, C, 0.9, ptr(logits), ptr(insT), ptr(cm), ptr(gout),
                                        ptr(gl), ptr(gI), ptr(gE), ptr(ws_cor)), 4 * B * N * (17 * 3 + 2 * 289)),
}
NU, MK = int(os.environ.get("NU", N)), int(os.environ.get("MK", "8192"))   # point-major cases: NU unknown <- MK known points
if CI >= 256 and CI % 4 == 0:
    # the point-major FP stage (csrc/channels_last.hip, profiles/DESIGN_r01_r03.md 4.11) at the same shape: 24000 <- 8192, CI channels
    from geot_amd import fused_norm as fnm
    lib = _lib.load()
    CS = 5
    unk, kn = xyz[:, :NU].contiguous(), xyz[:, :MK].contiguous()
    d3, i3c = p2.three_nn(unk, kn)
    w3 = p2.fp_weights(d3)
    a_cl = torch.randn(B, MK, CI, device=DEV)
    skip = torch.randn(B, CS, NU, device=DEV)
    wb = torch.randn(CI, CS, device=DEV)
    order_u, order_k = fnm.local_spatial_order(unk), fnm.local_spatial_order(kn)
    tiles = int(lib.geot_fp_front_cl_tiles(B, CI, NU, CS))
    y_cl = torch.empty(B, NU, CI, device=DEV)
    part = torch.empty(int(lib.geot_cl_stat_floats(tiles, CI)), device=DEV)
    dz_cl = torch.randn(B, NU, CI, device=DEV)
    rix = fnm.ReverseIndex(i3c, w3, MK, order_k)
    ga_cl = torch.empty(B, MK, CI, device=DEV)
    sc, sh, mu, rs, c1, c2 = (torch.rand(CI, device=DEV) + 0.5 for _ in range(6))
    sh = sh - 1.0
    tl = int(lib.geot_cl_tiles(1, B * NU, CI))
    pk = torch.empty(tl, 2 + 2 * CS, CI, device=DEV)
    z_cl = torch.empty_like(y_cl)
    call("geot_fp_front_cl", xyz.device, B, CI, MK, NU, CS, ptr(a_cl), ptr(i3c), ptr(w3), ptr(skip), ptr(wb), ptr(order_u), ptr(y_cl), ptr(part))
    row = 4 * B * CI
    cases.update({
        "fp_front_cl (Morton) C=%d" % CI: (lambda: call("geot_fp_front_cl", xyz.device, B, CI, MK, NU, CS, ptr(a_cl), ptr(i3c),
                                                        ptr(w3), ptr(skip), ptr(wb), ptr(order_u), ptr(y_cl), ptr(part)),
                                           row * (NU + MK) + 4 * B * CS * NU + 24 * B * NU),
        "bn_apply_cl C=%d" % CI: (lambda: call("geot_bn_apply_cl", xyz.device, B * NU, CI, 1, ptr(y_cl), ptr(sc), ptr(sh), ptr(z_cl)), 2 * row * NU),
        "bn_bwd_reduce_skip_cl C=%d" % CI: (lambda: call("geot_bn_bwd_reduce_skip_cl", xyz.device, B, NU, CI, CS, 1, ptr(y_cl), ptr(dz_cl), ptr(sc),
                                                         ptr(sh), ptr(mu), ptr(rs), ptr(skip), ptr(pk)), 2 * row * NU),
        "gather_rows_csr_cl C=%d" % CI: (lambda: call("geot_gather_rows_csr_cl", xyz.device, B, CI, NU, MK, 3, ptr(dz_cl),
                                                      ptr(rix.ws), ptr(order_k), ptr(ga_cl)), row * (NU + MK)),
        "gather_rows_csr_bn_cl C=%d" % CI: (lambda: call("geot_gather_rows_csr_bn_cl", xyz.device, B, CI, NU, MK, 3, 1,
                                                         ptr(y_cl), ptr(dz_cl), ptr(sc), ptr(sh), ptr(mu), ptr(rs), ptr(c1),
                                                         ptr(c2), ptr(rix.ws), ptr(order_k), ptr(ga_cl)),
                                            row * (2 * NU + MK)),
    })
only = os.environ.get("ONLY")
with torch.no_grad():
    for name, (fn, nbytes) in cases.items():
        if only and only not in name:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITER):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / ITER * 1e3
        print("%-36s %8.1f us  %7.1f MB  %6.2f TB/s  %4.1f %% of 8 TB/s" % (name, us, nbytes / MB, nbytes / us / 1e6, nbytes / us / 1e6 / 8 * 100), flush=True)
