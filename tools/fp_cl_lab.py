"""Lab: the point-major (B, N, C) FP front end, its BatchNorm passes and its gradient against the channels-first kernels
at the shapes of the model's FP modules (8 clouds): values compared, HIP-event times side by side."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd import _lib  # noqa: E402
from geot_amd.ext._common import call, ptr  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402
from geot_amd.fused_norm import fp_front  # noqa: E402
from geot_amd import ntm  # noqa: E402

B = int(os.environ.get("B", "8"))
C = int(os.environ.get("C", "1536"))
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


def local_order(pos):
    b, n, _ = pos.shape
    o = ntm.spatial_order(pos.contiguous()).view(b, n)
    return (o - torch.arange(b, device=pos.device, dtype=torch.int32).view(b, 1) * n).contiguous()


xyz = torch.from_numpy(make_batch(B, 24000)[0]).to(DEV)
for name, n, m, cs in (("prop0", 24000, 8192, 5), ("prop1", 8192, 512, 3), ("prop2", 4096, 512, 3)):
    torch.manual_seed(0)
    unknown, known = xyz[:, :n].contiguous(), xyz[:, :m].contiguous()
    d2, idx = p2.three_nn(unknown, known)
    w = p2.fp_weights(d2)
    a = torch.randn(B, C, m, device=DEV)
    skip = torch.randn(B, cs, n, device=DEV)
    wb = torch.randn(C, cs, device=DEV)
    a_cl = a.transpose(1, 2).contiguous()
    nbytes = 4.0 * B * (C * n + C * m + cs * n) + 24.0 * B * n
    with torch.no_grad():
        y_cf, part_cf = fp_front(a, idx, w, skip, wb)
        t_cf = timed(lambda: fp_front(a, idx, w, skip, wb))
        tiles = int(lib.geot_fp_front_cl_tiles(B, C, n, cs))
        y_cl = torch.empty(B, n, C, device=DEV)
        part = torch.empty(int(lib.geot_cl_stat_floats(tiles, C)), device=DEV)
        res = {}
        for tag, order in (("memory order", None), ("Morton order", local_order(unknown))):
            y_cl.zero_()
            f = lambda: call("geot_fp_front_cl", DEV, B, C, m, n, cs, ptr(a_cl), ptr(idx), ptr(w), ptr(skip), ptr(wb),  # noqa: E731
                             ptr(order), ptr(y_cl), ptr(part))
            f()
            same = torch.equal(y_cl.transpose(1, 2), y_cf)
            sums = torch.empty(C, 2, dtype=torch.float64, device=DEV)
            call("geot_bn_sums_shifted_cl", DEV, tiles, C, ptr(part), ptr(sums))
            ref = torch.stack([y_cf.double().sum((0, 2)), (y_cf.double() ** 2).sum((0, 2))], 1)
            serr = float(((sums - ref).abs() / torch.stack([y_cf.double().abs().sum((0, 2)), ref[:, 1]], 1)).max())
            res[tag] = (timed(f), same, serr)
        print("%s fwd  (n=%d m=%d C=%d): channels-first %7.1f us %5.2f TB/s" % (name, n, m, C, t_cf, nbytes / t_cf / 1e6))
        for tag, (t, same, serr) in res.items():
            print("      point-major, %s: %7.1f us %5.2f TB/s   values identical: %s   sums rel err %.1e" %
                  (tag, t, nbytes / t / 1e6, same, serr), flush=True)

        # gradient of the interpolation
        gy = torch.randn(B, C, n, device=DEV)
        gy_cl = gy.transpose(1, 2).contiguous()
        ga_cf = p2.three_interpolate_grad(gy, idx, w, m)
        t_gcf = timed(lambda: p2.three_interpolate_grad(gy, idx, w, m))
        ws_ints = int(lib.geot_rix_ws_ints(B, n, m, 3))
        ws = torch.empty(ws_ints, dtype=torch.int32, device=DEV)
        ga_cl = torch.empty(B, m, C, device=DEV)
        gb = 4.0 * B * (C * n + C * m)
        print("%s grad: channels-first (index build inside) %7.1f us %5.2f TB/s" % (name, t_gcf, gb / t_gcf / 1e6))
        for tag, order in (("memory order", None), ("Morton order", local_order(known))):
            build = lambda: call("geot_rix_build", DEV, B, n, m, 3, ptr(idx), ptr(w), ptr(order), ptr(ws), ws_ints)  # noqa: E731
            build()
            t_build = timed(build)
            ga_cl.fill_(float("nan"))
            g = lambda: call("geot_gather_rows_csr_cl", DEV, B, C, n, m, 3, ptr(gy_cl), ptr(ws), ptr(order), ptr(ga_cl))  # noqa: E731
            g()
            first = ga_cl.clone()
            build()
            g()
            err = float((ga_cl.transpose(1, 2) - ga_cf).abs().max() / ga_cf.abs().max())
            t = timed(g)
            print("      point-major gather, %s: %7.1f us %5.2f TB/s (+ index build %5.1f us)  vs channels-first max rel %.1e   "
                  "rebuild + rerun identical: %s" % (tag, t, gb / t / 1e6, t_build, err, torch.equal(first, ga_cl)), flush=True)

        # BatchNorm passes on the 1.18 GB tensor
        if name == "prop0":
            R = B * n
            scale, shift, mean, rstd, k0, c1, c2 = (torch.rand(C, device=DEV) + 0.5 for _ in range(7))
            shift = shift - 1.0
            out_cf, out_cl = torch.empty_like(y_cf), torch.empty_like(y_cl)
            t1 = timed(lambda: call("geot_bn_apply", DEV, B, C, n, 1, ptr(y_cf), ptr(scale), ptr(shift), ptr(out_cf)))
            t2 = timed(lambda: call("geot_bn_apply_cl", DEV, R, C, 1, ptr(y_cl), ptr(scale), ptr(shift), ptr(out_cl)))
            print("bn apply       cf %6.1f us  cl %6.1f us  identical %s" % (t1, t2, torch.equal(out_cl.transpose(1, 2), out_cf)))
            slices = int(lib.geot_bn_slices(B, C, n))
            p_cf = torch.empty(B, C, slices, 2, device=DEV)      # (backward-reduce partials: plain pairs)
            tl = int(lib.geot_cl_tiles(1, R, C))
            p_cl = torch.empty(tl, 2, C, device=DEV)
            t1 = timed(lambda: call("geot_bn_bwd_reduce", DEV, B, C, n, 1, ptr(y_cf), ptr(gy), ptr(scale), ptr(shift), ptr(mean),
                                    ptr(rstd), ptr(p_cf)))
            t2 = timed(lambda: call("geot_bn_bwd_reduce_cl", DEV, R, C, 1, ptr(y_cl), ptr(gy_cl), ptr(scale), ptr(shift), ptr(mean),
                                    ptr(rstd), ptr(p_cl)))
            s_cf = torch.empty(C, 2, dtype=torch.float64, device=DEV)
            s_cl = torch.empty(C, 2, dtype=torch.float64, device=DEV)
            call("geot_bn_sums", DEV, B, C, slices, ptr(p_cf), ptr(s_cf))
            call("geot_bn_sums_cl", DEV, tl, C, ptr(p_cl), ptr(s_cl))
            print("bn bwd reduce  cf %6.1f us  cl %6.1f us  sums rel diff %.1e" %
                  (t1, t2, float(((s_cf - s_cl).abs() / s_cf.abs().clamp_min(1e-30)).max())))
            dx_cf, dx_cl = torch.empty_like(y_cf), torch.empty_like(y_cl)
            t1 = timed(lambda: call("geot_bn_bwd_apply", DEV, B, C, n, 1, ptr(y_cf), ptr(gy), ptr(scale), ptr(shift), ptr(mean),
                                    ptr(rstd), ptr(k0), ptr(c1), ptr(c2), ptr(dx_cf)))
            t2 = timed(lambda: call("geot_bn_bwd_apply_cl", DEV, R, C, 1, ptr(y_cl), ptr(gy_cl), ptr(scale), ptr(shift), ptr(mean),
                                    ptr(rstd), ptr(k0), ptr(c1), ptr(c2), ptr(dx_cl)))
            print("bn bwd apply   cf %6.1f us  cl %6.1f us  identical %s" % (t1, t2, torch.equal(dx_cl.transpose(1, 2), dx_cf)))
            t1 = timed(lambda: call("geot_bn_sums_cl", DEV, tl, C, ptr(p_cl), ptr(s_cl)))
            print("bn sums cl (%d tiles) %6.1f us" % (tl, t1), flush=True)
            # the fused backward: reduce + skip sums, and the gather that forms gy on the fly (against apply + gather above)
            k = 2 + 2 * cs
            p_k = torch.empty(tl, k, C, device=DEV)
            t1 = timed(lambda: call("geot_bn_bwd_reduce_skip_cl", DEV, B, n, C, cs, 1, ptr(y_cl), ptr(gy_cl), ptr(scale), ptr(shift),
                                    ptr(mean), ptr(rstd), ptr(skip), ptr(p_k)))
            s_k = torch.empty(C, k, dtype=torch.float64, device=DEV)
            t2 = timed(lambda: call("geot_bn_sums_k_cl", DEV, tl, C, k, ptr(p_k), ptr(s_k)))
            order = local_order(known)
            call("geot_rix_build", DEV, B, n, m, 3, ptr(idx), ptr(w), ptr(order), ptr(ws), ws_ints)
            t3 = timed(lambda: call("geot_gather_rows_csr_bn_cl", DEV, B, C, n, m, 3, 1, ptr(y_cl), ptr(gy_cl), ptr(scale), ptr(shift),
                                    ptr(mean), ptr(rstd), ptr(c1), ptr(c2), ptr(ws), ptr(order), ptr(ga_cl)))
            call("geot_gather_rows_csr_cl", DEV, B, C, n, m, 3, ptr(dx_cl), ptr(ws), ptr(order), ptr(out_cl[:, :m].contiguous()))
            ref = torch.empty(B, m, C, device=DEV)
            call("geot_bn_bwd_apply_cl", DEV, R, C, 1, ptr(y_cl), ptr(gy_cl), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), ptr(scale),
                 ptr(c1), ptr(c2), ptr(dx_cl))
            call("geot_gather_rows_csr_cl", DEV, B, C, n, m, 3, ptr(dx_cl), ptr(ws), ptr(order), ptr(ref))
            err = float((ga_cl - ref).abs().max() / ref.abs().max())
            print("fused backward: reduce + skip sums %6.1f us, sums (%d x %d) %5.1f us, gather with the BatchNorm backward inside "
                  "%6.1f us (vs apply + gather: max rel %.1e)" % (t1, tl, k, t2, t3, err), flush=True)
    del a, a_cl, y_cf, y_cl, gy, gy_cl, ga_cf, ga_cl
