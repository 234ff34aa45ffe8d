"""Lab: time of the point-major FP front end against the number of points n and table rows m (8 clouds, C = 1536)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import _lib, ntm  # noqa: E402
from geot_amd.ext._common import call, ptr  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402

B, C, cs = 8, 1536, 5
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
for n, m in ((8192, 512), (24000, 512), (24000, 2048), (24000, 8192), (8192, 8192), (16000, 8192)):
    unknown, known = xyz[:, :n].contiguous(), xyz[:, :m].contiguous()
    d2, idx = p2.three_nn(unknown, known)
    w = p2.fp_weights(d2)
    o = ntm.spatial_order(unknown).view(B, n)
    order = (o - torch.arange(B, device=DEV, dtype=torch.int32).view(B, 1) * n).contiguous()
    a_cl = torch.randn(B, m, C, device=DEV)
    skip = torch.randn(B, cs, n, device=DEV)
    wb = torch.randn(C, cs, device=DEV)
    tiles = int(lib.geot_fp_front_cl_tiles(B, C, n, cs))
    y = torch.empty(B, n, C, device=DEV)
    part = torch.empty(int(lib.geot_cl_stat_floats(tiles, C)), device=DEV)
    t = timed(lambda: call("geot_fp_front_cl", DEV, B, C, m, n, cs, ptr(a_cl), ptr(idx), ptr(w), ptr(skip), ptr(wb), ptr(order),
                           ptr(y), ptr(part)))
    nbytes = 4.0 * B * C * (n + m)
    print("n=%5d m=%5d tiles %4d: %7.1f us  %5.2f TB/s  (%.2f ns per row)" % (n, m, tiles, t, nbytes / t / 1e6, t * 1e3 / (B * n)), flush=True)
