"""Lab: how fast the point-major FP front end runs when the neighbour ids are perfectly local (element e interpolates
from rows e/3, e/3+1, e/3+2) against the real three_nn ids in memory / Morton order: separates 'the kernel' from 'the
locality of the data'."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import _lib, ntm  # noqa: E402
from geot_amd.ext._common import call, ptr  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402

B, C, n, m, cs = 8, 1536, 24000, 8192, 5
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
unknown, known = xyz[:, :n].contiguous(), xyz[:, :m].contiguous()
d2, idx = p2.three_nn(unknown, known)
w = p2.fp_weights(d2)
o = ntm.spatial_order(unknown).view(B, n)
order = (o - torch.arange(B, device=DEV, dtype=torch.int32).view(B, 1) * n).contiguous()
e = torch.arange(n, device=DEV, dtype=torch.int32)
local = torch.stack([(e // 3 + t) % m for t in range(3)], 1).unsqueeze(0).expand(B, -1, -1).contiguous()
rnd = torch.randint(0, m, (B, n, 3), device=DEV, dtype=torch.int32)
a_cl = torch.randn(B, m, C, device=DEV)
skip = torch.randn(B, cs, n, device=DEV)
wb = torch.randn(C, cs, device=DEV)
tiles = int(lib.geot_fp_front_cl_tiles(B, C, n, cs))
y = torch.empty(B, n, C, device=DEV)
part = torch.empty(int(lib.geot_cl_stat_floats(tiles, C)), device=DEV)
# how local is the Morton sequence?  distinct table rows per window of consecutive elements
ids = torch.gather(idx.long(), 1, order.long().unsqueeze(-1).expand(-1, -1, 3))[0].cpu()
for win in (64, 256, 1024, 4096):
    d = sum(len(torch.unique(ids[s:s + win])) for s in range(0, n - win, win)) / len(range(0, n - win, win))
    print("Morton order: %5d consecutive elements touch %7.1f distinct table rows (%.2f per element)" % (win, d, d / win))
for tag, ii, oo in (("perfectly local ids", local, None), ("three_nn ids, memory order", idx, None),
                    ("three_nn ids, Morton order", idx, order), ("random ids", rnd, None)):
    t = timed(lambda: call("geot_fp_front_cl", DEV, B, C, m, n, cs, ptr(a_cl), ptr(ii), ptr(w), ptr(skip), ptr(wb), ptr(oo),
                           ptr(y), ptr(part)))
    print("%-28s %7.1f us" % (tag, t), flush=True)
