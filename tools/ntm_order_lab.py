"""Lab: how much do the threeD-loss kernels gain when points are processed in spatial order?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch, make_logits  # noqa: E402
from geot_amd import workloads as wl  # noqa: E402


def morton_order(x):
    q = np.clip(((x - x.min(0)) / (x.max(0) - x.min(0) + 1e-9) * 1023).astype(np.int64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")


B = 8
xyz_np = make_batch(B, 24000)[0]
for tag in ("random order", "morton order"):
    if tag == "morton order":
        xyz_np = np.stack([c[morton_order(c)] for c in xyz_np])
    xyz = torch.from_numpy(xyz_np).cuda()
    pw = torch.from_numpy(make_logits(xyz_np, 0)).cuda()
    ps = torch.from_numpy(make_logits(xyz_np, 1, sharp=3.0)).cuda()
    nt = wl.NtmHotPath().cuda()
    for _ in range(3):
        wl.ntm_step(nt, xyz, pw, ps)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        wl.ntm_step(nt, xyz, pw, ps)
    e1.record()
    torch.cuda.synchronize()
    print("%s: NTM step %.3f ms" % (tag, e0.elapsed_time(e1) / 10), flush=True)
