"""HIP-event time of the fused FP front end (geot_fp_front) at the three shapes of the model's FP modules, 8 clouds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2  # noqa: E402
from geot_amd.fused_norm import fp_front  # noqa: E402

B = int(os.environ.get("B", "8"))
xyz = torch.from_numpy(make_batch(B, 24000)[0]).cuda()
for name, n, m, cs in (("prop0", 24000, 8192, 5), ("prop1", 8192, 512, 3), ("prop2", 4096, 512, 3)):
    unknown, known = xyz[:, :n].contiguous(), xyz[:, :m].contiguous()
    d2, idx = p2.three_nn(unknown, known)
    w = p2.fp_weights(d2)
    a = torch.randn(B, 1536, m, device="cuda")
    skip = torch.randn(B, cs, n, device="cuda")
    wb = torch.randn(1536, cs, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            fp_front(a, idx, w, skip, wb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fp_front(a, idx, w, skip, wb)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = 4.0 * B * (1536 * n + 1536 * m + cs * n) + 24.0 * B * n
    print("fp_front %-6s n=%5d m=%5d  %8.1f us  %6.2f TB/s" % (name, n, m, us, nbytes / us / 1e6), flush=True)
