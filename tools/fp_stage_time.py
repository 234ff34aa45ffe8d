"""HIP-event time of one PointnetFPModule training pass (forward + backward) at the model's three FP shapes, 8 clouds, in
the channels-first and the point-major layout of the first stage (transformer._fp_factored), GEMM selection as in bench.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd import tuning  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.pointnet2.pointnet2_modules import PointnetFPModule  # noqa: E402
from geot_amd.pointnet2 import pointnet2_utils as pu  # noqa: E402
from geot_amd.openpoints.models.backbone.transformer import _fp_factored  # noqa: E402
from geot_amd import fused_norm as fn  # noqa: E402

tuning.enable(path=os.environ.get("GEOT_TUNE_FILE"))
B = int(os.environ.get("B", "8"))
DEV = torch.device("cuda:0")
xyz = torch.from_numpy(make_batch(B, 24000)[0]).to(DEV)
for name, n, m, cs in (("prop0", 24000, 8192, 5), ("prop1", 8192, 512, 3), ("prop2", 4096, 512, 3)):
    torch.manual_seed(0)
    unknown, known = xyz[:, :n].contiguous(), xyz[:, :m].contiguous()
    fp = PointnetFPModule(mlp=[384 + cs, 1536, 384]).to(DEV).train()
    kf = torch.randn(B, 384, m, device=DEV, requires_grad=True)
    sk = torch.randn(B, cs, n, device=DEV)
    up = torch.randn(B, 384, n, device=DEV)
    d2, idx = pu._ext.three_nn(unknown, known)
    weight = pu._ext.fp_weights(d2)
    for layout in ("cf", "cl"):
        def once():
            nn3 = (idx, weight)
            if layout == "cl":      # the index plan's share (side stream in the model): timed separately below
                nn3 = (idx, weight, order_u, fn.ReverseIndex(idx, weight, m, order_k) if PLAN_INSIDE else rix)
            y = _fp_factored(fp, unknown, known, sk, kf, nn3, layout=layout)
            y.backward(up)
            kf.grad = None
            for p in fp.parameters():
                p.grad = None
        order_u = fn.local_spatial_order(unknown) if layout == "cl" else None
        order_k = fn.local_spatial_order(known) if layout == "cl" else None
        rix = fn.ReverseIndex(idx, weight, m, order_k) if layout == "cl" else None
        res = []
        for PLAN_INSIDE in ((False, True) if layout == "cl" else (False,)):
            for _ in range(3):
                once()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                once()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 10)
        print("%s (n=%d m=%d) %s: %7.3f ms forward + backward%s" %
              (name, n, m, layout, res[0], "" if len(res) == 1 else "   (%.3f with the reverse-index build inside)" % res[1]), flush=True)
