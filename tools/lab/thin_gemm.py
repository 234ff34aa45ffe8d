"""Lab: the thin GEMMs of the FP front end's backward (grad of the 5-channel skip weights) as library GEMMs, 8 clouds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import tuning, _lib
from geot_amd.ext._common import call, ptr
tuning.enable(path=os.environ.get("GEOT_TUNE_FILE"))
DEV = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, C = 8, 1536
lib = _lib.load()
for n, cs in ((24000, 5), (8192, 3), (4096, 3)):
    gy = torch.randn(B, C, n, device=DEV); skip = torch.randn(B, cs, n, device=DEV)
    gy_cl = gy.transpose(1, 2).contiguous()
    t1 = timed(lambda: torch.bmm(gy, skip.transpose(1, 2)).sum(0))
    t2 = timed(lambda: torch.bmm(skip, gy_cl).sum(0).t())
    t3 = timed(lambda: torch.matmul(skip.transpose(0, 1).reshape(cs, B * n), gy_cl.view(B * n, C)))
    s = int(lib.geot_rowdot_small_slices(C, n))
    part = torch.empty(B, C, s, cs, device=DEV)
    def rd():
        for b in range(B):
            call("geot_rowdot_small", DEV, C, n, cs, ptr(gy[b]), ptr(skip[b]), ptr(part[b]))
        return part.sum((0, 2))
    t4 = timed(rd)
    ref = torch.bmm(gy.double(), skip.double().transpose(1, 2)).sum(0)
    err = float((rd().double() - ref).abs().max() / ref.abs().max())
    print("n=%5d cs=%d: bmm(gy, skip^T).sum %7.1f us | cl bmm(skip, gy_cl).sum %7.1f us | cl one GEMM (cs x Bn)(Bn x C) %7.1f us | rowdot_small x B %7.1f us (err %.1e)"
          % (n, cs, t1, t2, t3, t4, err), flush=True)
