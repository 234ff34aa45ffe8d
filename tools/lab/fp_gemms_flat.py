"""Lab: the second convolution of an FP stage (1536 -> 384 at B x n points) as batched GEMMs on (B, C, n) / (B, n, C)
operands against ONE flat GEMM over the B n rows of point-major tensors (output point-major too), forward, data gradient
and weight gradient; TunableOp tunes every shape first (TUNE=1 path=...)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import tuning
tuning.enable(tune=True, path=os.environ.get("GEOT_TUNE_FILE", "/tmp/flat_tune.csv"))
DEV = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, C, Co = 8, 1536, 384
for n in (24000, 8192, 4096):
    W2 = torch.randn(Co, C, device=DEV)
    z_cl = torch.randn(B, n, C, device=DEV); gy2 = torch.randn(B, Co, n, device=DEV); gy2_cl = gy2.transpose(1, 2).contiguous()
    ex = lambda w: w.unsqueeze(0).expand(B, -1, -1)
    zf, gf = z_cl.view(B * n, C), gy2_cl.view(B * n, Co)
    rows = [
        ("G2  forward      ", lambda: torch.bmm(ex(W2), z_cl.transpose(1, 2)), lambda: torch.mm(zf, W2.t())),
        ("G2d data gradient", lambda: torch.bmm(gy2.transpose(1, 2), ex(W2)), lambda: torch.mm(gf, W2)),
        ("G2w weight grad  ", lambda: torch.bmm(gy2, z_cl).sum(0), lambda: torch.mm(gf.t(), zf)),
    ]
    for name, f_b, f_f in rows:
        print("n=%5d %s batched (cf out) %7.1f us   flat (point-major out) %7.1f us" % (n, name, timed(f_b), timed(f_f)), flush=True)
