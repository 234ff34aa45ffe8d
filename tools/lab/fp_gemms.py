"""Lab: the six GEMMs of an FP stage (first conv on the known points, second conv, their gradients) in the channels-first
and the point-major operand layouts, prop0 shapes at 8 clouds; TunableOp file as given."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import tuning
tuning.enable(tune=os.environ.get("TUNE") == "1", path=os.environ.get("GEOT_TUNE_FILE"))
DEV = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, C, Ci, Co = 8, 1536, 384, 384
for n, m in ((24000, 8192), (8192, 512), (4096, 512)):
    Wa = torch.randn(C, Ci, device=DEV); W2 = torch.randn(Co, C, device=DEV)
    kf = torch.randn(B, Ci, m, device=DEV); z = torch.randn(B, C, n, device=DEV); gy2 = torch.randn(B, Co, n, device=DEV)
    ga = torch.randn(B, C, m, device=DEV)
    z_cl = z.transpose(1, 2).contiguous(); ga_cl = ga.transpose(1, 2).contiguous()
    ex = lambda w: w.unsqueeze(0).expand(B, -1, -1)
    rows = [
        ("G1  a = Wa kf        ", lambda: torch.bmm(ex(Wa), kf), lambda: torch.bmm(kf.transpose(1, 2), ex(Wa.t()))),
        ("G2  y2 = W2 z        ", lambda: torch.bmm(ex(W2), z), lambda: torch.bmm(ex(W2), z_cl.transpose(1, 2))),
        ("G2d gz = W2^T gy2    ", lambda: torch.bmm(ex(W2.t()), gy2), lambda: torch.bmm(gy2.transpose(1, 2), ex(W2))),
        ("G2w gW2 = gy2 z^T    ", lambda: torch.bmm(gy2, z.transpose(1, 2)).sum(0), lambda: torch.bmm(gy2, z_cl).sum(0)),
        ("G1d gkf = Wa^T ga    ", lambda: torch.bmm(ex(Wa.t()), ga), lambda: torch.bmm(ex(Wa.t()), ga_cl.transpose(1, 2))),
        ("G1w gWa = ga kf^T    ", lambda: torch.bmm(ga, kf.transpose(1, 2)).sum(0), lambda: torch.bmm(ga_cl.transpose(1, 2), kf.transpose(1, 2)).sum(0)),
    ]
    tc = tl = 0
    for name, f_cf, f_cl in rows:
        a, b = timed(f_cf), timed(f_cl)
        tc += a; tl += b
        print("n=%5d m=%5d %s cf %7.1f us   cl %7.1f us" % (n, m, name, a, b), flush=True)
    print("n=%5d m=%5d total                 cf %7.1f us   cl %7.1f us" % (n, m, tc, tl), flush=True)
