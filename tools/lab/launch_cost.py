import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from geot_amd.ext import pointnet2_ext as p2
x = torch.rand(1, 64, 3, device="cuda")
y = torch.rand(64, device="cuda")
for name, fn in (("geot fp_weights", lambda: p2.fp_weights(x)), ("torch add", lambda: y.add_(1.0))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: %-16s host %.2f us/call, with drain %.2f us/call" % (os.getcwd()[-24:], name, (t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
