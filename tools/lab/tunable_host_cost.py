import time, torch, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd import tuning
a = torch.randn(4096, 384, device="cuda"); b = torch.randn(384, 1536, device="cuda")
w = torch.randn(1536, 384, device="cuda")
def host_us(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("tunable off: mm %.1f us  linear %.1f us  bmm %.1f us" % (host_us(lambda: torch.mm(a, b)), host_us(lambda: torch.nn.functional.linear(a, w)), host_us(lambda: torch.bmm(a.view(8, 512, 384), b.expand(8, -1, -1)))))
path = tuning.enable()
print("file", path)
print("tunable on : mm %.1f us  linear %.1f us  bmm %.1f us" % (host_us(lambda: torch.mm(a, b)), host_us(lambda: torch.nn.functional.linear(a, w)), host_us(lambda: torch.bmm(a.view(8, 512, 384), b.expand(8, -1, -1)))))
