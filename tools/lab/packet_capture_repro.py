"""Minimal reproducer of the ROCm 7.0 graph-packet-capture hazard: a captured torch reduction that zeroes its semaphores with
hipMemsetAsync (a memset NODE) returns garbage after a few thousand eager launches between two replays.
usage: packet_capture_repro.py [early|late|unset]   (when DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is put into the environment)"""
import os, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "unset"
if mode == "early":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
import torch
if mode == "late":          # after `import torch` (libamdhip64 is loaded) but before the first HIP call
    assert not torch.cuda.is_initialized()
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
dev = torch.device("cuda:0")
x = torch.randn(8, 5, 24000, device=dev)
out = torch.zeros(5, dtype=torch.float64, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        out.copy_(x.sum((0, 2), dtype=torch.float64))
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
big = int(os.environ.get("REPRO_NODES", "0"))
b = torch.randn(1 << 16, device=dev)
lin = torch.nn.Linear(128, 384).to(dev)
xin = torch.randn(4096, 128, device=dev)
gb = torch.zeros(384, device=dev)
with torch.cuda.stream(s):
    for _ in range(2):
        gb.copy_(torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0])
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for i in range(20):                      # several reductions: several memset nodes
        for _ in range(big // 20):
            b.mul_(1.0)
        out.copy_(x.sum((0, 2), dtype=torch.float64))
        y = lin(xin)
        gb.copy_(torch.autograd.grad(y.square().sum(), lin.bias)[0])
want = x.sum((0, 2), dtype=torch.float64)
g.replay(); torch.cuda.synchronize()
ok0 = torch.equal(out, want)
a = torch.randn(1 << 16, device=dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40000):
    a.mul_(1.0)
out.zero_()
g.replay(); torch.cuda.synchronize()
ok1 = torch.equal(out, want)
gb_want = torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0]
ok1 = ok1 and torch.equal(gb, gb_want)
print("%-6s env=%s : replay right before the eager launches %s, after them %s  %s" %
      (mode, os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), ok0, ok1, "" if ok1 else out.tolist()), flush=True)
