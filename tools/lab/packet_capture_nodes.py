"""Which node TYPES does graph packet capture (the ROCm 7.0 default) break when eager launches run between two replays?
One graph per type -- kernel nodes only / memcpy nodes / memset nodes -- replay, N eager launches, replay, compare.
Run WITHOUT geot_amd imported (the package switches packet capture off)."""
import os, sys, ctypes
import torch
dev = torch.device("cuda:0")
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
a = torch.randn(1 << 18, device=dev)
outs = {k: torch.zeros(1 << 18, device=dev) for k in ("kernel", "memcpy", "memset", "mm", "linear", "linear_bwd", "bigsum", "combo", "combo_rev", "memset_f32", "kchain", "kcombo")}
red32 = torch.zeros(5, device=dev)
lin = torch.nn.Linear(128, 384).to(dev)
xin = torch.randn(4096, 128, device=dev)
wm = torch.randn(128, 384, device=dev)
ymm = torch.zeros(4096, 384, device=dev)
gb = torch.zeros(384, device=dev)
big = torch.randn(4096, 384, device=dev)
bs = torch.zeros(384, device=dev)
x = torch.randn(8, 5, 24000, device=dev)
red = torch.zeros(5, dtype=torch.float64, device=dev)

REPS = int(os.environ.get('REPS', '20'))
ONLY = os.environ.get('ONLY')
def body(kind):
    for _ in range(REPS):
        if kind == "kernel":
            torch.mul(a, 2.0, out=outs[kind])                      # plain kernel nodes
        elif kind == "memcpy":
            outs[kind].copy_(a)                                    # contiguous same-dtype copy: hipMemcpyAsync -> memcpy node
        elif kind == "memset":
            red.copy_(x.sum((0, 2), dtype=torch.float64))          # global reduction: hipMemsetAsync of its semaphores -> memset node
            outs[kind].copy_(a)
        elif kind == "mm":
            torch.mm(xin, wm, out=ymm)                             # rocBLAS / hipBLASLt GEMM node(s)
            outs[kind].copy_(a)
        elif kind == "linear":
            with torch.no_grad():
                ymm.copy_(lin(xin))                                # hipBLASLt GEMM + bias epilogue
            outs[kind].copy_(a)
        elif kind == "linear_bwd":
            gb.copy_(torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0])
            outs[kind].copy_(a)
        elif kind == "combo":
            red.copy_(x.sum((0, 2), dtype=torch.float64))
            gb.copy_(torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0])
            outs[kind].copy_(a)
        elif kind == "combo_rev":
            gb.copy_(torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0])
            red.copy_(x.sum((0, 2), dtype=torch.float64))
            outs[kind].copy_(a)
        elif kind == "memset_f32":
            red32.copy_(x.sum((0, 2)))
            outs[kind].copy_(a)
        elif kind == "kchain":
            y = a
            for _ in range(50):                                    # dependent out-of-place kernels: pool blocks are reused down the chain
                y = y * 1.0001 + 1.0
            torch.mul(y, 1.0, out=outs[kind])
        elif kind == "kcombo":                                     # the combo with its copies done by KERNELS (mul ... out=) -- memsets remain
            torch.mul(x.sum((0, 2), dtype=torch.float64), 1.0, out=red)
            torch.mul(torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0], 1.0, out=gb)
            torch.mul(a, 1.0, out=outs[kind])
        elif kind == "bigsum":
            bs.copy_(big.sum(0))                                   # column sums over 4096 rows
            outs[kind].copy_(a)
graphs = {}
s = torch.cuda.Stream()
for kind in outs:
    if ONLY and kind not in ONLY.split(','):
        continue
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(kind)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        body(kind)
    raw = ctypes.c_void_p(g.raw_cuda_graph()); n = ctypes.c_size_t(0)
    hip.hipGraphGetNodes(raw, None, ctypes.byref(n))
    nodes = (ctypes.c_void_p * n.value)(); hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n))
    types = []
    for nd in nodes:
        t = ctypes.c_int(-1); hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t)); types.append(t.value)
    graphs[kind] = (g, {k: types.count(v) for k, v in (("kernel", 0), ("memcpy", 1), ("memset", 2)) if types.count(v)})
want_red = x.sum((0, 2), dtype=torch.float64)
with torch.no_grad():
    want = {"mm": (ymm, torch.mm(xin, wm)), "linear": (ymm, lin(xin).clone()), "bigsum": (bs, big.sum(0))}
want["linear_bwd"] = (gb, torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0].clone())
def good(kind):
    ok = torch.equal(outs[kind], a * 2.0 if kind == "kernel" else a) and (kind != "memset" or torch.equal(red, want_red))
    if kind in want:
        ok = ok and torch.equal(want[kind][0], want[kind][1])
    if kind in ("combo", "combo_rev"):
        if os.environ.get("VERBOSE"):
            print("   ", kind, "red ok", torch.equal(red, want_red), "gb ok", torch.equal(gb, want["linear_bwd"][1]), "max|gb diff| %.3g" % float((gb - want["linear_bwd"][1]).abs().max()))
        ok = ok and torch.equal(red, want_red) and torch.equal(gb, want["linear_bwd"][1])
    if kind == "kchain":
        y = a
        for _ in range(50):
            y = y * 1.0001 + 1.0
        return torch.equal(outs[kind], y)
    if kind == "kcombo":
        return torch.equal(red, want_red) and torch.equal(gb, want["linear_bwd"][1]) and torch.equal(outs[kind], a)
    if kind == "memset_f32":
        ok = ok and torch.equal(red32, x.sum((0, 2)))
    return ok
for kind, (g, counts) in graphs.items():
    g.replay(); torch.cuda.synchronize()
    ok0 = good(kind)
    b = torch.randn(1 << 16, device=dev)
    for _ in range(N):
        b.mul_(1.0)
    outs[kind].zero_(); red.zero_(); ymm.zero_(); gb.zero_(); bs.zero_(); red32.zero_()
    g.replay(); torch.cuda.synchronize()
    ok1 = good(kind)
    print("%-7s nodes %-40s first replay %s, replay after %d eager launches %s" % (kind, counts, ok0, N, ok1), flush=True)
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
