"""Lab: can an RCCL collective be captured into a hipGraph on this runtime, on the CURRENT stream, as kernel nodes only?
One rank, backend nccl.  Prints the node census of (a) a bare all_reduce, (b) all_reduce between two kernels, (c) an
all_gather_into_tensor, replays each with eager launches in between, and the host time of a replay."""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import geot_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from geot_amd import streams  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
y = torch.zeros(27_000_000, device=dev)          # the 108 MB of gradients
g = torch.zeros(4, 1 << 20, device=dev)
for _ in range(3):
    dist.all_reduce(x)
    dist.all_reduce(y)
    dist.all_gather_into_tensor(g.view(-1)[: x.numel()], x)
torch.cuda.synchronize()
print("launch mode", geot_amd.GRAPH_LAUNCH, "safe", geot_amd.graph_replay_is_safe(), flush=True)


def probe(name, body, check):
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    try:
        with streams.capture(graph, dev):
            body()
    except Exception as e:      # noqa: BLE001
        print(name, "CAPTURE FAILED:", type(e).__name__, str(e)[:300], flush=True)
        return
    print(name, "nodes", streams.node_types(graph), flush=True)
    for it in range(3):
        for _ in range(2000):
            x.add_(0)           # eager launches between replays
        t0 = time.perf_counter()
        graph.replay()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("   replay %d: host %.3f ms, check %s" % (it, host * 1e3, check()), flush=True)


x.fill_(1.0)
probe("all_reduce", lambda: dist.all_reduce(x), lambda: float(x.sum()) == float(x.numel()))
probe("kernel + all_reduce(108 MB) + kernel", lambda: (y.add_(1.0), dist.all_reduce(y), y.mul_(1.0)), lambda: float(y[0]))
probe("all_gather", lambda: dist.all_gather_into_tensor(g.view(-1)[: x.numel()], x), lambda: float(g.view(-1)[0]))
side = torch.cuda.Stream()
dist.destroy_process_group()
print("done")
