"""What between two graph phases breaks the replay?  look x6, then <variant>, then one graph call + one more (loss shows the damage)."""
import os, sys, gc, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd import train_step as ts, graph_step as gs
variant = sys.argv[1]
if variant == "nogc":
    gc.disable()
batches = _sup_batches(8, 24000)
torch.manual_seed(0)
init = PointTransformer_seg_T(**TOOTH_SEG_CFG).state_dict()
res = {}
for mode in ("eager", "mixed"):
    m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV); m.load_state_dict(init)
    step = ts.SupervisedStep(m)
    graphed = gs.GraphedSupervisedStep(step)
    call = graphed if mode == "mixed" else step
    torch.manual_seed(7)
    losses = []
    i = 0
    for _ in range(6):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]; i += 1
        losses.append(float(call(cur[0], cur[1], cur[2], next_pos=nxt[0])))
    rng = torch.cuda.get_rng_state()
    if variant == "sleep":
        torch.cuda.synchronize(); time.sleep(3)
    elif variant in ("other", "other_gc"):
        m2 = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV); m2.load_state_dict(init)
        s2 = ts.SupervisedStep(m2)
        for j in range(13):
            float(s2(batches[j % 2][0], batches[j % 2][1], batches[j % 2][2]))
        del m2, s2
        if variant == "other_gc":
            gc.collect()
        torch.cuda.set_rng_state(rng)
    elif variant in ("same", "nogc", "same_gc"):
        for _ in range(13):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]; i += 1
            losses.append(float(step(cur[0], cur[1], cur[2], next_pos=nxt[0])))
        if variant == "same_gc":
            gc.collect()
    elif variant == "allocs":
        junk = [torch.randn(1 << 20, device=DEV) for _ in range(2000)]
        del junk
        torch.cuda.set_rng_state(rng)
    elif variant == "memsets":       # eager global reductions: each one hipMemsetAsync's its semaphores
        big = torch.randn(8, 5, 24000, device=DEV)
        for _ in range(int(sys.argv[2])):
            big.sum((0, 2), dtype=torch.float64)
        torch.cuda.set_rng_state(rng)
    elif variant == "memset_api":    # plain hipMemsetAsync calls (tensor.zero_() on a contiguous tensor is a fill KERNEL; use the API)
        buf = torch.empty(1024, dtype=torch.uint8, device=DEV)
        rt = torch.cuda.cudart()
        for _ in range(int(sys.argv[2])):
            rt.cudaMemsetAsync(buf.data_ptr(), 0, 64, torch.cuda.current_stream().cuda_stream) if hasattr(rt, "cudaMemsetAsync") else None
        torch.cuda.set_rng_state(rng)
    elif variant == "memcpys":
        a, b = torch.randn(1 << 16, device=DEV), torch.empty(1 << 16, device=DEV)
        for _ in range(int(sys.argv[2])):
            b.copy_(a)
        torch.cuda.set_rng_state(rng)
    elif variant == "kernels":
        a = torch.randn(1 << 16, device=DEV)
        for _ in range(int(sys.argv[2])):
            a.mul_(1.0)
        torch.cuda.set_rng_state(rng)
    elif variant == "empty":
        torch.cuda.synchronize(); torch.cuda.empty_cache()
    for _ in range(3):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]; i += 1
        losses.append(float(call(cur[0], cur[1], cur[2], next_pos=nxt[0])))
    res[mode] = losses
bad = [j for j, (a, b) in enumerate(zip(res["eager"], res["mixed"])) if a != b]
print(variant, "first difference at call", bad[:1], "of", len(res["eager"]), flush=True)
