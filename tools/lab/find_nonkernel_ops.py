"""Which torch ops of one training iteration issue hipMemsetAsync / hipMemcpyAsync (they become memset / memcpy NODES when the
iteration is captured)?  torch.profiler over one eager M body (and P body) at the bench's sizes.  usage: find_nonkernel_ops.py model|fixmatch [clouds] [small]   (small: the test suite's model and cloud sizes)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd
import torch
from torch.profiler import profile, ProfilerActivity
from test_graph_step_gpu import _sup_batches, _fix_batch, DEV, SMALL
from geot_amd import train_step as ts
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
small = "small" in sys.argv
torch.manual_seed(0)
if which == "model":
    m = PointTransformer_seg_T(**(SMALL if small else TOOTH_SEG_CFG)).to(DEV)
    step = ts.SupervisedStep(m)
    b = _sup_batches(B, 6000 if small else 24000)[0]
    bodies = {"P": lambda: step.lookahead_work(b[0]), }
    pre = step.lookahead_work(b[0])
    bodies["M"] = lambda: step.iteration(b[0], b[1], b[2], pre, None)
else:
    step = (ts.build_fixmatch(DEV, seg_cfg=SMALL, cfg=dict(ts.NTM_CFG, threed_k=8), use_ddp=False) if small
            else ts.build_fixmatch(DEV, use_ddp=False))
    d, u = _fix_batch(3, 4096 if small else 24000)
    bodies = {"P": lambda: step.lookahead_work(d, u)}
    pre = step.lookahead_work(d, u)
    bodies["M"] = lambda: step.student_iteration(d, u, pre["geom_s"], pre["pseudo"], pre["knn"], ema_in_place=True)
for name, fn in bodies.items():
    fn(); fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
        fn(); torch.cuda.synchronize()
    evs = prof.events()
    # runtime calls named hipMemsetAsync / hipMemcpyAsync, attributed to the innermost enclosing CPU op by time containment
    cpu_ops = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and not e.name.startswith("hip")]
    rt = [e for e in evs if e.name in ("hipMemsetAsync", "hipMemcpyAsync", "hipMemcpyWithStream", "hipMemsetD32Async", "hipMemsetD8Async")]
    counts = collections.Counter()
    for r in rt:
        best = None
        for o in cpu_ops:
            if o.time_range.start <= r.time_range.start and o.time_range.end >= r.time_range.end:
                if best is None or (o.time_range.end - o.time_range.start) < (best.time_range.end - best.time_range.start):
                    best = o
        shape = str(best.input_shapes)[:70] if best is not None else ""
        counts[(r.name, best.name if best is not None else "?", shape)] += 1
    print("== %s %s: %d memset / memcpy runtime calls in one iteration" % (which, name, len(rt)))
    for (rn, on, sh), c in sorted(counts.items(), key=lambda kv: -kv[1]):
        print("   %3d x %-16s in %-40s %s" % (c, rn, on, sh))
