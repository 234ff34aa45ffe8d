"""Dense flops of one supervised training step (torch's FlopCounterMode: mm / bmm / addmm / convolution), per operator
shape, to put the step's GEMM time (21.7 ms at 8 clouds) against the fp32 MFMA peak.  usage: step_flops.py [clouds]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd
import torch
from torch.utils.flop_counter import FlopCounterMode
from torch.utils._python_dispatch import TorchDispatchMode
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd import train_step as ts
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = ts.SupervisedStep(m)
b = _sup_batches(B, 24000)[0]
step(b[0], b[1], b[2]); step(b[0], b[1], b[2])
shapes = collections.Counter()


class Shapes(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name in ("mm", "bmm", "addmm", "baddbmm", "convolution", "convolution_backward"):
            sh = tuple(tuple(a.shape) for a in args if torch.is_tensor(a))
            if name in ("mm", "addmm"):
                a, bb = [a for a in args if torch.is_tensor(a)][-2:]
                fl = 2 * a.shape[0] * a.shape[1] * bb.shape[1]
            elif name in ("bmm", "baddbmm"):
                a, bb = [a for a in args if torch.is_tensor(a)][-2:]
                fl = 2 * a.shape[0] * a.shape[1] * a.shape[2] * bb.shape[2]
            else:
                fl = 0
            shapes[(name, sh)] += fl
        return func(*args, **(kwargs or {}))


with FlopCounterMode(display=False) as fc:
    step(b[0], b[1], b[2])
total = fc.get_total_flops()
print("clouds %d: %.1f GFLOP per step in mm/bmm/conv (%.1f per cloud); at 21.68 ms of GEMM kernels: %.1f TFLOP/s average"
      % (B, total / 1e9, total / 1e9 / B, total / 21.68e-3 / 1e12))
with Shapes():
    step(b[0], b[1], b[2])
rows = sorted(shapes.items(), key=lambda kv: -kv[1])
print("largest operator shapes (GFLOP per step):")
for (name, sh), fl in rows[:30]:
    print("  %8.1f  %-8s %s" % (fl / 1e9, name, sh))
