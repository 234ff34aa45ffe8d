"""Per GEMM shape of one supervised step: launches, device time (torch.profiler, kernels attributed to the enclosing aten op),
flops and the fraction of the fp32 MFMA peak -- which shapes the step's 21.7 ms of GEMM time is made of.  usage: gemm_table.py [clouds]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geot_amd
import torch
from torch.profiler import profile, ProfilerActivity
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd import train_step as ts, tuning
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
print("TunableOp file:", tuning.enable(path=os.environ.get("TUNE_FILE")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = ts.SupervisedStep(m)
b = _sup_batches(B, 24000)[0]
pre = step.lookahead_work(b[0])
for _ in range(3):
    step.iteration(b[0], b[1], b[2], pre, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step.iteration(b[0], b[1], b[2], pre, None)
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e in prof.key_averages(group_by_input_shape=True):
    if e.key not in ("aten::mm", "aten::bmm", "aten::addmm", "aten::baddbmm"):
        continue
    sh = [s for s in e.input_shapes if len(s) >= 2]
    if e.key in ("aten::mm", "aten::addmm"):
        a, bb = sh[-2:]
        fl = 2.0 * a[0] * a[1] * bb[1]
    else:
        a, bb = sh[-2:]
        fl = 2.0 * a[0] * a[1] * a[2] * bb[2]
    r = rows[(e.key, str(sh))]
    r[0] += e.count; r[1] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total; r[2] += fl * e.count
tot_t = sum(r[1] for r in rows.values()); tot_f = sum(r[2] for r in rows.values())
print("%d GEMM launches, %.2f ms, %.1f GFLOP, %.1f TFLOP/s average (%.0f %% of 157.3)" % (
    sum(r[0] for r in rows.values()), tot_t / 1e3, tot_f / 1e9, tot_f / tot_t / 1e6, 100 * tot_f / tot_t / 1e6 / 157.3))
print("%6s %9s %9s %8s %7s  %s" % ("calls", "ms total", "us each", "GFLOP", "% peak", "op, shapes"))
for (k, sh), r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%6d %9.3f %9.1f %8.2f %7.1f  %s %s" % (r[0], r[1] / 1e3, r[1] / r[0], r[2] / r[0] / 1e9, 100 * r[2] / r[1] / 1e6 / 157.3, k[6:], sh))
