"""Which TunableOp keys does the second half of the supervised iteration use (the backward of the transformer blocks and the patch
encoder: what runs beside the look-ahead graph)?  Tuning is enabled only around backward_rest_update, with a 1-iteration budget:
the file written holds exactly those keys.  usage: m2_gemm_keys.py out.csv [clouds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS"] = "1"
os.environ["PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS"] = "1"
os.environ["PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS"] = "0"
import geot_amd
import torch
import torch.cuda.tunable as tunable
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd import train_step as ts, tuning
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if os.path.exists(out):
    os.remove(out)
tuning.enable(tune=True, path=out)
tunable.tuning_enable(False)
torch.manual_seed(0)
m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = ts.SupervisedStep(m)
b = _sup_batches(B, 24000)[0]
for i in range(2):
    loss, rest = step.forward_backward_head(b[0], b[1], b[2])
    tunable.tuning_enable(True)
    step.backward_rest_update(rest)
    tunable.tuning_enable(False)
    torch.cuda.synchronize()
print("done")
