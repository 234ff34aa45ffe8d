"""TunableOp tuning of the supervised step's GEMMs WHILE an FPS runs on a side stream: picks, per shape, the solution that is
fastest under the contention of profiles/r04_fps_beside.txt instead of alone.  Writes the results file given as argv[1].
usage: tune_beside_fps.py out.csv [clouds] [beside=1|0]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "15")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "2")
import geot_amd
import torch
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd import train_step as ts, tuning
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd.pointops.functions import pointops as pops
out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
beside = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
if os.path.exists(out):
    os.remove(out)
print("tuning into", tuning.enable(tune=True, path=out), "beside an FPS" if beside else "alone", flush=True)
torch.manual_seed(0)
m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = ts.SupervisedStep(m)
b = _sup_batches(B, 24000)
xyz = b[1][0].reshape(-1, 3).contiguous()
stop = threading.Event()
launched = [0]


def load():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop.is_set():
            for _ in range(8):
                pops.furthestsampling_uniform(xyz, B, 24000, 8192)      # 4.7 ms each, 8 CUs
                launched[0] += 1
            s.synchronize()


th = threading.Thread(target=load, daemon=True)
if beside:
    th.start()
t0 = time.time()
for i in range(3):
    step(b[0][0], b[0][1], b[0][2])
    torch.cuda.synchronize()
    print("iteration %d done at %.0f s (%d background FPS launches)" % (i, time.time() - t0, launched[0]), flush=True)
stop.set()
if beside:
    th.join()
import torch.cuda.tunable as tunable
tunable.write_file()
print("wrote", out, sum(1 for _ in open(out)), "lines")
