"""Kernel by kernel: what the 8192-sample FPS on a side stream costs the training graph M.  Run under
rocprofv3 --kernel-trace: phase A = 6 bare M replays, phase B = 6 M replays each with the FPS graph beside it; marker kernels
(a fill of a 12345-element tensor) separate the phases.  usage: fps_beside_trace.py run [clouds] | fps_beside_trace.py report trace.csv"""
import os, sys, csv, collections
if sys.argv[1] == "report":
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""),
                         "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]),
                         "wg%s v%s a%s lds%s" % (r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])))
    rows.sort()
    fps = [i for i, r in enumerate(rows) if "fps_pruned" in r[2] and (r[1] - r[0]) > 3e6]
    assert len(fps) >= 6, len(fps)
    last6 = fps[-6:]
    tB0 = rows[last6[0]][0]
    # phase B = from the first of the last six FPS launches to the end; phase A = the same number of main-queue kernels before it
    mainq = collections.Counter(r[3] for r in rows).most_common(1)[0][0]
    main = [r for r in rows if r[3] == mainq]
    B = [r for r in main if r[0] >= tB0]
    A = [r for r in main if r[0] < tB0][-len(B):]
    def agg(rs):
        d = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n, q, g, w in rs:
            k = (n[:70], g, w)
            d[k][0] += 1; d[k][1] += (e - s) / 1e3
        return d
    a, b = agg(A), agg(B)
    print("main queue: phase A (alone) %d kernels %.3f ms busy | phase B (FPS beside) %d kernels %.3f ms busy"
          % (len(A), sum(v[1] for v in a.values()) / 1e3, len(B), sum(v[1] for v in b.values()) / 1e3))
    diff = sorted(((b[k][1] - a.get(k, [0, 0])[1], k) for k in b), reverse=True)
    print("largest slow-downs per 6 steps (us): delta, alone, beside, launches, kernel, grid, wg")
    for d, k in diff[:16]:
        print("%+9.1f %9.1f %9.1f %5d  %-58s %-14s %s" % (d, a.get(k, [0, 0])[1], b[k][1], b[k][0], k[0][:58], k[1], k[2]))
    for r in rows:
        if "fps_pruned" in r[2] and (r[1] - r[0]) > 3e6:
            print("the FPS:", r[2][:60], r[4], r[5], "%.1f us" % ((r[1] - r[0]) / 1e3))
            break
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GEOT_GRAPH_LAUNCH", "fast"); os.environ.setdefault("GEOT_GRAPH_SPLIT", "0")

import geot_amd
import torch
from test_graph_step_gpu import _sup_batches, DEV
from geot_amd import train_step as ts, graph_step as gs
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
from geot_amd.pointops.functions import pointops as pops
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if os.environ.get("TUNED", "1") == "1":        # the GEMM selection bench.py runs with (geot_amd/tuning)
    from geot_amd import tuning
    tuning.enable(path=os.environ.get("TUNE_FILE"))
torch.manual_seed(0)
m = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(DEV)
step = ts.SupervisedStep(m)
call = gs.GraphedSupervisedStep(step)
b = _sup_batches(B, 24000)
for i in range(5):
    call(b[i % 2][0], b[i % 2][1], b[i % 2][2], next_pos=b[(i + 1) % 2][0])
xyz = b[0][0].reshape(-1, 3).contiguous()
gf = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s2):
    pops.furthestsampling_uniform(xyz, B, 24000, 8192)
torch.cuda.current_stream().wait_stream(s2); torch.cuda.synchronize()
with torch.cuda.graph(gf):
    idx = pops.furthestsampling_uniform(xyz, B, 24000, 8192)
side = torch.cuda.Stream()
torch.cuda.synchronize()
if "M" in call.graphs:                     # unsplit: the FPS graph beside the whole of M
    M = call.graphs["M"][0]
    for _ in range(6):
        M.replay()
    torch.cuda.synchronize()
    for _ in range(6):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gf.replay()
        M.replay()
        torch.cuda.current_stream().wait_stream(side)
else:                                      # split (GEOT_GRAPH_SPLIT=1): the FPS graph starts behind M1, beside M2
    M1, M2 = call.graphs["M1"][0], call.graphs["M2"][0]
    for _ in range(6):
        M1.replay(); M2.replay()
    torch.cuda.synchronize()
    for _ in range(6):
        M1.replay()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gf.replay()
        M2.replay()
        torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
