"""Re-select the GEMM solutions of the kernels that run BESIDE the look-ahead work, under that contention.

PyTorch's TunableOp times every rocBLAS / hipBLASLt solution of a GEMM shape alone on the chip and keeps the fastest
(geot_amd/tuning/tunableop_gfx950_alone.csv).  For most shapes the winner is a launch of exactly 256 workgroups -- one per
CU, equal shares -- and such a launch waits for its slowest workgroup: beside the 8192-sample FPS of the look-ahead (8
workgroups on 8 CUs for 4.7 ms) it takes 50-85 % longer (profiles/r04_fps_beside.txt).  This tool re-tunes ONLY the shapes of
the second half of the supervised iteration -- the backward of the transformer blocks and of the patch encoder, which is what
the look-ahead runs beside (SupervisedStep.backward_rest_update) -- WHILE such an FPS runs on a side stream, and writes the
merged selection:  the `alone` entry for every other shape, the entry tuned under contention for these.

    python tools/tune_lookahead_gemms.py out.csv [clouds=8]      (GPU box; ~1 min)
    python tools/tune_lookahead_gemms.py out.csv fixmatch        the FixMatch+NTM iteration (2 + 2 clouds): replayed, its look-ahead
                                                                 graph runs beside the STUDENT'S FORWARD -- those shapes are re-tuned
The merge starts from GEOT_TUNE_BASE (default: the `alone` file), so two runs can be chained.
"""
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "40")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "5")
import geot_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402
from geot_amd import train_step as ts, tuning  # noqa: E402
from geot_amd.synth import make_batch, region_labels  # noqa: E402
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG  # noqa: E402
from geot_amd.pointops.functions import pointops as pops  # noqa: E402

ALONE = os.path.join(ROOT, "geot_amd", "tuning", "tunableop_gfx950_alone.csv")


def entries(path):
    head, rows = [], {}
    for line in open(path):
        p = line.rstrip("\n").split(",")
        if p[0] == "Validator":
            head.append(line)
        elif len(p) >= 3:
            rows[(p[0], p[1])] = line
    return head, rows


def main():
    out = sys.argv[1]
    fixmatch = len(sys.argv) > 2 and sys.argv[2] == "fixmatch"
    b = 6 if fixmatch else (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    dev = torch.device("cuda:0")
    xyz = make_batch(b, 24000)[0]
    pos = torch.from_numpy(xyz).to(dev)
    work = tempfile.mkstemp(prefix="geot_tune_", suffix=".csv")[1]
    os.remove(work)                                   # an empty selection: every shape met while tuning is on gets tuned
    tuning.enable(tune=True, path=work)
    tunable.tuning_enable(False)
    torch.manual_seed(0)
    if fixmatch:
        import numpy as np
        step = ts.build_fixmatch(dev, use_ddp=False)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
        lab, unl, strong = T(xyz[:2]), T(xyz[2:4]), T(xyz[2:4] * np.float32(1.04))
        z = torch.zeros(2, 1, dtype=torch.long, device=dev)
        data = {"pos": lab, "x": lab.transpose(1, 2).contiguous(), "cls": z, "y": T(region_labels(xyz[:2]))}
        data_u = {"pos_w": unl, "x_w": unl.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": strong,
                  "x_s": strong.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": unl}

        def iteration():
            pre = step.lookahead_work(data, data_u)                         # selection off: nothing recorded
            torch.cuda.synchronize()
            tunable.tuning_enable(True)                                     # the student's forward is tuned beside the FPS ...
            step.student_iteration(data, data_u, pre["geom_s"], pre["pseudo"], pre["knn"],
                                   after_forward=lambda: tunable.tuning_enable(False))      # ... and nothing behind it
    else:
        cls = torch.zeros(b, 1, dtype=torch.long, device=dev)
        target = torch.from_numpy(region_labels(xyz)).to(dev)
        step = ts.SupervisedStep(PointTransformer_seg_T(**TOOTH_SEG_CFG).to(dev))

        def iteration():
            loss, rest = step.forward_backward_head(pos, cls, target)      # selection off: library defaults, nothing recorded
            torch.cuda.synchronize()
            tunable.tuning_enable(True)
            step.backward_rest_update(rest)                                 # every shape in here is tuned beside the FPS
    flat = pos.reshape(-1, 3).contiguous()
    stop, launched = threading.Event(), [0]

    def load():                                       # the contention: back-to-back 8192-sample FPS launches on a stream of their own
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop.is_set():
                for _ in range(8):
                    pops.furthestsampling_uniform(flat, b, 24000, 8192)
                    launched[0] += 1
                s.synchronize()
    th = threading.Thread(target=load, daemon=True)
    th.start()
    t0 = time.time()
    for i in range(2):
        iteration()
        torch.cuda.synchronize()
        tunable.tuning_enable(False)
        print("iteration %d: %.0f s, %d FPS launches beside" % (i, time.time() - t0, launched[0]), flush=True)
    stop.set()
    th.join()
    results = tunable.get_results()                   # ((op signature, parameters, solution, time), ...): this process's tuning
    del step
    torch.cuda.synchronize()
    head, alone = entries(os.environ.get("GEOT_TUNE_BASE", ALONE))
    beside = {(r[0], r[1]): "%s,%s,%s,%s\n" % (r[0], r[1], r[2], r[3]) for r in results}
    if not beside:
        raise SystemExit("TunableOp wrote no results to %s" % work)
    with open(out, "w") as f:
        f.writelines(head)
        for key, line in alone.items():
            f.write(beside.get(key, line))
        for key, line in beside.items():
            if key not in alone:
                f.write(line)
    changed = sum(1 for k in beside if k in alone and alone[k].split(",")[2] != beside[k].split(",")[2])
    print("%d shapes tuned beside the FPS (%d with another solution than alone, %d new) -> %s" %
          (len(beside), changed, sum(1 for k in beside if k not in alone), out))
    with open(out + ".beside_only", "w") as f:
        f.writelines(head + list(beside.values()))


if __name__ == "__main__":
    main()
