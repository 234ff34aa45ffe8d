"""Time the dataloader-side ops on the GPU next to the reference's own CPU code.

    python tools/data_timing.py [--n 120000] [--dl 0.5] [--iters 20]

grid_subsampling: our HIP path vs oracle/_ref (the reference's C++ compiled here; it travels to the GPU box as
a built .so) on the same cloud; prepare_sample: HIP vs the numpy pipeline of tooth_dataset.py:108-147.
Prints one JSON line per op with algorithmic bytes and the achieved rate.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.openpoints.dataset import grid_subsampling, prepare_sample  # noqa: E402
from oracle import np_data  # noqa: E402  (CPU baseline leg only)


def gpu_ms(fn, iters):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / iters


def cpu_ms(fn, reps=5):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=120000)
    ap.add_argument("--dl", type=float, default=0.5)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    pc = (rng.standard_normal((a.n, 3)) * np.array([30, 20, 8]) + np.array([5, -40, 12])).astype(np.float32)
    feats = rng.standard_normal((a.n, 3)).astype(np.float32)
    labels = rng.integers(0, 17, a.n).astype(np.int32)
    dev = torch.device("cuda:0")
    tp, tf, tl = (torch.from_numpy(x).to(dev) for x in (pc, feats, labels))
    for name, f, l, fn, ln in (("points", None, None, None, None), ("points+features+labels", tf, tl, feats, labels)):
        ms = gpu_ms(lambda: grid_subsampling(tp, f, l, sampleDl=a.dl), a.iters)
        m = len(np_data.grid_subsampling(pc, None, None, a.dl)["points"])
        cols = 3 + (3 if f is not None else 0)
        alg = 4 * (a.n * (cols + (1 if l is not None else 0)) + m * (cols + (1 if l is not None else 0)))
        row = {"op": "grid_subsampling", "inputs": name, "n": a.n, "dl": a.dl, "voxels": m, "gpu_ms": round(ms, 4),
               "algorithmic_bytes": alg, "GB/s": round(alg / ms / 1e6, 2)}
        if np_data.have_reference():
            row["reference_cpu_ms"] = round(cpu_ms(lambda: np_data.grid_subsampling_reference(pc, fn, ln, a.dl)), 3)
            row["speedup"] = round(row["reference_cpu_ms"] / ms, 1)
        print(json.dumps(row), flush=True)
    m = 24000
    sel = rng.choice(a.n, m, replace=False)
    ts = torch.from_numpy(sel).to(dev)
    ms = gpu_ms(lambda: prepare_sample(tp, tl, ts, check=False), a.iters)

    def numpy_pipeline():
        q, c, s = np_data.pc_norm_numpy(pc)
        pos, y = torch.from_numpy(q[sel]).float(), torch.from_numpy(labels[sel]).long()
        w = torch.zeros(17)
        w += torch.histogram(y.float(), bins=17, range=(0., 17.))[0]
        return pos, y, w / w.sum()

    cms = cpu_ms(numpy_pipeline)
    alg = 4 * (2 * 3 * a.n + m * (3 + 3 + 1 + 2 + 2))      # two passes over the scan + gather in/out
    print(json.dumps({"op": "pc_norm+sample+class_weights", "n": a.n, "m": m, "gpu_ms": round(ms, 4),
                      "algorithmic_bytes": alg, "GB/s": round(alg / ms / 1e6, 2), "numpy_cpu_ms": round(cms, 3),
                      "speedup": round(cms / ms, 1)}), flush=True)
    # upload cost for scale: the scan has to cross PCIe once either way
    t = time.perf_counter()
    for _ in range(10):
        torch.from_numpy(pc).to(dev)
    torch.cuda.synchronize()
    print(json.dumps({"op": "upload (n,3) fp32 scan", "ms": round((time.perf_counter() - t) * 100, 4)}), flush=True)


if __name__ == "__main__":
    main()
