"""For a maintainer who HAS the authors' CUDA build of the reference (pointnet2._ext, pointops_cuda,
pointnet2_batch_cuda on an NVIDIA GPU): run the reference's own index-producing ops on seeded clouds and write
what they return, so this repository's oracle -- and through it every HIP kernel -- can be pinned against real
reference outputs for SURVEY.md section 8 rows a1-a3, a6, a9, a12 (K1, K1', K2, K6, K9, K12), which cannot be
executed in the build container (no nvcc, no NVIDIA device).

    cd <the reference checkout with its extensions built> && python <this repo>/tools/emit_reference_vectors.py OUT.npz
    cp OUT.npz <this repo>/tests/golden/external/

tests/test_external_vectors.py then (a) finds which squared-distance arithmetic of the oracle -- exact | fma |
fma_xy (csrc/geot_common.h sqdist3; nvcc's -fmad decides which one the CUDA binaries use) -- reproduces every
index of the file, (b) fails if none does, and (c) on a GPU box runs the HIP library built for that arithmetic
against the same file.  Without a file the test is skipped and DESIGN.md section 4 keeps saying "unpinned" for
these rows.

Only numpy + torch + the reference's own modules are needed; the clouds come from this repository's seeded
generator (geot_amd/synth.py, pure numpy), imported by file path so nothing else of the package is loaded.
"""
import importlib.util
import os
import sys

import numpy as np


def _synth():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("geot_synth", os.path.join(here, "geot_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


CASES = [  # (name, clouds, points, samples, duplicate fraction)
    ("c4096", 2, 4096, 1024, 0.0),
    ("c4096dup", 1, 4096, 1024, 0.01),
    ("c16000", 1, 16000, 4096, 0.0),
    ("c24000", 1, 24000, 8192, 0.0),
]


def main(out_path):
    import torch
    from pointnet2 import pointnet2_utils as pu                       # pointnet2/pointnet2_utils.py:48-288  (K1, K6, K9)
    from pointops.functions import pointops                          # pointops/functions/pointops.py:7-98  (K2, K12)
    try:
        from openpoints.cpp.pointnet2_batch import pointnet2_cuda as batch_cuda   # K1' (openpoints subsample.py)
    except Exception:                                                 # noqa: BLE001 -- optional extension
        batch_cuda = None
    synth = _synth()
    out = {"cases": np.array([c[0] for c in CASES])}
    for name, b, n, m, dup in CASES:
        xyz = synth.make_batch(b, n, start_index=500, dup_frac=dup)[0]
        x = torch.from_numpy(xyz).cuda()
        out[name + "_xyz"] = xyz
        out[name + "_m"] = np.int32(m)
        k1 = pu.furthest_point_sample(x, m)
        out[name + "_fps_k1"] = k1.cpu().numpy().astype(np.int32)
        if batch_cuda is not None:
            idx = torch.cuda.IntTensor(b, m)
            temp = torch.cuda.FloatTensor(b, n).fill_(1e10)
            batch_cuda.furthest_point_sampling_wrapper(b, n, m, x, temp, idx)
            out[name + "_fps_k1p"] = idx.cpu().numpy()
        out[name + "_fps_k2_xyz"] = pointops.fps(x, m).cpu().numpy()             # (b, m, 3): the sampled coordinates
        centres = pu.gather_operation(x.transpose(1, 2).contiguous(), k1).transpose(1, 2).contiguous()
        out[name + "_ball_r0.1_ns32"] = pu.ball_query(0.1, 32, x, centres).cpu().numpy().astype(np.int32)
        dist, idx3 = pu.three_nn(x, centres)
        out[name + "_three_nn_idx"] = idx3.cpu().numpy().astype(np.int32)
        out[name + "_three_nn_dist"] = dist.cpu().numpy()
        kidx, kdist = pointops.knn(centres, x, 5)                                  # heap order (knnquery_cuda_kernel.cu)
        out[name + "_knn5_idx"] = kidx.cpu().numpy().astype(np.int32)
        out[name + "_knn5_dist"] = kdist.cpu().numpy()
    np.savez_compressed(out_path, **out)
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "reference_cuda_vectors.npz")
