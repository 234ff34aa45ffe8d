"""One-off soak: multi-commit pruned FPS against the unpruned kernel on many full-size clouds (both tie rules)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_cloud  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2, pointnet2_batch_cuda as p2b  # noqa: E402

bad = 0
for chunk in range(8):
    B, n, m = 32, 24000, 6000
    xyz = np.stack([make_cloud(n, 5000 + chunk * 100 + i, dup_frac=0.01 * (i % 4))[0] for i in range(B)])
    if chunk % 2:
        xyz = ((xyz * 512).round() / 512).astype(np.float32)        # quantised coordinates: many exact ties
    x = torch.from_numpy(xyz).cuda()
    res = {}
    for impl in ("multi", "basic"):
        os.environ["GEOT_FPS_IMPL"] = impl
        out = torch.empty(B, m, dtype=torch.int32, device="cuda")
        temp = torch.full((B, n), 1e10, device="cuda")
        p2b.furthest_point_sampling_wrapper(B, n, m, x, temp, out)
        res[impl] = (out.clone(), temp.clone(), p2.furthest_point_sampling(x, 2048).clone())
    ok = all(torch.equal(a, b) for a, b in zip(res["multi"], res["basic"]))
    bad += 0 if ok else 1
    print("chunk", chunk, "ok" if ok else "MISMATCH", flush=True)
del os.environ["GEOT_FPS_IMPL"]
print("mismatching chunks:", bad)
