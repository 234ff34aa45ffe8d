"""kNN lab (developer tool): runs the kNN shapes of the hot path a few times (for rocprofv3 --kernel-trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402

B = int(os.environ.get("B", "1"))
xyz = torch.from_numpy(make_batch(B, 24000)[0]).cuda()
shapes = [(24000, 24000, 33), (512, 24000, 32), (8192, 8192, 4), (24000, 8192, 3)]
for nq, nr, k in shapes:
    q, r = xyz[:, :nq].contiguous(), xyz[:, :nr].contiguous()
    for _ in range(3):
        knn_sorted(q, r, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        knn_sorted(q, r, k)
    e1.record()
    torch.cuda.synchronize()
    print("B=%d  %5d x %5d k=%2d  %.3f ms" % (B, nq, nr, k, e0.elapsed_time(e1) / 10), flush=True)
