"""geot_gather_rows_csr_bn_cl at an FP stage with few targets: 8 clouds, n unknown points <- m known (their farthest-point
samples), C = 1536, Morton order of the targets.  python tools/lab/gr_small_time.py n m   (GEOT_GR_FORM=list: the list walk)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd.synth import make_batch  # noqa: E402


def main():
    from geot_amd import fused_norm as fn
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd.ext._common import call, ptr
    n, m = int(sys.argv[1]), int(sys.argv[2])
    B, C, dev = 8, 1536, torch.device("cuda")
    it = int(os.environ.get("ITER", "20"))
    xyz = torch.from_numpy(make_batch(B, 24000)[0]).to(dev)
    ids = p2.furthest_point_sampling(xyz, n).long()
    pos = torch.gather(xyz, 1, ids.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    known = pos[:, :m].contiguous()                 # a prefix of farthest-point samples = farthest-point samples
    d2, idx = p2.three_nn(pos, known)
    w = p2.fp_weights(d2)
    order = fn.local_spatial_order(known)
    rix = fn.ReverseIndex(idx, w, m, order)
    y, dz = torch.randn(B, n, C, device=dev), torch.randn(B, n, C, device=dev)
    sc, sh, mu, rs, c1, c2 = (torch.rand(C, device=dev) + 0.5 for _ in range(6))
    out = torch.empty(B, m, C, device=dev)

    def run():
        call("geot_gather_rows_csr_bn_cl", dev, B, C, n, m, 3, 1, ptr(y), ptr(dz), ptr(sc), ptr(sh), ptr(mu), ptr(rs), ptr(c1), ptr(c2),
             ptr(rix.ws), ptr(order), ptr(out))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    nbytes = 4.0 * B * C * (2 * n + m)
    print("gather_rows_csr_bn_cl %5d <- %4d, C = %d: %8.1f us  %7.1f MB  %5.2f TB/s" % (n, m, C, us, nbytes / 1e6, nbytes / us / 1e6))


if __name__ == "__main__":
    main()
